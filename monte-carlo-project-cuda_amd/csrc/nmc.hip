// nmc.hip — nested Monte Carlo, inner stage, for gfx950.
//
// For every stored outer point (step, path) run n_inner continuation paths of
// n_steps - 1 - step steps from the stored (St, count), apply the window payoff, average and
// discount by exp(-rT) (full maturity, as inc/nmc.cuh:101,268,379 do).  Replaces
//   compute_nmc_one_block_per_point  inc/nmc.cuh:12-108   -> MCAMD_NMC_BLOCK_PER_POINT
//   compute_nmc_optimal              inc/nmc.cuh:280-386  -> MCAMD_NMC_WAVE_PER_POINT
// Both variants give the same per-point prices up to fp64 summation order: the random numbers
// of inner path j of point q are Philox(seed, subsequence = q * n_inner + j), not the leftover
// state of whichever thread happens to run it (the reference re-uses one curandState per thread
// across tasks, so its numbers depend on the launch shape).
//
// WAVE_PER_POINT is the gfx950-shaped one: points are priced by ONE wavefront each way — no workgroup barrier,
// against the reference's 1024-thread block + 5-barrier tree per task.  A task is a GROUP of kPool points (one
// step of kPool adjacent outer paths); tasks are ordered step-major, so the long ones (small step, many remaining
// steps) come first and the short ones fill the tail.  A point whose count already exceeds P2 can never pay
// (inc/nmc.cuh:53,330) and none of its paths is started.  Tasks are handed out dynamically: the grid is
// persistent (a few workgroups per CU) and every wavefront pulls the next group from one device-scope counter.
// With the bullet window task lengths differ widely, and under a static assignment a finished wavefront idles
// until the slowest wavefront of its workgroup retires (measured: VALU 85 % busy); pulled tasks keep every
// wavefront busy until the queue is empty.  One returning atomic per group: 2M dequeues over ~0.2 s, far below
// the ~88 per microsecond one counter sustains (MI355X_MICROARCH.md, dequeue).  Every wavefront leaves its loop
// on the first index past the end, so the grid always drains.
// With a window, the group's continuation paths are ONE pool for the wavefront's lane compaction
// (nmc_compact.hpp): lanes whose path is over are refilled instead of waited for.  Without a window every path
// runs every step and the group's points are priced one after the other, lanes striding over the inner paths.
// Each inner path restarts from the stored (St, count); the reference's carry-over between
// successive inner paths of one thread is a defect and is not reproduced, and the
// output is written, not atomically added to unzeroed memory.
#include "path_consts.hpp"
#include "nmc_compact.hpp"

#include "mcamd.h"

namespace mcamd {

template <typename T>
struct NmcArgs {
    StepConsts<T> c;       // n_sim unused (per task)
    uint64_t seed;
    uint64_t path_offset;
    uint64_t n_local;
    uint64_t n_points;     // n_local * n_steps
    uint32_t n_steps;
    uint32_t n_inner;
    double scale;          // exp(-rT) / n_inner
    const T *prices;
    const int32_t *counts;
    T *out;
};

template <typename T, bool WINDOW, int LAYOUT>
__device__ __forceinline__ uint64_t point_index(const NmcArgs<T> &a, uint64_t task, uint32_t &step, uint64_t &path)
{
    step = static_cast<uint32_t>(task / a.n_local);
    path = task - static_cast<uint64_t>(step) * a.n_local;
    return LAYOUT == MCAMD_STEP_MAJOR ? task : path * a.n_steps + step;
}

// A task of the wave-per-point and fused kernels is a GROUP: one step of kPool adjacent outer paths.  Groups are cut
// at multiples of kPool of the GLOBAL path id, not of the shard's local index: group G of a step holds global paths
// [G kPool, (G + 1) kPool), of which a shard prices the ones it owns (its first and last group may be partial).  A
// compaction pool's schedule — and with it the summation order inside each of its points' means — depends on which
// points share the pool, so with globally cut groups every group that lies inside a shard is priced exactly as the
// whole job prices it: per-point prices are bit-identical under any sharding, except in a shard's partial edge groups,
// which differ from the whole job by fp64 summation order only (each path's payoff is the same bits everywhere).
// With a window the group's continuation paths are one pool for the lane compaction of nmc_compact.hpp; without one
// its points are simply priced one after the other (fixed order per point: bit-identical under any sharding).
// lead = path_offset mod kPool: slots of the shard's first group that belong to the previous shard.
template <typename T>
__device__ __forceinline__ uint32_t group_lead(const NmcArgs<T> &a)
{
    return static_cast<uint32_t>(a.path_offset % kPool);
}
template <typename T>
__device__ __forceinline__ uint64_t groups_per_step_of(const NmcArgs<T> &a)
{
    return (group_lead(a) + a.n_local + kPool - 1) / kPool;
}
template <typename T, int LAYOUT>
__device__ __forceinline__ uint64_t stored_index(const NmcArgs<T> &a, uint32_t step, uint64_t path)
{
    return LAYOUT == MCAMD_STEP_MAJOR ? static_cast<uint64_t>(step) * a.n_local + path : path * a.n_steps + step;
}

// Prices the points (step, path0 + s), s = 0 .. kPool - 1, that lie inside the shard (0 <= path0 + s < n_local; path0 is
// the LOCAL index of the group's slot 0 and is negative in a shard's partial first group) with one wavefront and adds
// them to the wavefront's record (lane 0): rec = {sum of point prices, sum of squares, wave-steps executed, lane-steps
// of paths with an open window}.
template <typename T, bool WINDOW, int LAYOUT, bool LOGSPACE>
__device__ __forceinline__ void price_group(const NmcArgs<T> &a, const StepConsts<T> &c, const MathCtx<T> &m,
                                            const PhiloxKeys &key, const T *prices, const int32_t *counts, uint32_t step,
                                            int64_t path0, ParkedPaths<T, WINDOW> &parked, double (&rec)[kNmcRecord])
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t remaining = a.n_steps - (step + 1);
    // slot s of the group holds a point of this shard
    auto in_shard = [&](uint32_t s) {
        const int64_t p = path0 + static_cast<int64_t>(s);
        return p >= 0 && p < static_cast<int64_t>(a.n_local);
    };
    if constexpr (WINDOW) {
        if (lane < kPool) {   // lane s describes point s of the group
            const bool present = in_shard(lane);
            const uint64_t idx = stored_index<T, LAYOUT>(a, step, present ? static_cast<uint64_t>(path0 + lane) : 0u);
            const T St0 = prices[idx];
            int32_t cnt0 = counts[idx];
            // a point whose count is already beyond P2 can never pay (inc/nmc.cuh:53,330): no path of it is started
            if (!present || cnt0 > c.P2) cnt0 = kNoPath;
            parked.pt_St0[lane] = St0;
            parked.pt_cnt0[lane] = cnt0;
            parked.pt_log_start[lane] =
                (cnt0 != kNoPath && (LOGSPACE || sizeof(T) == 8)) ? log_ratio(St0, c.S_start) : T(0);
            // global point id = global path * n_steps + step (a.path_offset + path0 >= 0: the group's global start)
            parked.pt_subsequence[lane] =
                ((a.path_offset + static_cast<uint64_t>(path0 + lane)) * a.n_steps + step) * a.n_inner;
            parked.pt_n[lane] = a.n_inner;
            parked.pt_sum[lane] = 0.0;
            parked.pt_sumsq[lane] = 0.0;
        }
        wave_lds_fence();
        uint64_t steps_run = 0, live_steps = 0;   // wave-uniform; a pool may hold billions of lane-steps
        group_sums_compacted<T, LOGSPACE>(c, m, key, remaining, parked, steps_run, live_steps);
        wave_lds_fence();
#pragma unroll
        for (uint32_t s = 0; s < kPool; ++s) {
            if (lane == 0 && in_shard(s)) {
                const double price = parked.pt_sum[s] * a.scale;
                a.out[stored_index<T, LAYOUT>(a, step, static_cast<uint64_t>(path0 + s))] = static_cast<T>(price);
                rec[0] += price;
                rec[1] = __builtin_fma(price, price, rec[1]);
            }
        }
        if (lane == 0) {
            rec[2] += static_cast<double>(steps_run);
            rec[3] += static_cast<double>(live_steps);
        }
        wave_lds_fence();   // the next group's description must not overtake this group's last reads
    } else {
        for (uint32_t s = 0; s < kPool; ++s) {
            if (!in_shard(s)) continue;   // wave-uniform
            const uint64_t local = static_cast<uint64_t>(path0 + s);
            const uint64_t idx = stored_index<T, LAYOUT>(a, step, local);
            const T St0 = prices[idx];
            const uint64_t point_id = (a.path_offset + local) * a.n_steps + step;
            double acc = 0.0;
            uint64_t steps_run = 0;   // 64-bit: n_inner / 64 passes of up to 2^32 steps each
            for (uint32_t j = lane; j < a.n_inner; j += kWave)
                acc += static_cast<double>(simulate_path<T, WINDOW, LOGSPACE>(c, m, key, point_id * a.n_inner + j, St0, 0,
                                                                               remaining, T(0), &steps_run));
            acc = wave_sum(acc);
            if (lane == 0) {
                const double price = acc * a.scale;
                a.out[idx] = static_cast<T>(price);
                rec[0] += price;
                rec[1] = __builtin_fma(price, price, rec[1]);
                rec[2] += static_cast<double>(steps_run);   // lane 0 takes part in every pass over the inner paths
                rec[3] += static_cast<double>(a.n_inner) * remaining;   // no window: every path runs every step
            }
        }
    }
}

template <typename T, bool WINDOW, int LAYOUT, bool LOGSPACE>
__global__ __launch_bounds__(kBlock) void nmc_wave_kernel(NmcArgs<T> a, double *__restrict__ partials,
                                                          unsigned long long *__restrict__ queue)
{
    const MathCtx<T> m = MathCtx<T>::init();
    const PhiloxKeys key = PhiloxKeys::make(a.seed);
    const StepConsts<T> c = resident(a.c);
    const int lane = threadIdx.x & (kWave - 1);
    __shared__ ParkedPaths<T, WINDOW> s_parked[kBlock / kWave];   // one buffer per wavefront (nmc_compact.hpp)
    double rec[kNmcRecord] = {0.0, 0.0, 0.0, 0.0};
    const uint64_t groups_per_step = groups_per_step_of(a);
    const uint64_t n_groups = groups_per_step * a.n_steps;   // step-major: the long tasks come first
    const int64_t lead = group_lead(a);
    for (;;) {
        unsigned long long first = 0;
        if (lane == 0) first = atomicAdd(queue, 1ull);
        const uint64_t g = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(first >> 32))) << 32) |
                           __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(first));
        if (g >= n_groups) break;
        const uint32_t step = static_cast<uint32_t>(g / groups_per_step);
        const int64_t path0 = static_cast<int64_t>((g - static_cast<uint64_t>(step) * groups_per_step) * kPool) - lead;
        price_group<T, WINDOW, LAYOUT, LOGSPACE>(a, c, m, key, a.prices, a.counts, step, path0,
                                                 s_parked[threadIdx.x / kWave], rec);
    }
    block_sumN<kBlock, kNmcRecord>(rec);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < kNmcRecord; ++i) partials[kNmcRecord * static_cast<uint64_t>(blockIdx.x) + i] = rec[i];
    }
}

// One WORKGROUP per point (compute_nmc_one_block_per_point, inc/nmc.cuh:12-108).  COMPACT (the default with a window):
// each of the workgroup's four wavefronts takes a quarter of the point's continuation paths (a multiple of 64) and runs
// them as a compaction pool of its own (nmc_compact.hpp), so a wavefront refills its lanes as paths leave the window
// instead of waiting for its slowest lane; the pool is the point's alone, so a point's price does not depend on which
// other points are priced beside it.  !COMPACT (MCAMD_NMC_BLOCK_PER_POINT_PLAIN): path j goes to thread j mod 256 and
// every wavefront waits for its last lane — no shared machinery with the compacting kernels, which is why the
// differential fuzzers (tools/fuzz_nmc.py) keep it as their reference.
template <typename T, bool WINDOW, int LAYOUT, bool LOGSPACE, bool COMPACT>
__global__ __launch_bounds__(kBlock) void nmc_block_kernel(NmcArgs<T> a, double *__restrict__ partials)
{
    constexpr int kWaves = kBlock / kWave;
    const MathCtx<T> m = MathCtx<T>::init();
    const PhiloxKeys key = PhiloxKeys::make(a.seed);
    const StepConsts<T> c = resident(a.c);
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = threadIdx.x / kWave;
    __shared__ ParkedPaths<T, WINDOW && COMPACT> s_parked[kWaves];
    // this wavefront's share of a point's continuation paths: [j_lo, j_hi), whole wavefront-loads except the last
    const uint32_t share = ((a.n_inner + kWaves - 1) / kWaves + kWave - 1) / kWave * kWave;
    const uint32_t j_lo = wave * share < a.n_inner ? wave * share : a.n_inner;
    const uint32_t j_hi = j_lo + share < a.n_inner ? j_lo + share : a.n_inner;
    double psum = 0.0, psumsq = 0.0, pwork = 0.0, plive = 0.0;
    for (uint64_t task = blockIdx.x; task < a.n_points; task += gridDim.x) {
        uint32_t step;
        uint64_t path;
        const uint64_t idx = point_index<T, WINDOW, LAYOUT>(a, task, step, path);
        const T St0 = a.prices[idx];
        const int32_t cnt0 = WINDOW ? a.counts[idx] : 0;
        const uint32_t remaining = a.n_steps - (step + 1);
        const uint64_t point_id = (a.path_offset + path) * a.n_steps + step;
        double acc = 0.0;
        uint64_t steps_run = 0, live_steps = 0;   // 64-bit: a point may hold more than 2^32 lane-steps
        if constexpr (WINDOW && COMPACT) {
            ParkedPaths<T, true> &buf = s_parked[wave];
            if (lane < kPool) {   // slot 0: this wavefront's share of the point; the other slots stay empty
                const bool mine = lane == 0 && j_hi > j_lo && cnt0 <= c.P2;
                buf.pt_St0[lane] = St0;
                buf.pt_cnt0[lane] = mine ? cnt0 : kNoPath;
                buf.pt_log_start[lane] = (mine && (LOGSPACE || sizeof(T) == 8)) ? log_ratio(St0, c.S_start) : T(0);
                buf.pt_subsequence[lane] = point_id * a.n_inner + j_lo;
                buf.pt_n[lane] = j_hi - j_lo;
                buf.pt_sum[lane] = 0.0;
                buf.pt_sumsq[lane] = 0.0;
            }
            wave_lds_fence();
            group_sums_compacted<T, LOGSPACE>(c, m, key, remaining, buf, steps_run, live_steps);
            wave_lds_fence();
            acc = lane == 0 ? buf.pt_sum[0] : 0.0;
            wave_lds_fence();   // the next point's description must not overtake this read
        } else {
            if (!WINDOW || cnt0 <= c.P2) {
                const T ls = (WINDOW && (LOGSPACE || sizeof(T) == 8)) ? log_ratio(St0, c.S_start) : T(0);
                for (uint32_t j = threadIdx.x; j < a.n_inner; j += kBlock)
                    acc += static_cast<double>(simulate_path<T, WINDOW, LOGSPACE>(
                        c, m, key, point_id * a.n_inner + j, St0, cnt0, remaining, ls, &steps_run, &live_steps));
            }
        }
        // wave-steps and live lane-steps: each wavefront's first lane runs every pass that wavefront makes
        const bool first_lane = lane == 0;
        double pt[3] = {acc, first_lane ? static_cast<double>(steps_run) : 0.0,
                        first_lane ? static_cast<double>(WINDOW ? live_steps : 0ull) : 0.0};
        block_sumN<kBlock, 3>(pt);
        acc = pt[0];
        const double work = pt[1];
        const double live = WINDOW ? pt[2] : static_cast<double>(a.n_inner) * remaining;
        if (threadIdx.x == 0) {
            const double price = acc * a.scale;
            a.out[idx] = static_cast<T>(price);
            psum += price;
            psumsq = __builtin_fma(price, price, psumsq);
            pwork += work;
            plive += live;
        }
        __syncthreads();  // block_sumN's LDS slots are reused by the next task
    }
    if (threadIdx.x == 0) {
        partials[kNmcRecord * static_cast<uint64_t>(blockIdx.x)] = psum;
        partials[kNmcRecord * static_cast<uint64_t>(blockIdx.x) + 1] = psumsq;
        partials[kNmcRecord * static_cast<uint64_t>(blockIdx.x) + 2] = pwork;
        partials[kNmcRecord * static_cast<uint64_t>(blockIdx.x) + 3] = plive;
    }
}

// ---------------------------------------------------------------------------------------------
// Fused outer + inner stage in ONE launch.  Replaces compute_nmc_one_block_per_point_with_outter
// (inc/nmc.cuh:113-275).  A persistent grid; every wavefront works through two device-scope queues:
//   stage 1  slices of 64 consecutive outer paths (one lane per path): simulate, store every (St, count), then
//            publish — s_waitcnt vmcnt(0), agent-scope release fence, one atomic add on `done`;
//   stage 2  once `done` has reached the number of slices (relaxed poll by lane 0 with s_sleep, then an agent-
//            scope acquire fence by the polling wave, whose own loads follow it), the groups of the inner stage,
//            exactly as nmc_wave_kernel pulls and prices them.
// No grid-wide barrier and no assumption about residency: a wavefront enters stage 2 only when the stage-1 queue is
// EMPTY, i.e. every slice has been TAKEN by a wavefront that is running and that waits for nothing while it
// simulates its slice — so the count a stage-2 wavefront polls for is always reached, whether or not the whole grid
// is resident (another process may hold part of the chip).  The wait happens once per wavefront, in the first
// ~100 us of a launch that runs for ~0.2 s at BASELINE configs[3].
// Outer stream: (outer_seed, subsequence = global path id); inner stream and pools as nmc_wave_kernel, so the
// stored arrays and the per-point prices are bit-identical to the two-launch route.
// (r02's form — a workgroup owned whole outer paths, simulated them, then priced only its own points behind a
// workgroup barrier — needed no device-scope hand-off but left every workgroup with ~23 ms of private work: at the
// end of the launch the last workgroups finished one by one, 225 ms against the two-launch route's 188.)
// ---------------------------------------------------------------------------------------------
struct FusedQueues {
    unsigned long long *slices;   // stage-1 queue: next slice to simulate            (zero at launch)
    unsigned long long *groups;   // stage-2 queue: next group to price               (zero at launch)
    unsigned int *done;           // slices whose rows are stored and published        (zero at launch)
};

template <typename T, bool WINDOW, int LAYOUT, bool LOGSPACE>
__global__ __launch_bounds__(kBlock) void nmc_fused_kernel(NmcArgs<T> a, uint64_t outer_seed, T *prices,
                                                          int32_t *counts, double *__restrict__ partials, FusedQueues q)
{
    constexpr int kWaves = kBlock / kWave;
    constexpr int NB = Normals<T>::kPerBlock;
    const MathCtx<T> m = MathCtx<T>::init();
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const StepConsts<T> c = resident(a.c);
    auto next = [&](unsigned long long *counter) {   // one returning atomic per task, broadcast to the wavefront
        unsigned long long first = 0;
        if (lane == 0) first = atomicAdd(counter, 1ull);
        return (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(first >> 32))) << 32) |
               __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(first));
    };

    // ---- stage 1: outer trajectories (inc/nmc.cuh:144-202), one slice of 64 paths per wavefront at a time ----
    const uint64_t n_slices = (a.n_local + kWave - 1) / kWave;
    {
        const PhiloxKeys outer_key = PhiloxKeys::make(outer_seed);   // its 20 registers are free again after this stage
        for (;;) {
            const uint64_t sl = next(q.slices);
            if (sl >= n_slices) break;
            const uint64_t path = sl * kWave + lane;
            if (path < a.n_local) {
                PathState<T> ps = PathState<T>::start(c.S_start);
                int32_t cnt = c.Ik;
                Exponents<T> ex;
                for (uint32_t step = 0; step < a.n_steps; ++step) {
                    if (step % NB == 0) ex.fill(m, c, outer_key, a.path_offset + path, step / NB);
                    T x = ex.x[0];
#pragma unroll
                    for (int j = 1; j < NB; ++j) x = (step % NB == static_cast<uint32_t>(j)) ? ex.x[j] : x;
                    ps.step(x, m);
                    const T St = ps.value(m);
                    if (WINDOW) cnt += (c.B > St) ? 1 : 0;
                    const uint64_t idx = stored_index<T, LAYOUT>(a, step, path);
                    prices[idx] = St;
                    if (WINDOW) counts[idx] = cnt;
                }
            }
            // publish the slice: every lane's stores have left the wavefront, the XCD's L2 is written back, then the count
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(q.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- every slice has been taken; wait until they are all published (see the header: always reached) ----
    {
        unsigned int seen = 0;
        do {
            if (lane == 0) seen = __hip_atomic_load(q.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            seen = __builtin_amdgcn_readfirstlane(seen);
            if (seen < n_slices) __builtin_amdgcn_s_sleep(32);
        } while (seen < n_slices);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const PhiloxKeys key = PhiloxKeys::make(a.seed);

    // ---- stage 2: the inner stage, as nmc_wave_kernel ----
    __shared__ ParkedPaths<T, WINDOW> s_parked[kWaves];   // one buffer per wavefront (nmc_compact.hpp)
    double rec[kNmcRecord] = {0.0, 0.0, 0.0, 0.0};
    const uint64_t groups_per_step = groups_per_step_of(a);
    const uint64_t n_groups = groups_per_step * a.n_steps;   // step-major: the long tasks come first
    const int64_t lead = group_lead(a);
    for (;;) {
        const uint64_t g = next(q.groups);
        if (g >= n_groups) break;
        const uint32_t step = static_cast<uint32_t>(g / groups_per_step);
        const int64_t path0 = static_cast<int64_t>((g - static_cast<uint64_t>(step) * groups_per_step) * kPool) - lead;
        price_group<T, WINDOW, LAYOUT, LOGSPACE>(a, c, m, key, prices, counts, step, path0, s_parked[wave], rec);
    }
    block_sumN<kBlock, kNmcRecord>(rec);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < kNmcRecord; ++i) partials[kNmcRecord * static_cast<uint64_t>(blockIdx.x) + i] = rec[i];
    }
}

uint32_t nmc_fused_grid(const NmcJob &job)
{
    // the same persistent grid as the wave-per-point kernel (its stage 2 IS that kernel), and never fewer workgroups
    // than the outer stage has slices for: a slice is one wavefront's worth of outer paths
    const uint64_t per_block = static_cast<uint64_t>(kBlock / kWave);
    const uint64_t slices = (job.path.n_local + kWave - 1) / kWave;
    const uint64_t need = (slices + per_block - 1) / per_block;
    const uint64_t inner = nmc_grid(job, MCAMD_NMC_WAVE_PER_POINT);
    const uint64_t resident = static_cast<uint64_t>(job.compute_units ? job.compute_units : 256) * 8;
    const uint64_t want = inner > need ? inner : (need < resident ? need : resident);
    return static_cast<uint32_t>(want < 1 ? 1 : want);
}

template <typename T, bool WINDOW, int LAYOUT>
static void launch_fused_variant(const NmcArgs<T> &a, uint64_t outer_seed, bool logspace, T *prices, int32_t *counts,
                                 double *d_partials, const FusedQueues &q, uint32_t grid, hipStream_t stream)
{
    const dim3 g(grid), b(kBlock);
    if (logspace)
        hipLaunchKernelGGL((nmc_fused_kernel<T, WINDOW, LAYOUT, true>), g, b, 0, stream, a, outer_seed, prices, counts,
                           d_partials, q);
    else
        hipLaunchKernelGGL((nmc_fused_kernel<T, WINDOW, LAYOUT, false>), g, b, 0, stream, a, outer_seed, prices, counts,
                           d_partials, q);
}

uint32_t nmc_grid(const NmcJob &job, int variant)
{
    if (variant == MCAMD_NMC_BLOCK_PER_POINT || variant == MCAMD_NMC_BLOCK_PER_POINT_PLAIN) return clamp_grid(job.n_points);
    // wave per point: persistent grid, 8 workgroups per CU (all that can be resident), tasks pulled from a queue
    const uint64_t groups = (job.path.path_offset % kPool + job.path.n_local + kPool - 1) / kPool * job.path.n_steps;   // the kernel's tasks
    const uint64_t per_block = static_cast<uint64_t>(kBlock / kWave);
    const uint64_t need = (groups + per_block - 1) / per_block;
    const uint64_t resident = static_cast<uint64_t>(job.compute_units ? job.compute_units : 256) * 8;
    return static_cast<uint32_t>(need < 1 ? 1 : (need < resident ? need : resident));
}

template <typename T, bool WINDOW, int LAYOUT>
static void launch_variant(const NmcArgs<T> &a, int variant, bool logspace, double *d_partials,
                           unsigned long long *d_queue, uint32_t grid, hipStream_t stream)
{
    const dim3 g(grid), b(kBlock);
    if (variant == MCAMD_NMC_BLOCK_PER_POINT) {
        if (logspace) hipLaunchKernelGGL((nmc_block_kernel<T, WINDOW, LAYOUT, true, true>), g, b, 0, stream, a, d_partials);
        else hipLaunchKernelGGL((nmc_block_kernel<T, WINDOW, LAYOUT, false, true>), g, b, 0, stream, a, d_partials);
    } else if (variant == MCAMD_NMC_BLOCK_PER_POINT_PLAIN) {
        if (logspace) hipLaunchKernelGGL((nmc_block_kernel<T, WINDOW, LAYOUT, true, false>), g, b, 0, stream, a, d_partials);
        else hipLaunchKernelGGL((nmc_block_kernel<T, WINDOW, LAYOUT, false, false>), g, b, 0, stream, a, d_partials);
    } else {
        if (logspace)
            hipLaunchKernelGGL((nmc_wave_kernel<T, WINDOW, LAYOUT, true>), g, b, 0, stream, a, d_partials, d_queue);
        else
            hipLaunchKernelGGL((nmc_wave_kernel<T, WINDOW, LAYOUT, false>), g, b, 0, stream, a, d_partials, d_queue);
    }
}

template <typename T>
static NmcArgs<T> make_args(const NmcJob &job, const void *d_prices, const int32_t *d_counts, void *d_point_prices)
{
    NmcArgs<T> a;
    a.c = make_consts<T>(job.path);
    // the compacting kernels mark a lane without a path by count = INT32_MAX (kNoPath) and test liveness as
    // count <= P2: keep P2 below that (no count reaches it: counts start at a stored int32 and grow by one per step)
    if (a.c.P2 >= kNoPath) a.c.P2 = kNoPath - 1;
    a.seed = job.path.seed;
    a.path_offset = job.path.path_offset;
    a.n_local = job.path.n_local;
    a.n_points = job.n_points;
    a.n_steps = job.path.n_steps;
    a.n_inner = job.n_inner;
    a.scale = job.discount / static_cast<double>(job.n_inner);
    a.prices = static_cast<const T *>(d_prices);
    a.counts = d_counts;
    a.out = static_cast<T *>(d_point_prices);
    return a;
}

template <typename T>
static hipError_t launch_fused_t(const NmcJob &job, uint64_t outer_seed, int layout, void *d_prices, int32_t *d_counts,
                                 void *d_point_prices, double *d_partials, const FusedQueues &q, uint32_t grid,
                                 hipStream_t stream)
{
    const NmcArgs<T> a = make_args<T>(job, d_prices, d_counts, d_point_prices);
    T *pr = static_cast<T *>(d_prices);
    const bool w = job.path.window, ls = job.path.logspace;
    if (layout == MCAMD_STEP_MAJOR) {
        if (w) launch_fused_variant<T, true, MCAMD_STEP_MAJOR>(a, outer_seed, ls, pr, d_counts, d_partials, q, grid, stream);
        else launch_fused_variant<T, false, MCAMD_STEP_MAJOR>(a, outer_seed, ls, pr, d_counts, d_partials, q, grid, stream);
    } else {
        if (w) launch_fused_variant<T, true, MCAMD_PATH_MAJOR>(a, outer_seed, ls, pr, d_counts, d_partials, q, grid, stream);
        else launch_fused_variant<T, false, MCAMD_PATH_MAJOR>(a, outer_seed, ls, pr, d_counts, d_partials, q, grid, stream);
    }
    return hipGetLastError();
}

hipError_t launch_nmc_fused(const NmcJob &job, uint64_t outer_seed, int layout, void *d_prices, int32_t *d_counts,
                            void *d_point_prices, double *d_partials, unsigned long long *d_queue, uint32_t grid,
                            hipStream_t stream)
{
    // the context's 64-byte counter block: word 0 = stage-2 queue (as the wave kernel), bytes 16..19 = the pricing
    // kernels' arrival ticket (zero between launches: left alone), word 3 = stage-1 queue, bytes 32..35 = `done`
    hipError_t e = hipMemsetAsync(d_queue, 0, sizeof(unsigned long long), stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_queue + 3, 0, 2 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    const FusedQueues q{d_queue + 3, d_queue, reinterpret_cast<unsigned int *>(d_queue + 4)};
    return job.path.precision == 32
               ? launch_fused_t<float>(job, outer_seed, layout, d_prices, d_counts, d_point_prices, d_partials, q, grid, stream)
               : launch_fused_t<double>(job, outer_seed, layout, d_prices, d_counts, d_point_prices, d_partials, q, grid,
                                        stream);
}

template <typename T>
static hipError_t launch_nmc_t(const NmcJob &job, int layout, int variant, const void *d_prices,
                               const int32_t *d_counts, void *d_point_prices, double *d_partials,
                               unsigned long long *d_queue, uint32_t grid, hipStream_t stream)
{
    const NmcArgs<T> a = make_args<T>(job, d_prices, d_counts, d_point_prices);
    const bool w = job.path.window, ls = job.path.logspace;
    if (layout == MCAMD_STEP_MAJOR) {
        if (w) launch_variant<T, true, MCAMD_STEP_MAJOR>(a, variant, ls, d_partials, d_queue, grid, stream);
        else launch_variant<T, false, MCAMD_STEP_MAJOR>(a, variant, ls, d_partials, d_queue, grid, stream);
    } else {
        if (w) launch_variant<T, true, MCAMD_PATH_MAJOR>(a, variant, ls, d_partials, d_queue, grid, stream);
        else launch_variant<T, false, MCAMD_PATH_MAJOR>(a, variant, ls, d_partials, d_queue, grid, stream);
    }
    return hipGetLastError();
}

hipError_t launch_nmc_inner(const NmcJob &job, int layout, int variant, const void *d_prices, const int32_t *d_counts,
                            void *d_point_prices, double *d_partials, unsigned long long *d_queue, uint32_t grid,
                            hipStream_t stream)
{
    if (variant == MCAMD_NMC_WAVE_PER_POINT) {
        const hipError_t e = hipMemsetAsync(d_queue, 0, sizeof(unsigned long long), stream);
        if (e != hipSuccess) return e;
    }
    return job.path.precision == 32 ? launch_nmc_t<float>(job, layout, variant, d_prices, d_counts, d_point_prices,
                                                          d_partials, d_queue, grid, stream)
                                    : launch_nmc_t<double>(job, layout, variant, d_prices, d_counts, d_point_prices,
                                                           d_partials, d_queue, grid, stream);
}

}  // namespace mcamd
