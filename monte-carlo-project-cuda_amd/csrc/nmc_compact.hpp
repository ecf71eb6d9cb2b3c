// nmc_compact.hpp — lane compaction for paths with a barrier window: one wavefront runs a POOL of paths and keeps its
// lanes busy with paths that are still alive.  Used by the nested-MC kernels (nmc.hip: the pool is the continuation
// paths of a group of kPool stored points) and by bullet pricing of many paths (price_impl.hpp: the pool is kPool slices
// of consecutive paths).
//
// With a barrier window (inc/nmc.cuh:47-66, inc/trajectories.cuh:149: payoff only while P1 <= count <= P2) a path is
// over as soon as its count passes P2, and most are after a few dozen steps — but the few that climbed above the
// barrier live to maturity, and a wavefront that waits for its last lane runs ~250 steps with two or three lanes
// working (measured on BASELINE configs[3] with the reference's window B = 120, P1 = 10, P2 = 50: 37 % of the
// executed lane-steps belong to paths that are still alive).  So the wavefront does not wait: once kCompactBelow or
// fewer of its lanes are still running, those lanes park their path (product, barrier accumulator, exponent, count,
// path index, point, next Philox block: 36 bytes in fp64) in the wavefront's LDS buffer and the wavefront starts the
// next 64 fresh paths; whenever 64 parked paths have gathered they are resumed together, full width, under the same
// rule; the last batch of a pool runs to completion whatever its width.  Philox is counter-based, so a resumed path
// continues its own stream at its own block whatever lane it lands in: every path's payoff is bit-identical to the
// uncompacted loop, only the summation order of a point's sum changes (deterministically: the schedule depends on
// the pool alone, not on timing).
// A pool spans kPool points with the same number of steps to go (nested MC: one step of kPool adjacent outer paths),
// which leaves far fewer half-empty batches at the end than one point alone would (lane efficiency 0.84 -> 0.94).
#pragma once

#include "mc_device.hpp"

namespace mcamd {

#ifndef MCAMD_POOL            // overridable for the same-box comparisons of profiles/r02_nmc_variants.txt only
#define MCAMD_POOL 8
#endif
#ifndef MCAMD_COMPACT_BELOW
#define MCAMD_COMPACT_BELOW 48
#endif
constexpr uint32_t kPool = MCAMD_POOL;                       // points per group (a group is one task of the kernels)
constexpr uint32_t kCompactBelow = MCAMD_COMPACT_BELOW;      // hand over when this many lanes or fewer still run
constexpr uint32_t kSurvivorCap = kWave + kCompactBelow;     // at most 63 parked + one hand-over

// the loops below rely on these: kCompactBelow >= kWave would make every non-last resumed batch hand over at once and
// re-park all of its lanes (the resume loop never ends); the pool's descriptors are written by lanes 0..kPool-1
static_assert(kCompactBelow >= 1 && kCompactBelow < static_cast<uint32_t>(kWave), "MCAMD_COMPACT_BELOW must be in [1, 64)");
static_assert(kPool >= 1 && kPool <= static_cast<uint32_t>(kWave), "MCAMD_POOL must be in [1, 64]");

constexpr int32_t kNoPath = 0x7fffffff;

// One wavefront's parked paths (structure of arrays: lane-consecutive slots, conflict-free).
template <typename T>
struct SurvivorBuf {
    T a[kSurvivorCap];             // fp64 product form: P;  fp32: St;  log-space: ln(St / S_start)
    T b[kSurvivorCap];             // fp64 product form: kq (barrier accumulator);  unused otherwise
    int32_t k[kSurvivorCap];       // fp64 product form: integer exponent;  unused otherwise
    int32_t count[kSurvivorCap];
    uint32_t j[kSurvivorCap];      // index of the path among its point's continuation paths
    uint32_t blk[kSurvivorCap];    // next Philox block of its stream
    uint32_t slot[kSurvivorCap];   // which point of the group it belongs to
    // the group's points (written by the wavefront when it takes the group)
    uint64_t pt_subsequence[kPool];   // Philox subsequence of the point's continuation path 0
    T pt_St0[kPool];                  // start price (nested MC: the stored outer price)
    T pt_log_start[kPool];            // ln(St0 / S_start) in exponent units (0 where unused)
    int32_t pt_cnt0[kPool];           // start count (the stored outer count); kNoPath: point absent or already closed
    uint32_t pt_n[kPool];             // paths of the point
    double pt_sum[kPool];             // running sum of the point's payoffs (lane 0 adds each batch's total)
    double pt_sumsq[kPool];           // and of their squares
};

// What a kernel declares in LDS per wavefront: the buffer when there is a window, nothing otherwise.
template <typename T, bool WINDOW>
struct ParkedPaths : SurvivorBuf<T> {};
template <typename T>
struct ParkedPaths<T, false> {};

// A continuation path in flight.
template <typename T>
struct InnerLane {
    PathState<T> ps;   // product form
    T acc;             // log-space form
    int32_t count;     // barrier count; kNoPath in a lane that holds no path (so "count <= P2" means: a live path)
    uint32_t blk, slot;
    uint64_t subsequence;   // Philox subsequence of the path (its point's first one + the path's index)
};

__device__ __forceinline__ void park(SurvivorBuf<float> &buf, uint32_t slot, const InnerLane<float> &L, bool logspace)
{
    buf.a[slot] = logspace ? L.acc : L.ps.St;
}
__device__ __forceinline__ void unpark(const SurvivorBuf<float> &buf, uint32_t slot, InnerLane<float> &L, bool logspace)
{
    if (logspace) L.acc = buf.a[slot];
    else L.ps.St = buf.a[slot];
}
__device__ __forceinline__ void park(SurvivorBuf<double> &buf, uint32_t slot, const InnerLane<double> &L, bool logspace)
{
    if (logspace) {
        buf.a[slot] = L.acc;
    } else {
        buf.a[slot] = L.ps.a.P;
        buf.b[slot] = L.ps.kq;
        buf.k[slot] = L.ps.a.k;
    }
}
__device__ __forceinline__ void unpark(const SurvivorBuf<double> &buf, uint32_t slot, InnerLane<double> &L, bool logspace)
{
    if (logspace) {
        L.acc = buf.a[slot];
    } else {
        L.ps.a.P = buf.a[slot];
        L.ps.kq = buf.b[slot];
        L.ps.a.k = buf.k[slot];
    }
}

// The start price of a resumed path (fp64 keeps it beside the factored product; fp32 carries the price itself).
__device__ __forceinline__ void set_start_price(PathState<float> &, float) {}
__device__ __forceinline__ void set_start_price(PathState<double> &ps, double St0) { ps.S0 = St0; }

// LDS traffic between lanes of one wavefront: the hardware serves a wavefront's LDS instructions in order; this
// keeps the compiler from moving them across the hand-over.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One step of a continuation path: the GBM update and the barrier count (inc/nmc.cuh:56-58).
template <typename T, bool LOGSPACE>
__device__ __forceinline__ void inner_step(const StepConsts<T> &c, const MathCtx<T> &m, InnerLane<T> &L, T x_or_z)
{
    if (LOGSPACE) {   // x = drift + vol G, the step's exponent: ln(St / S_start) grows by it
        L.acc += x_or_z;
        L.count += (c.logB > L.acc) ? 1 : 0;
    } else {
        L.ps.step(x_or_z, m);
        L.count += L.ps.below_barrier(c, m);
    }
}

// The draws of one Philox block of a path, as inner_step consumes them: the steps' exponents x = drift + vol G, which
// the product form multiplies the price by (e^x) and the log-space form adds up.
template <typename T, bool LOGSPACE>
struct BlockDraws {
    T v[Normals<T>::kPerBlock];
    __device__ __forceinline__ void fill(const StepConsts<T> &c, const MathCtx<T> &m, const PhiloxKeys &key,
                                         uint64_t subsequence, uint32_t block)
    {
        constexpr int NB = Normals<T>::kPerBlock;
        Exponents<T> ex;
        ex.fill(m, c, key, subsequence, block);
#pragma unroll
        for (int s = 0; s < NB; ++s) v[s] = ex.x[s];
    }
};

// Runs the wavefront's paths through their FULL Philox blocks (n_full of them per path) until live_limit or fewer are
// still running (0: to the end).  UNIFORM: every running lane is at the same block (fresh paths), so the block index
// stays scalar and the first Philox round keeps its scalar half.
template <typename T, bool LOGSPACE, bool UNIFORM>
__device__ __forceinline__ void run_batch(const StepConsts<T> &c, const MathCtx<T> &m, const PhiloxKeys &key,
                                          InnerLane<T> &L, uint32_t n_full, uint32_t live_limit, uint64_t &wave_steps,
                                          uint64_t &live_steps)
{
    constexpr int NB = Normals<T>::kPerBlock;
    const uint64_t subsequence = L.subsequence;
    uint32_t kb = 0;   // blocks this call ran (wave-uniform); UNIFORM: also the block index of every running lane
    for (; !UNIFORM || kb < n_full; ++kb) {
        // liveness is read off the count every time (one compare straight into a lane mask) rather than carried
        // as a flag, which the compiler would keep re-materialising in a vector register
        const bool run = UNIFORM ? (L.count <= c.P2) : (L.count <= c.P2 && L.blk < n_full);
        const uint32_t live = __builtin_amdgcn_readfirstlane(
            static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(run))));
        if (live <= live_limit) break;
        live_steps += static_cast<uint64_t>(live) * NB;
        if (run) {
            BlockDraws<T, LOGSPACE> d;
            d.fill(c, m, key, subsequence, UNIFORM ? kb : L.blk);
#pragma unroll
            for (int s = 0; s < NB; ++s) inner_step<T, LOGSPACE>(c, m, L, d.v[s]);
            if (!UNIFORM) ++L.blk;
        }
    }
    if (UNIFORM) L.blk = kb;
    wave_steps += static_cast<uint64_t>(NB) * static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(kb));
}

// Sums of the window payoffs of the continuation paths of a group's points, into buf.pt_sum[s] for point s (the
// caller zeroes them), their squares into buf.pt_sumsq[s].  The points are described in buf.pt_* (pt_cnt0 = kNoPath:
// skip; pt_n paths each); all have
// `remaining` steps to go and n_inner paths; path j of point s uses Philox subsequence pt_subsequence[s] + j, as the
// uncompacted loop does.
template <typename T, bool LOGSPACE>
__device__ __forceinline__ void group_sums_compacted(const StepConsts<T> &c, const MathCtx<T> &m, const PhiloxKeys &key,
                                                     uint32_t remaining, SurvivorBuf<T> &buf, uint64_t &wave_steps,
                                                     uint64_t &live_steps)
{
    constexpr int NB = Normals<T>::kPerBlock;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t n_full = remaining / NB;
    const uint32_t rem = remaining - n_full * NB;
    uint32_t parked = 0;   // wave-uniform

    InnerLane<T> L;
    L.ps = PathState<T>::start(T(0));
    L.slot = 0;

    // finished paths take the steps of the partial last block (rem of them) and pay; paths still running wait in
    // the buffer.  one_point: the point every lane's path belongs to, kPool when they are mixed
    auto settle = [&](uint32_t one_point) {
        double pay = 0.0;
        const bool done = L.count <= c.P2 && L.blk >= n_full;
        const uint32_t finishing = static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(done)));
        if (done) {
            if (rem != 0) {
                BlockDraws<T, LOGSPACE> d;
                d.fill(c, m, key, L.subsequence, n_full);
#pragma unroll
                for (int s = 0; s < NB - 1; ++s)
                    if (static_cast<uint32_t>(s) < rem) inner_step<T, LOGSPACE>(c, m, L, d.v[s]);
            }
            const T St = LOGSPACE ? exp_of_logreturn(c.S_start, L.acc, m) : L.ps.value(m);
            pay = static_cast<double>(payoff<T, true>(St, L.count, c));
        }
        if (finishing != 0) {
            // the batch's payoffs, point by point, onto the points' running sums: a few dozen instructions in the
            // batches where a path reaches maturity, against eight vector registers for per-lane sums all the time
            if (one_point != kPool) {   // a batch of fresh paths: they all belong to that point
                const double total = wave_sum(pay), total_sq = wave_sum(pay * pay);
                if (lane == 0) {
                    buf.pt_sum[one_point] += total;
                    buf.pt_sumsq[one_point] += total_sq;
                }
            } else {
#pragma unroll
                for (uint32_t s = 0; s < kPool; ++s) {
                    const double mine = L.slot == s ? pay : 0.0;
                    const double total = wave_sum(mine), total_sq = wave_sum(mine * mine);
                    if (lane == 0) {
                        buf.pt_sum[s] += total;
                        buf.pt_sumsq[s] += total_sq;
                    }
                }
            }
            wave_steps += rem;
            live_steps += static_cast<uint64_t>(finishing) * rem;
        }
        const bool waits = L.count <= c.P2 && L.blk < n_full;
        const uint64_t mask = __builtin_amdgcn_ballot_w64(waits);
        if (mask != 0) {
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                                             __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
            if (waits) {
                const uint32_t at = parked + below;
                park(buf, at, L, LOGSPACE);
                buf.count[at] = L.count;
                buf.j[at] = static_cast<uint32_t>(L.subsequence - buf.pt_subsequence[L.slot]);
                buf.blk[at] = L.blk;
                buf.slot[at] = L.slot;
            }
            parked += static_cast<uint32_t>(__builtin_popcountll(mask));
            wave_lds_fence();
        }
    };
    // resumes the last `take` parked paths, one per lane
    auto resume = [&](uint32_t take) {
        parked -= take;
        L.count = kNoPath;
        if (lane < take) {
            const uint32_t at = parked + lane;
            unpark(buf, at, L, LOGSPACE);
            L.count = buf.count[at];
            L.blk = buf.blk[at];
            L.slot = buf.slot[at];
            L.subsequence = buf.pt_subsequence[L.slot] + buf.j[at];
            set_start_price(L.ps, buf.pt_St0[L.slot]);
        }
        wave_lds_fence();
    };

    uint32_t slot = 0, j0 = 0;   // the next fresh paths: j0.. of point `slot`
    for (;;) {
        while (slot < kPool && __builtin_amdgcn_readfirstlane(buf.pt_cnt0[slot]) == kNoPath) ++slot;   // absent or closed
        if (slot < kPool) {   // the next 64 fresh paths
            const T St0 = buf.pt_St0[slot], log_start = buf.pt_log_start[slot];
            const uint32_t n_here = __builtin_amdgcn_readfirstlane(buf.pt_n[slot]);
            L.ps = PathState<T>::start(St0);
            if (!LOGSPACE) L.ps.arm_barrier(c.logB - log_start);   // ln(B / St0) = ln(B / S_start) - ln(St0 / S_start)
            L.acc = log_start;
            L.slot = slot;
            L.subsequence = buf.pt_subsequence[slot] + (j0 + lane);
            L.count = j0 + lane < n_here ? buf.pt_cnt0[slot] : kNoPath;
            L.blk = 0;
            run_batch<T, LOGSPACE, true>(c, m, key, L, n_full, kCompactBelow, wave_steps, live_steps);
            settle(slot);
            j0 += kWave;
            if (j0 >= n_here) {
                j0 = 0;
                ++slot;
            }
            while (slot < kPool && __builtin_amdgcn_readfirstlane(buf.pt_cnt0[slot]) == kNoPath) ++slot;
        }
        const bool none_fresh = slot >= kPool;
        // a full wavefront of parked paths — or, when no fresh ones are left, whatever is parked
        while (parked >= kWave || (none_fresh && parked != 0)) {
            const uint32_t take = parked < kWave ? parked : kWave;
            const bool last = none_fresh && parked == take;   // nothing will join them: run to the end
            resume(take);
            run_batch<T, LOGSPACE, false>(c, m, key, L, n_full, last ? 0u : kCompactBelow, wave_steps, live_steps);
            settle(kPool);
        }
        if (none_fresh) break;
    }
}

}  // namespace mcamd
