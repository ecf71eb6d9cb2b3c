// price_impl.hpp — in-register Monte Carlo pricing kernel for gfx950 (included by price_f32.hip and
// price_f64.hip, one translation unit per path precision so the two compile in parallel).
//
// One kernel fuses RNG -> GBM stepping -> payoff -> block reduction, like the reference's
// simulateOptionPriceMultipleBlockGPUwithReduce (inc/trajectories.cuh:54-113, one exact step)
// and simulateBulletOptionPriceMultipleBlockGPU[atomic] (inc/trajectories.cuh:115-271, N_STEPS
// steps + barrier window).  Differences by design:
//   - Philox counters in registers instead of a curandState array in HBM (no setup kernel);
//   - the path is the unit of work, indexed by its 64-bit GLOBAL id, so any shard of any job
//     draws the same numbers;
//   - per-thread fp64 sums -> wave64 shuffle -> one LDS slot per wave -> one partial record per
//     block, summed in a fixed order by the LAST workgroup to finish (grid_finish, mc_device.hpp: one
//     launch, like the reference's in-kernel atomicAdd finish, inc/trajectories.cuh:77-111) or, for
//     grids with many records, by a second one-workgroup kernel: deterministic, no float atomics, no
//     reliance on pre-zeroed payoff memory (SURVEY 2.4-2,7);
//   - the tail is handled by predicating the work, not the reduction.
// HBM traffic: one record (16 or 40 bytes) per block.  The kernel is VALU-bound (integer multiplies
// of Philox, Box-Muller, exp).
//
// VR selects the opt-in variance reduction (new capability):
//   bit 0  antithetic: a sample is the pair (G, -G), its payoff the pair's mean;
//   bit 1  control variate: besides (sum y, sum y^2) the record carries (sum c, sum c^2, sum y c) of
//          the centred control c = S_T - E[S_T]; the host solves for beta (mcamd_finalize_cv).
#pragma once

#include "path_consts.hpp"
#include "nmc_compact.hpp"

namespace mcamd {

template <typename T>
struct PriceArgs {
    StepConsts<T> c;
    uint64_t seed;
    uint64_t path_offset;
    uint64_t n_local;
    double control_mean;  // E[S_T] = S_start exp(r T_remaining)
    GridFinish fin;       // where the grid's final record goes when the kernel finishes the sum itself
};

template <typename T, bool WINDOW, bool LOGSPACE, int VR>
__global__ __launch_bounds__(kBlock) void price_kernel(PriceArgs<T> a, double *__restrict__ partials)
{
    constexpr bool ANTI = (VR & 1) != 0, CV = (VR & 2) != 0;
    constexpr int N = CV ? 5 : 2;
    // the window-less log-space loop adds up pair sums and looks the sine up in the rotated table (PairSum)
    constexpr bool PAIRSUM = LOGSPACE && !WINDOW;
    const MathCtx<T> m = MathCtx<T>::template init<PAIRSUM>();
    const PhiloxKeys key = PhiloxKeys::make(a.seed);
    const StepConsts<T> c = resident(a.c);
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    double acc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0.0;
    auto add_sample = [&](const Sample<T> &smp) {
        const double y = static_cast<double>(smp.pay);
        acc[0] += y;
        acc[1] = __builtin_fma(y, y, acc[1]);
        if (CV) {
            const double cc = static_cast<double>(smp.ctrl) - a.control_mean;
            acc[2] += cc;
            acc[3] = __builtin_fma(cc, cc, acc[3]);
            acc[4] = __builtin_fma(y, cc, acc[4]);
        }
    };
    if constexpr (PAIRSUM) {
        // a thread owns kPairSumPaths consecutive paths and walks them together (mc_device.hpp pair_sums_of_paths);
        // the job's last thread may own fewer
        constexpr int NP = kPairSumPaths;
        for (uint64_t g = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; g * NP < a.n_local; g += stride) {
            T sums[NP];
            pair_sums_of_paths<T, NP>(m, key, a.path_offset + g * NP, c.n_sim, sums);
#pragma unroll
            for (int p = 0; p < NP; ++p)
                if (g * NP + p < a.n_local) add_sample(sample_from_pair_sum<T, ANTI>(c, m, sums[p], c.S_start, c.n_sim));
        }
    } else {
        for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < a.n_local; i += stride)
            add_sample(simulate_sample<T, WINDOW, LOGSPACE, ANTI, WINDOW && !CV>(c, m, key, a.path_offset + i, c.S_start,
                                                                               c.Ik, c.n_sim));
    }
    block_sumN<kBlock, N>(acc);
    grid_finish<kBlock, N>(acc, partials, a.fin);
}

// ---------------------------------------------------------------------------------------------
// Window payoff, plain estimator, MANY paths: the lane-compacting form (nmc_compact.hpp).  With a window that can
// close (bullet option: payoff only while P1 <= count <= P2) most paths are over long before maturity, but a
// wavefront of the kernel above runs until its LAST path is — with 64 unrelated paths per wavefront that is nearly
// always maturity.  Here a wavefront takes GROUPS of kPool x kSlice consecutive paths from a device-scope queue and
// runs each group as one compaction pool: lanes whose path is over are refilled with fresh paths, the few long-lived
// paths are parked in LDS and resumed 64 at a time.  Every path draws the same numbers and takes the same steps as
// in price_kernel (same Philox subsequence = global path id, same arithmetic), so its payoff is bit-identical; the
// sums differ by summation order only.  Chosen by the launcher when the job has enough groups to fill the chip.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kSlice = 128;                      // paths per pool slot: a group is kPool * kSlice = 1024 paths
constexpr uint64_t kGroupPaths = static_cast<uint64_t>(kPool) * kSlice;

template <typename T, bool LOGSPACE>
__global__ __launch_bounds__(kBlock) void price_window_compact_kernel(PriceArgs<T> a, double *__restrict__ partials,
                                                                      unsigned long long *__restrict__ queue)
{
    const MathCtx<T> m = MathCtx<T>::init();
    const PhiloxKeys key = PhiloxKeys::make(a.seed);
    StepConsts<T> c = resident(a.c);
    if (c.P2 >= kNoPath) c.P2 = kNoPath - 1;   // "count <= P2" must stay false for a lane without a path
    const uint32_t lane = threadIdx.x & (kWave - 1);
    __shared__ SurvivorBuf<T> s_parked[kBlock / kWave];
    SurvivorBuf<T> &buf = s_parked[threadIdx.x / kWave];
    const uint64_t n_groups = (a.n_local + kGroupPaths - 1) / kGroupPaths;
    double acc[2] = {0.0, 0.0};
    for (;;) {
        unsigned long long first = 0;
        if (lane == 0) first = atomicAdd(queue, 1ull);
        const uint64_t g = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(first >> 32))) << 32) |
                           __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(first));
        if (g >= n_groups) break;
        if (lane < kPool) {   // lane s describes slice s of the group: kSlice consecutive paths (the last may be short)
            const uint64_t base = g * kGroupPaths + static_cast<uint64_t>(lane) * kSlice;
            const uint64_t left = base < a.n_local ? a.n_local - base : 0;
            buf.pt_n[lane] = static_cast<uint32_t>(left < kSlice ? left : kSlice);
            buf.pt_cnt0[lane] = (left == 0 || c.Ik > c.P2) ? kNoPath : c.Ik;
            buf.pt_St0[lane] = c.S_start;
            buf.pt_log_start[lane] = T(0);
            buf.pt_subsequence[lane] = a.path_offset + base;
            buf.pt_sum[lane] = 0.0;
            buf.pt_sumsq[lane] = 0.0;
        }
        wave_lds_fence();
        uint64_t steps_run = 0, live_steps = 0;
        group_sums_compacted<T, LOGSPACE>(c, m, key, c.n_sim, buf, steps_run, live_steps);
        wave_lds_fence();
        if (lane < kPool) {
            acc[0] += buf.pt_sum[lane];
            acc[1] += buf.pt_sumsq[lane];
        }
        wave_lds_fence();   // the next group's description must not overtake these reads
    }
    block_sumN<kBlock, 2>(acc);
    grid_finish<kBlock, 2>(acc, partials, a.fin);
}

// Compaction pays once every SIMD can be kept busy with whole groups; below that the one-path-per-thread kernel's
// wider parallelism wins.  Measured crossover (fp64, B = 120, P2 = 50): ~0.9M paths at 252 steps, ~3M at 100 steps;
// at 10M x 252 the compacting kernel takes 1.2 ms against 3.4 (fp32: 0.41 against 1.13).
inline bool price_compacts(const PathJob &j, uint32_t compute_units)
{
    const uint64_t cus = compute_units ? compute_units : 256;
    return j.window && j.vr == 0 && j.n_sim >= 8 && j.n_local >= cus * 12 * kGroupPaths;
}

template <typename T>
static hipError_t launch_price_compact_t(const PathJob &j, double *d_partials, unsigned long long *d_queue, uint32_t grid,
                                         const GridFinish &fin, hipStream_t stream)
{
    const PriceArgs<T> a{make_consts<T>(j), j.seed, j.path_offset, j.n_local, j.control_mean, fin};
    const hipError_t e = hipMemsetAsync(d_queue, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    const dim3 g(grid), b(kBlock);
    if (j.logspace) hipLaunchKernelGGL((price_window_compact_kernel<T, true>), g, b, 0, stream, a, d_partials, d_queue);
    else hipLaunchKernelGGL((price_window_compact_kernel<T, false>), g, b, 0, stream, a, d_partials, d_queue);
    return hipGetLastError();
}

template <typename T, bool WINDOW, bool LOGSPACE>
static void launch_price_vr(const PriceArgs<T> &a, int vr, double *d_partials, uint32_t grid, hipStream_t stream)
{
    const dim3 g(grid), b(kBlock);
    switch (vr) {
        case 0: hipLaunchKernelGGL((price_kernel<T, WINDOW, LOGSPACE, 0>), g, b, 0, stream, a, d_partials); break;
        case 1: hipLaunchKernelGGL((price_kernel<T, WINDOW, LOGSPACE, 1>), g, b, 0, stream, a, d_partials); break;
        case 2: hipLaunchKernelGGL((price_kernel<T, WINDOW, LOGSPACE, 2>), g, b, 0, stream, a, d_partials); break;
        default: hipLaunchKernelGGL((price_kernel<T, WINDOW, LOGSPACE, 3>), g, b, 0, stream, a, d_partials); break;
    }
}

template <typename T>
static hipError_t launch_price_t(const PathJob &j, double *d_partials, uint32_t grid, const GridFinish &fin,
                                 hipStream_t stream)
{
    const PriceArgs<T> a{make_consts<T>(j), j.seed, j.path_offset, j.n_local, j.control_mean, fin};
    if (j.window) {
        if (j.logspace) launch_price_vr<T, true, true>(a, j.vr, d_partials, grid, stream);
        else launch_price_vr<T, true, false>(a, j.vr, d_partials, grid, stream);
    } else {
        if (j.logspace) launch_price_vr<T, false, true>(a, j.vr, d_partials, grid, stream);
        else launch_price_vr<T, false, false>(a, j.vr, d_partials, grid, stream);
    }
    return hipGetLastError();
}

}  // namespace mcamd
