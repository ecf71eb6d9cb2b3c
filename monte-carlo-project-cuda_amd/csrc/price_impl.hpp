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
//     block, finished by a second tiny kernel: deterministic, no float atomics, no reliance on
//     pre-zeroed memory (SURVEY 2.4-2,7);
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

namespace mcamd {

template <typename T>
struct PriceArgs {
    StepConsts<T> c;
    uint64_t seed;
    uint64_t path_offset;
    uint64_t n_local;
    double control_mean;  // E[S_T] = S_start exp(r T_remaining)
};

template <typename T, bool WINDOW, bool LOGSPACE, int VR>
__global__ __launch_bounds__(kBlock) void price_kernel(PriceArgs<T> a, double *__restrict__ partials)
{
    constexpr bool ANTI = (VR & 1) != 0, CV = (VR & 2) != 0;
    constexpr int N = CV ? 5 : 2;
    const MathCtx<T> m = MathCtx<T>::init();
    const PhiloxKeys key = PhiloxKeys::make(a.seed);
    const StepConsts<T> c = resident(a.c);
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    double acc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0.0;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < a.n_local; i += stride) {
        const Sample<T> smp = simulate_sample<T, WINDOW, LOGSPACE, ANTI, WINDOW && !CV>(
            c, m, key, a.path_offset + i, c.S_start, c.Ik, c.n_sim);
        const double y = static_cast<double>(smp.pay);
        acc[0] += y;
        acc[1] = __builtin_fma(y, y, acc[1]);
        if (CV) {
            const double cc = static_cast<double>(smp.ctrl) - a.control_mean;
            acc[2] += cc;
            acc[3] = __builtin_fma(cc, cc, acc[3]);
            acc[4] = __builtin_fma(y, cc, acc[4]);
        }
    }
    block_sumN<kBlock, N>(acc);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) partials[static_cast<uint64_t>(N) * blockIdx.x + i] = acc[i];
    }
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
static hipError_t launch_price_t(const PathJob &j, double *d_partials, uint32_t grid, hipStream_t stream)
{
    const PriceArgs<T> a{make_consts<T>(j), j.seed, j.path_offset, j.n_local, j.control_mean};
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
