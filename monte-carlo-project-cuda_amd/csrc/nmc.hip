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
// WAVE_PER_POINT is the gfx950-shaped one: a point is a task for ONE wavefront; its 64 lanes
// stride over the inner paths and the point's sum is a pure wave64 shuffle reduction — no LDS,
// no barrier, no atomics, against the reference's 1024-thread block + 5-barrier tree per task.
// Tasks are ordered step-major, so the long tasks (small step, many remaining steps) are issued
// first and the short ones fill the tail.  A point whose count already exceeds P2 can never pay
// (inc/nmc.cuh:53,330) and is skipped by the whole wave.
// Each inner path restarts from the stored (St, count); the reference's carry-over between
// successive inner paths of one thread is a defect and is not reproduced, and the
// output is written, not atomically added to unzeroed memory.
#include "path_consts.hpp"

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

template <typename T, bool WINDOW, int LAYOUT, bool LOGSPACE>
__global__ __launch_bounds__(kBlock) void nmc_wave_kernel(NmcArgs<T> a, double *__restrict__ partials)
{
    constexpr int kWaves = kBlock / kWave;
    const MathCtx<T> m = MathCtx<T>::init();
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const uint64_t wave_stride = static_cast<uint64_t>(gridDim.x) * kWaves;
    double psum = 0.0, psumsq = 0.0;
    for (uint64_t task = static_cast<uint64_t>(blockIdx.x) * kWaves + wave; task < a.n_points; task += wave_stride) {
        uint32_t step;
        uint64_t path;
        const uint64_t idx = point_index<T, WINDOW, LAYOUT>(a, task, step, path);
        const T St0 = a.prices[idx];
        const int32_t cnt0 = WINDOW ? a.counts[idx] : 0;
        const uint32_t remaining = a.n_steps - (step + 1);
        const uint64_t point_id = (a.path_offset + path) * a.n_steps + step;
        double acc = 0.0;
        if (!WINDOW || cnt0 <= a.c.P2) {
            const T ls = (WINDOW && LOGSPACE) ? log_ratio(St0, a.c.S_start) : T(0);
            for (uint32_t j = lane; j < a.n_inner; j += kWave)
                acc += static_cast<double>(simulate_path<T, WINDOW, LOGSPACE>(a.c, m, a.seed, point_id * a.n_inner + j,
                                                                               St0, cnt0, remaining, ls));
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            const double price = acc * a.scale;
            a.out[idx] = static_cast<T>(price);
            psum += price;
            psumsq = __builtin_fma(price, price, psumsq);
        }
    }
    block_sum2<kBlock>(psum, psumsq);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = psum;
        partials[2 * blockIdx.x + 1] = psumsq;
    }
}

template <typename T, bool WINDOW, int LAYOUT, bool LOGSPACE>
__global__ __launch_bounds__(kBlock) void nmc_block_kernel(NmcArgs<T> a, double *__restrict__ partials)
{
    const MathCtx<T> m = MathCtx<T>::init();
    double psum = 0.0, psumsq = 0.0;
    for (uint64_t task = blockIdx.x; task < a.n_points; task += gridDim.x) {
        uint32_t step;
        uint64_t path;
        const uint64_t idx = point_index<T, WINDOW, LAYOUT>(a, task, step, path);
        const T St0 = a.prices[idx];
        const int32_t cnt0 = WINDOW ? a.counts[idx] : 0;
        const uint32_t remaining = a.n_steps - (step + 1);
        const uint64_t point_id = (a.path_offset + path) * a.n_steps + step;
        double acc = 0.0, zero = 0.0;
        if (!WINDOW || cnt0 <= a.c.P2) {
            const T ls = (WINDOW && LOGSPACE) ? log_ratio(St0, a.c.S_start) : T(0);
            for (uint32_t j = threadIdx.x; j < a.n_inner; j += kBlock)
                acc += static_cast<double>(simulate_path<T, WINDOW, LOGSPACE>(a.c, m, a.seed, point_id * a.n_inner + j,
                                                                               St0, cnt0, remaining, ls));
        }
        block_sum2<kBlock>(acc, zero);
        if (threadIdx.x == 0) {
            const double price = acc * a.scale;
            a.out[idx] = static_cast<T>(price);
            psum += price;
            psumsq = __builtin_fma(price, price, psumsq);
        }
        __syncthreads();  // block_sum2's LDS slots are reused by the next task
    }
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = psum;
        partials[2 * blockIdx.x + 1] = psumsq;
    }
}

uint32_t nmc_grid(const NmcJob &job, int variant)
{
    if (variant == MCAMD_NMC_BLOCK_PER_POINT) return clamp_grid(job.n_points);
    return clamp_grid((job.n_points + (kBlock / kWave) - 1) / (kBlock / kWave));
}

template <typename T, bool WINDOW, int LAYOUT>
static void launch_variant(const NmcArgs<T> &a, int variant, bool logspace, double *d_partials, uint32_t grid,
                           hipStream_t stream)
{
    const dim3 g(grid), b(kBlock);
    if (variant == MCAMD_NMC_BLOCK_PER_POINT) {
        if (logspace) hipLaunchKernelGGL((nmc_block_kernel<T, WINDOW, LAYOUT, true>), g, b, 0, stream, a, d_partials);
        else hipLaunchKernelGGL((nmc_block_kernel<T, WINDOW, LAYOUT, false>), g, b, 0, stream, a, d_partials);
    } else {
        if (logspace) hipLaunchKernelGGL((nmc_wave_kernel<T, WINDOW, LAYOUT, true>), g, b, 0, stream, a, d_partials);
        else hipLaunchKernelGGL((nmc_wave_kernel<T, WINDOW, LAYOUT, false>), g, b, 0, stream, a, d_partials);
    }
}

template <typename T>
static hipError_t launch_nmc_t(const NmcJob &job, int layout, int variant, const void *d_prices,
                               const int32_t *d_counts, void *d_point_prices, double *d_partials, uint32_t grid,
                               hipStream_t stream)
{
    NmcArgs<T> a;
    a.c = make_consts<T>(job.path);
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
    const bool w = job.path.window;
    if (layout == MCAMD_STEP_MAJOR) {
        if (w) launch_variant<T, true, MCAMD_STEP_MAJOR>(a, variant, job.path.logspace, d_partials, grid, stream);
        else launch_variant<T, false, MCAMD_STEP_MAJOR>(a, variant, job.path.logspace, d_partials, grid, stream);
    } else {
        if (w) launch_variant<T, true, MCAMD_PATH_MAJOR>(a, variant, job.path.logspace, d_partials, grid, stream);
        else launch_variant<T, false, MCAMD_PATH_MAJOR>(a, variant, job.path.logspace, d_partials, grid, stream);
    }
    return hipGetLastError();
}

hipError_t launch_nmc_inner(const NmcJob &job, int layout, int variant, const void *d_prices, const int32_t *d_counts,
                            void *d_point_prices, double *d_partials, uint32_t grid, hipStream_t stream)
{
    return job.path.precision == 32
               ? launch_nmc_t<float>(job, layout, variant, d_prices, d_counts, d_point_prices, d_partials, grid, stream)
               : launch_nmc_t<double>(job, layout, variant, d_prices, d_counts, d_point_prices, d_partials, grid,
                                      stream);
}

}  // namespace mcamd
