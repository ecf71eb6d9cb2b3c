// price.hip — in-register Monte Carlo pricing kernel for gfx950.
//
// One kernel fuses RNG -> GBM stepping -> payoff -> block reduction, like the reference's
// simulateOptionPriceMultipleBlockGPUwithReduce (inc/trajectories.cuh:54-113, one exact step)
// and simulateBulletOptionPriceMultipleBlockGPU[atomic] (inc/trajectories.cuh:115-271, N_STEPS
// steps + barrier window).  Differences by design:
//   - Philox counters in registers instead of a curandState array in HBM (no setup kernel);
//   - the path is the unit of work, indexed by its 64-bit GLOBAL id, so any shard of any job
//     draws the same numbers;
//   - per-thread fp64 (sum, sumsq) -> wave64 shuffle -> one LDS slot per wave -> one partial pair
//     per block, finished by a second tiny kernel: deterministic, no float atomics, no reliance
//     on pre-zeroed memory (SURVEY 2.4-2,7);
//   - the tail is handled by predicating the work, not the reduction.
// HBM traffic: 16 B per block written.  The kernel is VALU-bound (integer multiplies of Philox,
// transcendental ops of Box-Muller and exp).
#include "path_consts.hpp"

namespace mcamd {

template <typename T>
struct PriceArgs {
    StepConsts<T> c;
    uint64_t seed;
    uint64_t path_offset;
    uint64_t n_local;
};

template <typename T, bool WINDOW, bool LOGSPACE>
__global__ __launch_bounds__(kBlock) void price_kernel(PriceArgs<T> a, double *__restrict__ partials)
{
    const MathCtx<T> m = MathCtx<T>::init();
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    double s = 0.0, s2 = 0.0;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < a.n_local; i += stride) {
        const double pay = static_cast<double>(
            simulate_path<T, WINDOW, LOGSPACE>(a.c, m, a.seed, a.path_offset + i, a.c.S_start, a.c.Ik, a.c.n_sim));
        s += pay;
        s2 = __builtin_fma(pay, pay, s2);
    }
    block_sum2<kBlock>(s, s2);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = s;
        partials[2 * blockIdx.x + 1] = s2;
    }
}

// Final pass: sums n_pairs (a, b) pairs with ONE workgroup in a fixed order -> deterministic for a
// given launch shape.  1024 threads, 16 B loads, four independent accumulator pairs per thread so the
// L2 latency of the single workgroup is overlapped (39k pairs: ~10 us instead of ~60 us).
constexpr int kFinalBlock = 1024;

__global__ __launch_bounds__(kFinalBlock) void final_reduce_kernel(const double *__restrict__ partials,
                                                                  uint32_t n_pairs, double *__restrict__ out)
{
    using D2V = double __attribute__((ext_vector_type(2)));
    const D2V *p = reinterpret_cast<const D2V *>(partials);
    double s[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    uint32_t i = threadIdx.x;
    for (; i + 3 * kFinalBlock < n_pairs; i += 4 * kFinalBlock) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const D2V v = p[i + u * kFinalBlock];
            s[u] += v.x;
            s2[u] += v.y;
        }
    }
    for (; i < n_pairs; i += kFinalBlock) {
        const D2V v = p[i];
        s[0] += v.x;
        s2[0] += v.y;
    }
    double a = (s[0] + s[1]) + (s[2] + s[3]);
    double b = (s2[0] + s2[1]) + (s2[2] + s2[3]);
    block_sum2<kFinalBlock>(a, b);
    if (threadIdx.x == 0) {
        out[0] = a;
        out[1] = b;
    }
}

// One path per thread when a path is long (fine-grained blocks keep the tail short); for short paths
// (few steps) a thread takes several, so that a block still carries a few thousand path-steps and the
// partial array stays small (1-step pricer at 100M paths: 12k partial pairs instead of 390k).
uint32_t price_grid(uint64_t n_local, uint32_t n_sim)
{
    const uint64_t per_thread = n_sim >= 32 ? 1 : (32 + n_sim - 1) / n_sim;
    const uint64_t threads = (n_local + per_thread - 1) / per_thread;
    return clamp_grid((threads + kBlock - 1) / kBlock);
}

template <typename T>
static hipError_t launch_price_t(const PathJob &j, double *d_partials, uint32_t grid, hipStream_t stream)
{
    PriceArgs<T> a{make_consts<T>(j), j.seed, j.path_offset, j.n_local};
    const dim3 g(grid), b(kBlock);
    if (j.window) {
        if (j.logspace) hipLaunchKernelGGL((price_kernel<T, true, true>), g, b, 0, stream, a, d_partials);
        else hipLaunchKernelGGL((price_kernel<T, true, false>), g, b, 0, stream, a, d_partials);
    } else {
        if (j.logspace) hipLaunchKernelGGL((price_kernel<T, false, true>), g, b, 0, stream, a, d_partials);
        else hipLaunchKernelGGL((price_kernel<T, false, false>), g, b, 0, stream, a, d_partials);
    }
    return hipGetLastError();
}

hipError_t launch_price(const PathJob &j, double *d_partials, uint32_t grid, hipStream_t stream)
{
    return j.precision == 32 ? launch_price_t<float>(j, d_partials, grid, stream)
                             : launch_price_t<double>(j, d_partials, grid, stream);
}

hipError_t launch_final_reduce(const double *d_partials, uint32_t n_pairs, double *d_out, hipStream_t stream)
{
    hipLaunchKernelGGL(final_reduce_kernel, dim3(1), dim3(kFinalBlock), 0, stream, d_partials, n_pairs, d_out);
    return hipGetLastError();
}

}  // namespace mcamd
