// where do 58 us go before the fp64 step loop?  stamps inside a copy of MathCtx<double>::init
#include "price_impl.hpp"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace mcamd;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(kBlock) void k(PriceArgs<double> a, double *partials, uint64_t *st, int mode)
{
    const uint64_t w_entry = __builtin_amdgcn_s_memrealtime();
    __shared__ f64::D2 s_log[MCAMD_TAB_N];
    __shared__ f64::D2 s_sincos[MCAMD_TAB_N];
    __shared__ double s_exp_hi[256];
    __shared__ double s_exp_lo[256];
    for (int i = threadIdx.x; i < MCAMD_TAB_N; i += blockDim.x) {
        s_log[i] = f64::D2{kLogTab[i][0], kLogTab[i][1]};
        s_sincos[i] = f64::D2{kSinCosTab[i][0], kSinCosTab[i][1]};
    }
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        s_exp_hi[i] = kExpHiTab[i];
        s_exp_lo[i] = kExpLoTab[i];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const uint64_t w_copied = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    const uint64_t w_bar = __builtin_amdgcn_s_memrealtime();
    const MathCtx<double> m{f64::Tables{s_log, s_sincos, s_exp_hi, s_exp_lo}, f64::exp_c1_resident()};
    const PhiloxKeys key = PhiloxKeys::make(a.seed);
    const StepConsts<double> c = resident(a.c);
    double acc = 0;
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i < a.n_local && mode == 0)
        acc = simulate_sample<double, false, false, false>(c, m, key, i, c.S_start, c.Ik, c.n_sim).pay;
    const uint64_t w_end = __builtin_amdgcn_s_memrealtime();
    partials[static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        const uint64_t w = static_cast<uint64_t>(blockIdx.x) * 4 + threadIdx.x / 64;
        st[4 * w] = w_copied - w_entry; st[4 * w + 1] = w_bar - w_copied; st[4 * w + 2] = w_end - w_bar; st[4 * w + 3] = w_entry;
    }
}
int main()
{
    const uint64_t n = 10000000; const uint32_t steps = 252;
    PathJob j{}; const double dt = 1.0 / steps; j.drift = (0.1 - 0.02) * dt; j.vol = 0.2 * std::sqrt(dt); j.K = 100; j.S_start = 100; j.n_sim = steps; j.n_steps = steps; j.seed = 1234; j.n_local = n; j.precision = 64;
    PriceArgs<double> a{make_consts<double>(j), j.seed, 0, j.n_local, 0.0};
    const uint32_t grid = (n + kBlock - 1) / kBlock;
    double *p; uint64_t *st; CK(hipMalloc(&p, n * 8 + 4096)); CK(hipMalloc(&st, (size_t)grid * 4 * 4 * 8));
    for (int mode = 0; mode < 2; ++mode) {
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), 0, 0, a, p, st, mode);
        CK(hipDeviceSynchronize());
        std::vector<uint64_t> h((size_t)grid * 16); CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> c0, c1, c2;
        for (size_t w = 0; w < (size_t)grid * 4; ++w) { c0.push_back(h[4 * w] * 0.01); c1.push_back(h[4 * w + 1] * 0.01); c2.push_back(h[4 * w + 2] * 0.01); }
        auto med = [](std::vector<double> &v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
        printf("mode %d (0 = with step loop, 1 = tables only): copy us median %.2f p95 %.2f | barrier wait median %.2f p95 %.2f | loop median %.2f\n", mode,
               med(c0, 0.5), med(c0, 0.95), med(c1, 0.5), med(c1, 0.95), med(c2, 0.5));
    }
    return 0;
}
