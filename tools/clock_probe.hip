// clock_probe.hip — DIAGNOSTIC build of the in-register pricing loop with cycle stamps around it.
//
// Answers "at what clock does the chip run the fp64 (or fp32) step loop?" (MI355X_MICROARCH.md, DVFS
// give-back item 6): the body of price_kernel (csrc/price_impl.hpp — the same loops, the same launch shapes: the
// default pair-sum loop with two paths per thread, PAIRSUM = true, or the product form, simulate_sample) is stamped
// once before and once after the path loop with s_memtime (shader cycles) and
// s_memrealtime (100 MHz), by lane 0 of every wave; in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz,
// median over waves, read from the last launch after >= 2 s of back-to-back launches.  Stamps go to a buffer
// of their own; no output value depends on them.  The shipped library contains no stamp.
//
// Also derives, from the stamps alone, the cycles one SIMD spends per wave-iteration of the step loop
// (busy cycles of the SIMD / wave-iterations it ran), to set beside the ISA issue-cycle count of
// tools/count_valu_slots.py and the SQ_ACTIVE_INST_VALU counter.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imonte-carlo-project-cuda_amd/csrc tools/clock_probe.hip -o tools/clock_probe
//   tools/clock_probe [paths=10000000] [steps=252] [seconds=2.5]   -> one JSON line per precision
#include "price_impl.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace mcamd;

#define CK(x)                                                                               \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) {                                                             \
            std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);      \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

template <typename T, bool PAIRSUM>
__global__ __launch_bounds__(kBlock) void probe_kernel(PriceArgs<T> a, double *__restrict__ partials,
                                                       uint64_t *__restrict__ stamps)
{
    // optional dynamic LDS (launch parameter): only there to cap the workgroups a CU can hold, for the occupancy sweep
    extern __shared__ char occupancy_pad[];
    if (threadIdx.x == 1023) occupancy_pad[0] = 0;   // never true for 256-thread blocks: keeps the symbol referenced
    const uint64_t w_entry = __builtin_amdgcn_s_memrealtime();   // kernel entry, before the LDS tables are filled
    const MathCtx<T> m = MathCtx<T>::template init<PAIRSUM>();
    const PhiloxKeys key = PhiloxKeys::make(a.seed);
    const StepConsts<T> c = resident(a.c);
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    double acc[2] = {0.0, 0.0};
    auto add = [&](const Sample<T> &smp) {
        const double y = static_cast<double>(smp.pay);
        acc[0] += y;
        acc[1] = __builtin_fma(y, y, acc[1]);
    };
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (PAIRSUM) {
        constexpr int NP = kPairSumPaths;
        for (uint64_t g = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; g * NP < a.n_local; g += stride) {
            T sums[NP];
            pair_sums_of_paths<T, NP>(m, key, a.path_offset + g * NP, c.n_sim, sums);
#pragma unroll
            for (int p = 0; p < NP; ++p)
                if (g * NP + p < a.n_local) add(sample_from_pair_sum<T, false>(c, m, sums[p], c.S_start, c.n_sim));
        }
    } else {
        for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < a.n_local; i += stride)
            add(simulate_sample<T, false, false, false>(c, m, key, a.path_offset + i, c.S_start, c.Ik, c.n_sim));
    }
    // the payoff must be complete before the closing stamp: make the stamp depend on it
    const uint32_t done = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(__double2hiint(acc[0])));
    asm volatile("" ::"s"(done));
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    block_sumN<kBlock, 2>(acc);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = acc[0];
        partials[2 * blockIdx.x + 1] = acc[1];
    }
    if ((threadIdx.x & 63) == 0) {
        const uint64_t wave = static_cast<uint64_t>(blockIdx.x) * (kBlock / 64) + threadIdx.x / 64;
        stamps[4 * wave + 0] = t0;
        stamps[4 * wave + 1] = t1 | ((w0 - w_entry) << 48);   // top 16 bits: 100 MHz ticks spent before the step loop
        stamps[4 * wave + 2] = w0;
        stamps[4 * wave + 3] = w1;
    }
}

template <typename T, bool PAIRSUM = true>
static int run(const char *name, uint64_t n_paths, uint32_t n_steps, double seconds, uint32_t pad_bytes = 0)
{
    constexpr uint64_t kPathsPerThread = PAIRSUM ? kPairSumPaths : 1;
    PathJob j{};
    const double dt = 1.0 / n_steps, r = 0.1, v = 0.2;
    j.drift = (r - 0.5 * v * v) * dt;
    j.vol = v * std::sqrt(dt);
    j.K = 100.0;
    j.B = 0.0;
    j.S_start = 100.0;
    j.n_sim = n_steps;
    j.n_steps = n_steps;
    j.seed = 1234;
    j.n_local = n_paths;
    j.precision = sizeof(T) * 8;
    PriceArgs<T> a{make_consts<T>(j), j.seed, 0, j.n_local, 0.0, GridFinish{nullptr, nullptr, -1.0}};
    const uint32_t grid = static_cast<uint32_t>(((n_paths + kPathsPerThread - 1) / kPathsPerThread + kBlock - 1) / kBlock);
    const uint64_t n_waves = static_cast<uint64_t>(grid) * (kBlock / 64);
    double *d_part = nullptr;
    uint64_t *d_st = nullptr;
    CK(hipMalloc(&d_part, 2 * sizeof(double) * grid));
    CK(hipMalloc(&d_st, 4 * sizeof(uint64_t) * n_waves));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // >= `seconds` of back-to-back launches on changing seeds, so the clock is the one held under sustained load
    const auto t_start = std::chrono::steady_clock::now();
    int launches = 0;
    float last_ms = 0.0f;
    std::vector<float> ms_all;
    while (true) {
        a.seed = 1234 + launches;
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((probe_kernel<T, PAIRSUM>), dim3(grid), dim3(kBlock), pad_bytes, 0, a, d_part, d_st);
        CK(hipEventRecord(e1, 0));
        ++launches;
        if (launches % 8 == 0) {
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&last_ms, e0, e1));
            ms_all.push_back(last_ms);
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
            if (el >= seconds) break;
        }
    }
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> st(4 * n_waves);
    std::vector<double> part(2 * grid);
    CK(hipMemcpy(st.data(), d_st, st.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    CK(hipMemcpy(part.data(), d_part, part.size() * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<double> ghz(n_waves), cyc(n_waves), pre_ticks(n_waves);
    uint64_t tmin = ~0ull, tmax = 0, wmin = ~0ull, wmax = 0;
    for (uint64_t w = 0; w < n_waves; ++w) {
        pre_ticks[w] = static_cast<double>(st[4 * w + 1] >> 48);
        st[4 * w + 1] &= (1ull << 48) - 1;
        const double dc = static_cast<double>(st[4 * w + 1] - st[4 * w + 0]);
        const double dr = static_cast<double>(st[4 * w + 3] - st[4 * w + 2]);
        ghz[w] = dr > 0 ? dc / dr * 0.1 : 0.0;
        cyc[w] = dc;
        wmin = std::min(wmin, st[4 * w + 2]);
        wmax = std::max(wmax, st[4 * w + 3]);
        (void)tmin;
        (void)tmax;
    }
    // census: waves inside their step loop at three instants of the launch, on the 100 MHz s_memrealtime clock (the
    // one counter all XCDs share; s_memtime is per-XCD)
    uint64_t c_lo = ~0ull, c_hi = 0;
    for (uint64_t w = 0; w < n_waves; ++w) {
        c_lo = std::min(c_lo, st[4 * w + 2]);
        c_hi = std::max(c_hi, st[4 * w + 3]);
    }
    double resident[3] = {0, 0, 0};
    for (int q = 0; q < 3; ++q) {
        const uint64_t t_q = c_lo + (c_hi - c_lo) * (q + 1) / 4;
        uint64_t alive = 0;
        for (uint64_t w = 0; w < n_waves; ++w) alive += (st[4 * w + 2] <= t_q && t_q < st[4 * w + 3]) ? 1 : 0;
        resident[q] = static_cast<double>(alive) / 1024.0;
    }
    std::sort(pre_ticks.begin(), pre_ticks.end());
    std::sort(ghz.begin(), ghz.end());
    std::sort(cyc.begin(), cyc.end());
    std::sort(ms_all.begin(), ms_all.end());
    const double clk = ghz[n_waves / 2];
    const double span_ms = static_cast<double>(wmax - wmin) / 1e5;  // 100 MHz ticks -> ms
    const double kernel_ms = ms_all[ms_all.size() / 2];
    // busy cycles of one SIMD over the launch / wave-iterations that SIMD ran (1024 SIMDs share the waves evenly)
    const double iters_per_wave = std::ceil(static_cast<double>(n_steps) / Normals<T>::kPerBlock);
    const double wave_iters_per_simd = static_cast<double>(n_waves) * iters_per_wave / 1024.0;
    const double cyc_per_wave_iter = span_ms * 1e-3 * clk * 1e9 / wave_iters_per_simd;
    double sum = 0.0;
    for (uint32_t b = 0; b < grid; ++b) sum += part[2 * b];
    std::printf("{\"probe\": \"%s\", \"lds_pad_bytes\": %u, \"paths\": %llu, \"steps\": %u, \"launches\": %d, \"kernel_ms_median\": %.4f, "
                "\"span_ms_last_launch\": %.4f, \"in_kernel_clock_ghz_median\": %.4f, \"clock_ghz_p05\": %.4f, "
                "\"clock_ghz_p95\": %.4f, \"wave_lifetime_cycles_median\": %.0f, \"paths_per_thread\": %d, \"path_steps_per_wave_iteration\": %d, "
                "\"simd_cycles_per_wave_iteration\": %.1f, \"waves_in_loop_per_simd\": [%.2f, %.2f, %.2f], "
                "\"us_before_loop_median\": %.2f, \"us_before_loop_p95\": %.2f, \"price\": %.6f}\n",
                name, pad_bytes, static_cast<unsigned long long>(n_paths), n_steps, launches, kernel_ms, span_ms, clk,
                ghz[n_waves / 20], ghz[n_waves - 1 - n_waves / 20], cyc[n_waves / 2], static_cast<int>(kPathsPerThread),
                static_cast<int>(kPathsPerThread) * Normals<T>::kPerBlock,
                cyc_per_wave_iter, resident[0], resident[1], resident[2], pre_ticks[n_waves / 2] * 0.01,
                pre_ticks[n_waves - 1 - n_waves / 20] * 0.01, std::exp(-0.1) * sum / static_cast<double>(n_paths));
    CK(hipFree(d_part));
    CK(hipFree(d_st));
    return 0;
}

int main(int argc, char **argv)
{
    const uint64_t n_paths = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 10000000ull;
    const uint32_t n_steps = argc > 2 ? static_cast<uint32_t>(std::atoi(argv[2])) : 252u;
    const double seconds = argc > 3 ? std::atof(argv[3]) : 2.5;
    if (run<double, true>("price_f64 (default: pair sums, log space)", n_paths, n_steps, seconds)) return 1;
    if (run<float, true>("price_f32 (default: pair sums, log space)", n_paths, n_steps, seconds)) return 1;
    if (run<double, false>("price_f64_product", n_paths, n_steps, seconds)) return 1;
    if (run<float, false>("price_f32_product", n_paths, n_steps, seconds)) return 1;
    if (argc > 4) {   // occupancy sweep: pad the workgroup's LDS so that fewer workgroups (= waves per SIMD) fit a CU
        // fp64 default loop: 106 VGPRs -> 4 workgroups per CU unpadded; LDS pads cap it further: 53 KB -> 3; 80 KB -> 2; 159 KB -> 1
        for (uint32_t pad : {33u << 10, 60u << 10, 139u << 10})
            if (run<double, true>("price_f64_occupancy", n_paths, n_steps, 0.6, pad)) return 1;
        // fp32: no tables -> 8 per CU unpadded
        for (uint32_t pad : {0u, 26u << 10, 39u << 10, 53u << 10, 80u << 10, 159u << 10})
            if (run<float, true>("price_f32_occupancy", n_paths, n_steps, 0.4, pad)) return 1;
    }
    return 0;
}
