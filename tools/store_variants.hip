// store_variants.hip — experiment harness for the trajectory-store path (BASELINE configs[2]: 100M paths x 252
// steps, fp32, step-major).  Times launch-shape / store-flavour variants of the shipped store loop (same device
// functions: Philox, Exponents<float>, PathState<float>) against the pure-store ceiling of the same access shape,
// so that a change to csrc/store.hip is chosen from measurements.  Not part of the product.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imonte-carlo-project-cuda_amd/csrc tools/store_variants.hip -o tools/store_variants
//   tools/store_variants [paths=100000000] [steps=252]     -> one JSON line per variant
#include "path_consts.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace mcamd;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

using f4 = float __attribute__((ext_vector_type(4)));

// G 16-byte groups per thread; group g of a thread sits G_STRIDE lanes-groups away so every store instruction of a
// wave still writes 1 KiB contiguous.  NT: non-temporal hint.  PRIO: raise the wave's priority around the stores.
// MINW: __launch_bounds__ minimum waves per SIMD (register budget).
// EVERY: store one row in EVERY (1 = all rows; 2 = half the bytes with the same arithmetic; 0 = no trajectory stores):
// separates "the arithmetic is slow" from "the arithmetic is slow WHILE the memory system is busy".
template <int G, bool NT, int PRIO, int MINW, int EVERY = 1>
__global__ __launch_bounds__(kBlock, MINW) void variant_kernel(StepConsts<float> c_arg, uint64_t seed, uint64_t n_local,
                                                               float *__restrict__ traj, float *__restrict__ payoffs,
                                                               double *__restrict__ partials,
                                                               uint64_t *__restrict__ stamps)
{
    constexpr int V = 4, NB = 4;
    const MathCtx<float> m = MathCtx<float>::init();
    const PhiloxKeys key = PhiloxKeys::make(seed);
    const StepConsts<float> c = resident(c_arg);
    // diagnostic stamps (this harness only): shader cycles and 100 MHz ticks around the wave's whole life
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t n_groups = n_local / V;                       // n_local % (V * G * 64) == 0 assumed by the harness
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock * G;
    const uint32_t n_full = c.n_sim / NB;
    double s = 0.0, s2 = 0.0;
    const int lane = threadIdx.x & 63;
    for (uint64_t w0 = (static_cast<uint64_t>(blockIdx.x) * kBlock + (threadIdx.x - lane)) * G; w0 < n_groups; w0 += stride) {
        float St[G][V];
        uint64_t base[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            base[g] = (w0 + static_cast<uint64_t>(g) * 64 + lane) * V;
#pragma unroll
            for (int p = 0; p < V; ++p) St[g][p] = c.S_start;
        }
        for (uint32_t k = 0; k < n_full; ++k) {
            Exponents<float> ex[G][V];
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int p = 0; p < V; ++p) ex[g][p].fill(m, c, key, base[g] + p, k);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                f4 pack[G];
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int p = 0; p < V; ++p) {
                        St[g][p] *= __builtin_amdgcn_exp2f(ex[g][p].x[j]);
                        pack[g][p] = St[g][p];
                    }
                const uint64_t row = static_cast<uint64_t>(k * NB + j) * n_local;
                if (EVERY == 0 || (EVERY > 1 && (j % EVERY) != 0)) continue;
                if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    f4 *dst = reinterpret_cast<f4 *>(traj + row + base[g]);
                    if (NT) __builtin_nontemporal_store(pack[g], dst);
                    else *dst = pack[g];
                }
                if (PRIO) __builtin_amdgcn_s_setprio(0);
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            f4 pay;
#pragma unroll
            for (int p = 0; p < V; ++p) {
                const float y = St[g][p] - c.K > 0.0f ? St[g][p] - c.K : 0.0f;
                pay[p] = y;
                s += static_cast<double>(y);
                s2 = __builtin_fma(static_cast<double>(y), static_cast<double>(y), s2);
            }
            if (payoffs) *reinterpret_cast<f4 *>(payoffs + base[g]) = pay;
        }
    }
    const uint32_t done = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(__double2hiint(s)));
    asm volatile("" ::"s"(done));
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    block_sum2<kBlock>(s, s2);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = s;
        partials[2 * blockIdx.x + 1] = s2;
    }
    if (stamps && lane == 0 && blockIdx.x % 16 == 0) {   // a sample of the waves is enough for a median
        const uint64_t w = (static_cast<uint64_t>(blockIdx.x) / 16) * (kBlock / 64) + threadIdx.x / 64;
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = w1 - w0;
    }
}

// pure stores of the same shape: the ceiling
template <bool NT>
__global__ __launch_bounds__(kBlock) void rows_kernel(float *out, uint64_t n_paths, uint32_t n_steps)
{
    const uint64_t groups = n_paths / 4, stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t g = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; g < groups; g += stride) {
        f4 v = {1.0f, 2.0f, 3.0f, static_cast<float>(g)};
        for (uint32_t s = 0; s < n_steps; ++s) {
            v.x += 1.0f;
            f4 *p = reinterpret_cast<f4 *>(out + static_cast<uint64_t>(s) * n_paths + g * 4);
            if (NT) __builtin_nontemporal_store(v, p);
            else *p = v;
        }
    }
}

struct Timer {
    hipEvent_t e0, e1;
};

template <typename F>
static int time_it(const char *name, double bytes, Timer &t, F launch, double *price_out = nullptr)
{
    float best = 1e30f, sum = 0.0f;
    const int reps = 6;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(t.e0, 0));
        launch();
        CK(hipEventRecord(t.e1, 0));
        CK(hipEventSynchronize(t.e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, t.e0, t.e1));
        if (r) {
            sum += ms;
            if (ms < best) best = ms;
        }
    }
    std::printf("{\"variant\": \"%s\", \"avg_ms\": %.3f, \"best_ms\": %.3f, \"avg_GBs\": %.1f, \"best_GBs\": %.1f%s", name,
                sum / (reps - 1), best, bytes / (sum / (reps - 1) * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9,
                price_out ? "" : "}\n");
    return 0;
}

int main(int argc, char **argv)
{
    const uint64_t n_paths = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 99999744ull;  // 512 x 195312: whole wavefronts for every variant (BASELINE configs[2] is 1e8)
    const uint32_t n_steps = argc > 2 ? static_cast<uint32_t>(std::atoi(argv[2])) : 252u;
    if (n_paths % 512 || n_steps % 4) {
        std::printf("paths must be a multiple of 512 and steps of 4\n");
        return 1;
    }
    PathJob j{};
    const double dt = 1.0 / n_steps;
    j.drift = (0.1 - 0.5 * 0.04) * dt;
    j.vol = 0.2 * std::sqrt(dt);
    j.K = 100.0;
    j.S_start = 100.0;
    j.n_sim = n_steps;
    j.n_steps = n_steps;
    j.n_local = n_paths;
    j.precision = 32;
    const StepConsts<float> c = make_consts<float>(j);
    float *traj = nullptr, *pay = nullptr;
    double *part = nullptr;
    CK(hipMalloc(&traj, n_paths * n_steps * sizeof(float)));
    CK(hipMalloc(&pay, n_paths * sizeof(float)));
    const uint32_t grid1 = static_cast<uint32_t>(n_paths / 4 / kBlock + 1);
    CK(hipMalloc(&part, 2 * sizeof(double) * grid1));
    Timer t;
    CK(hipEventCreate(&t.e0));
    CK(hipEventCreate(&t.e1));
    const double bytes = static_cast<double>(n_paths) * n_steps * 4.0 + static_cast<double>(n_paths) * 4.0;
    uint64_t *stamps = nullptr;
    const size_t n_stamp_waves = (static_cast<size_t>(grid1) / 16 + 1) * (kBlock / 64);
    CK(hipMalloc(&stamps, 2 * sizeof(uint64_t) * n_stamp_waves));
    auto clock_ghz = [&](uint32_t grid) {   // median in-kernel clock of the last launch: d(memtime) / d(memrealtime) x 100 MHz
        const size_t n = (static_cast<size_t>(grid) / 16) * (kBlock / 64);
        std::vector<uint64_t> h(2 * n);
        if (n == 0) return 0.0;
        (void)hipMemcpy(h.data(), stamps, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost);
        std::vector<double> g(n);
        for (size_t i = 0; i < n; ++i) g[i] = h[2 * i + 1] ? 0.1 * static_cast<double>(h[2 * i]) / static_cast<double>(h[2 * i + 1]) : 0.0;
        std::sort(g.begin(), g.end());
        return g[n / 2];
    };
    auto price = [&](uint32_t grid) {
        std::vector<double> h(2 * grid);
        (void)hipMemcpy(h.data(), part, h.size() * sizeof(double), hipMemcpyDeviceToHost);
        double s = 0.0;
        for (uint32_t b = 0; b < grid; ++b) s += h[2 * b];
        return std::exp(-0.1) * s / static_cast<double>(n_paths);
    };
#define RUN_E(NAME, G, NT, PRIO, MINW, GRID, EVERY, BYTES)                                                              \
    do {                                                                                                                \
        const uint32_t grid_ = (GRID);                                                                                  \
        double dummy;                                                                                                   \
        if (time_it(NAME, BYTES, t, [&] {                                                                               \
                hipLaunchKernelGGL((variant_kernel<G, NT, PRIO, MINW, EVERY>), dim3(grid_), dim3(kBlock), 0, 0, c,       \
                                   1234ull, n_paths, traj, pay, part, stamps);                                          \
            }, &dummy)) return 1;                                                                                       \
        std::printf(", \"grid\": %u, \"price\": %.6f, \"in_kernel_clock_ghz\": %.3f}\n", grid_, price(grid_),         \
                    clock_ghz(grid_));                                                                                  \
    } while (0)
#define RUN(NAME, G, NT, PRIO, MINW, GRID) RUN_E(NAME, G, NT, PRIO, MINW, GRID, 1, bytes)
    const uint32_t full1 = static_cast<uint32_t>(n_paths / 4 / kBlock);        // one 16-byte group per thread
    const uint32_t full2 = static_cast<uint32_t>(n_paths / 8 / kBlock);        // two groups per thread
    if (time_it("pure_stores_nt", bytes, t, [&] { hipLaunchKernelGGL(rows_kernel<true>, dim3(full1), dim3(kBlock), 0, 0, traj, n_paths, n_steps); })) return 1;
    if (time_it("pure_stores_plain", bytes, t, [&] { hipLaunchKernelGGL(rows_kernel<false>, dim3(full1), dim3(kBlock), 0, 0, traj, n_paths, n_steps); })) return 1;
    RUN("g1_nt_prio0_w1 (shipped shape)", 1, true, 0, 1, full1);
    RUN("g1_plain_prio0_w1", 1, false, 0, 1, full1);
    RUN("g1_nt_prio1_w1", 1, true, 1, 1, full1);
    RUN("g1_plain_prio1_w1", 1, false, 1, 1, full1);
    RUN("g1_nt_prio0_w8", 1, true, 0, 8, full1);
    RUN("g1_plain_prio0_w8", 1, false, 0, 8, full1);
    RUN("g1_nt_prio0_w4", 1, true, 0, 4, full1);
    RUN("g2_nt_prio0_w1", 2, true, 0, 1, full2);
    RUN("g2_plain_prio0_w1", 2, false, 0, 1, full2);
    RUN("g2_plain_prio0_w4", 2, false, 0, 4, full2);
    // coupling experiment: the same arithmetic with half / none of the rows stored
    RUN_E("g1_nt_half_the_rows_stored", 1, true, 0, 1, full1, 2, bytes / 2);
    RUN_E("g1_no_rows_stored", 1, true, 0, 1, full1, 0, static_cast<double>(n_paths) * 4.0);
    RUN("g1_nt_prio0_w1 (again)", 1, true, 0, 1, full1);
    RUN("g1_nt_persistent_2048", 1, true, 0, 1, 2048);
    RUN("g1_plain_persistent_4096", 1, false, 0, 1, 4096);
    return 0;
}
