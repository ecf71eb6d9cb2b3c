// store_variants.hip — experiment harness for the trajectory-store path (BASELINE configs[2]: 100M paths x 252
// steps, fp32, step-major).  Times launch-shape / store-flavour variants of the shipped store loop (same device
// functions: Philox, Exponents<float>, PathState<float>) against the pure-store ceiling of the same access shape,
// so that a change to csrc/store.hip is chosen from measurements.  Not part of the product.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imonte-carlo-project-cuda_amd/csrc tools/store_variants.hip -o tools/store_variants
//   tools/store_variants [paths=100000000] [steps=252]     -> one JSON line per variant
#include "path_consts.hpp"

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
template <int G, bool NT, int PRIO, int MINW>
__global__ __launch_bounds__(kBlock, MINW) void variant_kernel(StepConsts<float> c, uint64_t seed, uint64_t n_local,
                                                               float *__restrict__ traj, float *__restrict__ payoffs,
                                                               double *__restrict__ partials)
{
    constexpr int V = 4, NB = 4;
    const MathCtx<float> m = MathCtx<float>::init();
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
                for (int p = 0; p < V; ++p) ex[g][p].fill(m, c, seed, base[g] + p, k);
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
    block_sum2<kBlock>(s, s2);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = s;
        partials[2 * blockIdx.x + 1] = s2;
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
    auto price = [&](uint32_t grid) {
        std::vector<double> h(2 * grid);
        (void)hipMemcpy(h.data(), part, h.size() * sizeof(double), hipMemcpyDeviceToHost);
        double s = 0.0;
        for (uint32_t b = 0; b < grid; ++b) s += h[2 * b];
        return std::exp(-0.1) * s / static_cast<double>(n_paths);
    };
#define RUN(NAME, G, NT, PRIO, MINW, GRID)                                                                              \
    do {                                                                                                                \
        const uint32_t grid_ = (GRID);                                                                                  \
        double dummy;                                                                                                   \
        if (time_it(NAME, bytes, t, [&] {                                                                               \
                hipLaunchKernelGGL((variant_kernel<G, NT, PRIO, MINW>), dim3(grid_), dim3(kBlock), 0, 0, c, 1234ull,     \
                                   n_paths, traj, pay, part);                                                           \
            }, &dummy)) return 1;                                                                                       \
        std::printf(", \"grid\": %u, \"price\": %.6f}\n", grid_, price(grid_));                                         \
    } while (0)
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
    RUN("g1_nt_persistent_2048", 1, true, 0, 1, 2048);
    RUN("g1_plain_persistent_4096", 1, false, 0, 1, 4096);
    return 0;
}
