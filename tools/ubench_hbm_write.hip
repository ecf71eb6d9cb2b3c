// ubench_hbm_write.hip — measures the streaming-write ceiling of the GPU it runs on for the store path's
// access shape: 100.8 GB written once as 252 rows of 1e8 floats, each lane writing 16 bytes per row
// (step-major trajectory layout), with and without the non-temporal hint, against a flat memset-like fill.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

using f4 = float __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void rows_kernel(float *out, uint64_t n_paths, uint32_t n_steps)
{
    const uint64_t groups = n_paths / 4, stride = (uint64_t)gridDim.x * 256;
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += stride) {
        f4 v = {1.0f, 2.0f, 3.0f, (float)g};
        for (uint32_t s = 0; s < n_steps; ++s) {
            v.x += 1.0f;
            f4 *p = reinterpret_cast<f4 *>(out + (uint64_t)s * n_paths + g * 4);
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void flat_kernel(float *out, uint64_t n)
{
    const uint64_t groups = n / 4, stride = (uint64_t)gridDim.x * 256;
    f4 v = {1.0f, 2.0f, 3.0f, 4.0f};
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += stride) {
        f4 *p = reinterpret_cast<f4 *>(out) + g;
        if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
}

int main()
{
    const uint64_t n_paths = 100000000ull; const uint32_t n_steps = 252;
    const uint64_t n = n_paths * n_steps;
    float *buf; CK(hipMalloc(&buf, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch, const char *name) -> int {
        float best = 1e30f, sum = 0;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) { sum += ms; if (ms < best) best = ms; }
        }
        printf("{\"kernel\": \"%s\", \"avg_ms\": %.3f, \"best_ms\": %.3f, \"avg_GBs\": %.1f, \"best_GBs\": %.1f}\n", name, sum / 3, best,
               n * 4.0 / (sum / 3 * 1e-3) / 1e9, n * 4.0 / (best * 1e-3) / 1e9);
        return 0;
    };
    const uint32_t g1 = (uint32_t)((n_paths / 4 + 255) / 256);
    time([&] { hipLaunchKernelGGL(rows_kernel<true>, dim3(g1), dim3(256), 0, 0, buf, n_paths, n_steps); }, "rows_16B_nt_grid97657");
    time([&] { hipLaunchKernelGGL(rows_kernel<false>, dim3(g1), dim3(256), 0, 0, buf, n_paths, n_steps); }, "rows_16B_plain_grid97657");
    time([&] { hipLaunchKernelGGL(rows_kernel<true>, dim3(2048), dim3(256), 0, 0, buf, n_paths, n_steps); }, "rows_16B_nt_grid2048");
    time([&] { hipLaunchKernelGGL(rows_kernel<true>, dim3(8192), dim3(256), 0, 0, buf, n_paths, n_steps); }, "rows_16B_nt_grid8192");
    time([&] { hipLaunchKernelGGL(flat_kernel<true>, dim3(2048 * 8), dim3(256), 0, 0, buf, n); }, "flat_16B_nt");
    time([&] { hipLaunchKernelGGL(flat_kernel<false>, dim3(2048 * 8), dim3(256), 0, 0, buf, n); }, "flat_16B_plain");
    time([&] { (void)hipMemsetAsync(buf, 0, n * 4, 0); }, "hipMemsetAsync");
    return 0;
}
