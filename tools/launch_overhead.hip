// launch_overhead.hip — DIAGNOSTIC: where do the ~15 us between a small pricing kernel's HIP-event time and the
// wall time of the whole synchronous call go?  Times, on the host, an empty kernel (a) launched and waited for with
// hipStreamSynchronize, (b) the same between two hipEventRecord calls, (c) plus hipEventElapsedTime, (d) waited for
// by spinning on a word the kernel writes into pinned host memory instead of hipStreamSynchronize.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/launch_overhead.hip -o tools/launch_overhead
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void flag_kernel(volatile unsigned long long *flag, unsigned long long epoch, int spin)
{
    // ~spin x 1 us of device time, then publish
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(127);
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        *flag = epoch;
        __threadfence_system();
    }
}

template <typename F>
static double median_us(int reps, F f)
{
    std::vector<double> t(reps);
    for (int i = 0; i < reps; ++i) {
        const auto a = std::chrono::steady_clock::now();
        f(i);
        t[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
    }
    std::sort(t.begin(), t.end());
    return t[reps / 2];
}

int main()
{
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    unsigned long long *h = nullptr, *d = nullptr;
    hipHostMalloc(&h, 64, hipHostMallocDefault);
    hipHostGetDevicePointer(reinterpret_cast<void **>(&d), h, 0);
    *h = 0;
    unsigned long long epoch = 0;
    for (int spin : {0, 100}) {
        const double a = median_us(400, [&](int) {
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, d, ++epoch, spin);
            hipStreamSynchronize(s);
        });
        const double b = median_us(400, [&](int) {
            hipEventRecord(e0, s);
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, d, ++epoch, spin);
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
        });
        float ms = 0.0f;
        const double c = median_us(400, [&](int) {
            hipEventRecord(e0, s);
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, d, ++epoch, spin);
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            hipEventElapsedTime(&ms, e0, e1);
        });
        const double dd = median_us(400, [&](int) {
            hipEventRecord(e0, s);
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, d, ++epoch, spin);
            hipEventRecord(e1, s);
            while (*reinterpret_cast<volatile unsigned long long *>(h) != epoch) __builtin_ia32_pause();
        });
        const double ee = median_us(400, [&](int) {
            hipEventRecord(e0, s);
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, d, ++epoch, spin);
            hipEventRecord(e1, s);
            while (*reinterpret_cast<volatile unsigned long long *>(h) != epoch) __builtin_ia32_pause();
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        });
        std::printf("{\"device_sleep_iterations\": %d, \"kernel_event_us\": %.2f, \"launch_sync_us\": %.2f, \"with_two_event_records_us\": %.2f, "
                    "\"plus_elapsed_time_us\": %.2f, \"spin_on_pinned_flag_us\": %.2f, \"spin_then_event_sync_and_elapsed_us\": %.2f}\n",
                    spin, ms * 1e3, a, b, c, dd, ee);
    }
    return 0;
}
