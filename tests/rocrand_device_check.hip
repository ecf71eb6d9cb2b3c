// rocrand_device_check.hip — test helper (built by tests/test_gpu_rocrand.py with hipcc): fills device buffers
// with rocRAND's OWN device API (rocrand_init(seed, subsequence, 0) + rocrand_normal4 / rocrand_normal_double2 /
// rocrand) so the engine's hand-written Philox + Box-Muller can be compared with the library it claims to
// reproduce, on the same GPU.  Not part of the product.
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <stdint.h>

__global__ void k_normal4(uint64_t seed, uint64_t sub0, uint64_t n_sub, uint32_t blocks, float *out)
{
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n_sub) return;
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, sub0 + i, 0, &st);
    for (uint32_t b = 0; b < blocks; ++b) {
        const float4 z = rocrand_normal4(&st);
        float *o = out + (i * blocks + b) * 4;
        o[0] = z.x; o[1] = z.y; o[2] = z.z; o[3] = z.w;
    }
}

__global__ void k_normal_double2(uint64_t seed, uint64_t sub0, uint64_t n_sub, uint32_t blocks, double *out)
{
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n_sub) return;
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, sub0 + i, 0, &st);
    for (uint32_t b = 0; b < blocks; ++b) {
        const double2 z = rocrand_normal_double2(&st);
        out[(i * blocks + b) * 2] = z.x;
        out[(i * blocks + b) * 2 + 1] = z.y;
    }
}

extern "C" int rr_normal4(uint64_t seed, uint64_t sub0, uint64_t n_sub, uint32_t blocks, float *d_out)
{
    hipLaunchKernelGGL(k_normal4, dim3((n_sub + 255) / 256), dim3(256), 0, 0, seed, sub0, n_sub, blocks, d_out);
    return (int)hipDeviceSynchronize();
}

extern "C" int rr_normal_double2(uint64_t seed, uint64_t sub0, uint64_t n_sub, uint32_t blocks, double *d_out)
{
    hipLaunchKernelGGL(k_normal_double2, dim3((n_sub + 255) / 256), dim3(256), 0, 0, seed, sub0, n_sub, blocks, d_out);
    return (int)hipDeviceSynchronize();
}
