// naive_port_baseline.hip — NOT part of the engine.  A measuring stick: the reference's design carried to this GPU
// the obvious way — one RNG state per thread in HBM initialised by a setup kernel (rocRAND XORWOW, the analogue of the
// curandState / curand_init(seed, tid, 0) of inc/tool.cuh:192-195), library normals, library exp, a 1024-thread block
// with a shared-memory tree and one atomicAdd per block (the structure of inc/trajectories.cuh:115-271, written from
// its description in SURVEY.md, with the window test removed = European call).  Prints the time for 10M paths x 252
// steps in fp32 (the reference's precision) and fp64 (BASELINE configs[1]) so bench.py's numbers have a same-GPU
// "straight port" figure beside them.   hipcc --offload-arch=gfx950 -O3 tools/naive_port_baseline.hip -o tools/naive_port_baseline
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <cstdio>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void setup_states(rocrand_state_xorwow *st, unsigned long long seed, int n)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < n) rocrand_init(seed, tid, 0, &st[tid]);
}

template <typename T> __device__ T draw(rocrand_state_xorwow *s);
template <> __device__ float draw<float>(rocrand_state_xorwow *s) { return rocrand_normal(s); }
template <> __device__ double draw<double>(rocrand_state_xorwow *s) { return rocrand_normal_double(s); }

template <typename T>
__global__ void price_naive(T *out, rocrand_state_xorwow *states, int n_paths, int n_steps, T S0, T K, T drift, T vol)
{
    __shared__ T sdata[1024];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, tid = threadIdx.x;
    T pay = 0;
    if (idx < n_paths) {
        rocrand_state_xorwow st = states[idx];
        T St = S0;
        for (int i = 0; i < n_steps; ++i) St *= exp(drift + vol * draw<T>(&st));
        pay = St > K ? St - K : T(0);
    }
    sdata[tid] = pay;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (tid < s) sdata[tid] += sdata[tid + s];
        __syncthreads();
    }
    if (tid == 0) atomicAdd(out, sdata[0]);
}

template <typename T>
int run(const char *name, int n_paths, int n_steps)
{
    const int tpb = 1024, blocks = (n_paths + tpb - 1) / tpb;
    rocrand_state_xorwow *st; T *d_out;
    CK(hipMalloc(&st, sizeof(rocrand_state_xorwow) * (size_t)blocks * tpb));
    CK(hipMalloc(&d_out, sizeof(T)));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const double dt = 1.0 / n_steps;
    float best_setup = 1e30f, best_price = 1e30f; T h = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(d_out, 0, sizeof(T)));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(setup_states, dim3(blocks), dim3(tpb), 0, 0, st, 1234ULL, n_paths);
        CK(hipEventRecord(e1));
        hipLaunchKernelGGL(price_naive<T>, dim3(blocks), dim3(tpb), 0, 0, d_out, st, n_paths, n_steps, (T)100, (T)100,
                           (T)((0.1 - 0.02) * dt), (T)(0.2 * std::sqrt(dt)));
        CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
        float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
        if (rep) { if (a < best_setup) best_setup = a; if (b < best_price) best_price = b; }
        CK(hipMemcpy(&h, d_out, sizeof(T), hipMemcpyDeviceToHost));
    }
    printf("{\"kernel\": \"%s\", \"paths\": %d, \"steps\": %d, \"setup_ms\": %.3f, \"price_ms\": %.3f, \"total_ms\": %.3f, "
           "\"paths_per_s\": %.4g, \"price\": %.5f, \"state_bytes\": %zu}\n", name, n_paths, n_steps, best_setup, best_price,
           best_setup + best_price, n_paths / ((best_setup + best_price) * 1e-3), std::exp(-0.1) * (double)h / n_paths,
           sizeof(rocrand_state_xorwow) * (size_t)blocks * tpb);
    CK(hipFree(st)); CK(hipFree(d_out));
    return 0;
}

int main()
{
    if (run<float>("naive_port_f32", 10000000, 252)) return 1;
    if (run<double>("naive_port_f64", 10000000, 252)) return 1;
    return 0;
}
