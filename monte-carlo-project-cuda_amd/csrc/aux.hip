// aux.hip — the kernels beside the two pricing paths: array-driven pricer, bulk normal fill,
// stand-alone sum reductions.  gfx950 only.
#include "path_consts.hpp"

#include "mcamd.h"

namespace mcamd {

// ---------------------------------------------------------------------------------------------
// Array-driven European pricer: normals are an input, d_normals[path * n_steps + step] (the
// reference's layout).  Replaces simulateOptionPriceGPU / simulateOptionPriceMultipleBlockGPU
// (array overloads), inc/trajectories.cuh:14-52; this is the deterministic parity path (its CPU
// twin is inc/testing.cuh:75-91).  A wavefront stages a 64-path x 128-byte tile through LDS:
// global reads are row-contiguous (one full 128 B line per path row), LDS reads are column-wise
// (lane = path); rows are padded by one 16-byte vector so that 16-byte reads of consecutive rows fall on
// different banks.
// ---------------------------------------------------------------------------------------------
template <typename T>
struct ArrayArgs {
    StepConsts<T> c;
    uint64_t n_local;
    const T *normals;
    T *payoffs;
};

// VEC: every path row starts 16-byte aligned (n_sim a multiple of 16 / sizeof(T), aligned buffer): a lane
// fetches 16 bytes, so one wave-wide load covers 8 path rows x 128 bytes instead of 2.
template <typename T, bool WINDOW, bool VEC>
__global__ __launch_bounds__(kBlock) void from_normals_kernel(ArrayArgs<T> a, double *__restrict__ partials)
{
    constexpr int kWaves = kBlock / kWave;
    constexpr int TS = 128 / sizeof(T);        // tile width in steps: one 128 B line per path row
    constexpr int kRowsPerLoad = kWave / TS;   // path rows one wave-wide load covers
    constexpr int kPad = 16 / sizeof(T);       // one 16-byte vector of padding: rows stay 16-byte aligned
    __shared__ alignas(16) T tile[kWaves][kWave][TS + kPad];
    const StepConsts<T> &c = a.c;
    const MathCtx<T> m = MathCtx<T>::init();
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int lcol = lane % TS, lrow = lane / TS;
    const uint64_t n_tiles = (a.n_local + kWave - 1) / kWave;  // one tile row = 64 paths
    const uint64_t wave_stride = static_cast<uint64_t>(gridDim.x) * kWaves;
    double s = 0.0, s2 = 0.0;

    // VEC staging geometry: V elements (16 bytes) per lane per load, 8 lanes per path row, 8 rows per load
    constexpr int V = 16 / sizeof(T);
    constexpr int kLanesPerRow = TS / V;
    constexpr int kRowsPerVecLoad = kWave / kLanesPerRow;
    constexpr int kVecLoads = kWave / kRowsPerVecLoad;  // wave-wide loads per 64-path x TS-step tile
    using VT = T __attribute__((ext_vector_type(V)));
    const int vrow = lane / kLanesPerRow, vcol = (lane % kLanesPerRow) * V;

    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * kWaves + wave; t < n_tiles; t += wave_stride) {
        const uint64_t path0 = t * kWave;
        const uint64_t my_path = path0 + lane;
        PathState<T> ps = PathState<T>::start(c.S_start);
        int32_t count = c.Ik;
        // VEC: the next tile's global loads are issued into registers before the current tile is consumed, so a
        // wave always has loads in flight underneath its own (serially dependent) step loop
        VT pre[kVecLoads];
        auto prefetch = [&](uint32_t s0) {
#pragma unroll
            for (int i = 0; i < kVecLoads; ++i) {
                const uint64_t p = path0 + i * kRowsPerVecLoad + vrow;
                const bool ok = p < a.n_local && s0 + static_cast<uint32_t>(vcol) < c.n_sim;
                const uint64_t off = ok ? p * c.n_sim + s0 + vcol : 0;   // lanes past the end re-read element 0
                pre[i] = *reinterpret_cast<const VT *>(a.normals + off);
            }
        };
        if (VEC) prefetch(0);
        for (uint32_t s0 = 0; s0 < c.n_sim; s0 += TS) {
            const uint32_t n_cols = (c.n_sim - s0 < static_cast<uint32_t>(TS)) ? c.n_sim - s0 : TS;
            // stage: global reads are contiguous along a path's row, LDS holds [path][step]
            if (VEC) {
#pragma unroll
                for (int i = 0; i < kVecLoads; ++i)   // one ds_write_b128 per staged load
                    *reinterpret_cast<VT *>(&tile[wave][i * kRowsPerVecLoad + vrow][vcol]) = pre[i];
            } else {
                for (int r = 0; r < kWave; r += kRowsPerLoad) {
                    const uint64_t p = path0 + r + lrow;
                    if (p < a.n_local && static_cast<uint32_t>(lcol) < n_cols)
                        tile[wave][r + lrow][lcol] = a.normals[p * c.n_sim + s0 + lcol];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (VEC && s0 + TS < c.n_sim) prefetch(s0 + TS);
            if (VEC) {
                // the lane's TS normals come back as TS / V 16-byte LDS reads; the step loop is fully unrolled
                VT zz[TS / V];
#pragma unroll
                for (int i = 0; i < TS / V; ++i) zz[i] = *reinterpret_cast<const VT *>(&tile[wave][lane][i * V]);
                if (my_path < a.n_local) {
#pragma unroll
                    for (int i = 0; i < TS / V; ++i)
#pragma unroll
                        for (int k = 0; k < V; ++k)
                            if (static_cast<uint32_t>(i * V + k) < n_cols) {
                                ps.step(__builtin_fma(zz[i][k], c.vol, c.drift), m);
                                if (WINDOW) count += (c.B > ps.value(m)) ? 1 : 0;
                            }
                }
            } else if (my_path < a.n_local) {
                for (uint32_t j = 0; j < n_cols; ++j) {
                    ps.step(__builtin_fma(tile[wave][lane][j], c.vol, c.drift), m);
                    if (WINDOW) count += (c.B > ps.value(m)) ? 1 : 0;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (my_path < a.n_local) {
            const T pay = payoff<T, WINDOW>(ps.value(m), count, c);
            if (a.payoffs) a.payoffs[my_path] = pay;
            const double pd = static_cast<double>(pay);
            s += pd;
            s2 = __builtin_fma(pd, pd, s2);
        }
    }
    block_sum2<kBlock>(s, s2);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = s;
        partials[2 * blockIdx.x + 1] = s2;
    }
}

uint32_t array_grid(uint64_t n_local)
{
    return clamp_grid((n_local + kBlock - 1) / kBlock);
}

template <typename T>
static hipError_t launch_from_normals_t(const PathJob &j, const void *d_normals, void *d_payoffs, double *d_partials,
                                        uint32_t grid, hipStream_t stream)
{
    ArrayArgs<T> a{make_consts<T>(j), j.n_local, static_cast<const T *>(d_normals), static_cast<T *>(d_payoffs)};
    const bool vec = (j.n_sim % (16 / sizeof(T)) == 0) && (reinterpret_cast<uintptr_t>(d_normals) % 16 == 0);
    const dim3 g(grid), b(kBlock);
    if (j.window) {
        if (vec) hipLaunchKernelGGL((from_normals_kernel<T, true, true>), g, b, 0, stream, a, d_partials);
        else hipLaunchKernelGGL((from_normals_kernel<T, true, false>), g, b, 0, stream, a, d_partials);
    } else {
        if (vec) hipLaunchKernelGGL((from_normals_kernel<T, false, true>), g, b, 0, stream, a, d_partials);
        else hipLaunchKernelGGL((from_normals_kernel<T, false, false>), g, b, 0, stream, a, d_partials);
    }
    return hipGetLastError();
}

hipError_t launch_from_normals(const PathJob &j, const void *d_normals, void *d_payoffs, double *d_partials,
                               uint32_t grid, hipStream_t stream)
{
    return j.precision == 32 ? launch_from_normals_t<float>(j, d_normals, d_payoffs, d_partials, grid, stream)
                             : launch_from_normals_t<double>(j, d_normals, d_payoffs, d_partials, grid, stream);
}

// ---------------------------------------------------------------------------------------------
// Bulk normal fill: out[NB k .. NB k + NB) = normals of Philox block k of subsequence 0.
// Replaces curandGenerateNormal (inc/testing.cuh:17-24).  One 16 B store per lane per block:
// pure HBM-write-bound, 4 B (fp32) / 8 B (fp64) per element, no reads.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void normals_kernel(uint64_t seed, uint64_t n, T *__restrict__ out, bool vec_ok)
{
    constexpr int NB = Normals<T>::kPerBlock;
    const MathCtx<T> m = MathCtx<T>::init();
    const uint64_t n_blocks = (n + NB - 1) / NB;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t k = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; k < n_blocks; k += stride) {
        Normals<T> nrm;
        nrm.fill(m, seed, 0, k);
        const uint64_t base = k * NB;
        if (vec_ok && base + NB <= n) {
            using VT = T __attribute__((ext_vector_type(NB)));
            VT pack;
#pragma unroll
            for (int j = 0; j < NB; ++j) pack[j] = nrm.z[j];
            __builtin_nontemporal_store(pack, reinterpret_cast<VT *>(out + base));
        } else {
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (base + j < n) out[base + j] = nrm.z[j];
        }
    }
}

hipError_t launch_generate_normals(uint64_t seed, uint64_t n, int precision, void *d_out, hipStream_t stream)
{
    const uint64_t nb = precision == 32 ? 4 : 2;
    const uint32_t grid = clamp_grid(((n + nb - 1) / nb + kBlock - 1) / kBlock);
    const bool vec_ok = reinterpret_cast<uintptr_t>(d_out) % 16 == 0;
    if (precision == 32)
        hipLaunchKernelGGL(normals_kernel<float>, dim3(grid), dim3(kBlock), 0, stream, seed, n,
                           static_cast<float *>(d_out), vec_ok);
    else
        hipLaunchKernelGGL(normals_kernel<double>, dim3(grid), dim3(kBlock), 0, stream, seed, n,
                           static_cast<double *>(d_out), vec_ok);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Stand-alone sum reductions.  The reference keeps four teaching variants of the NVIDIA SDK
// reduction (inc/reduce.cuh:9-227: reduce3 sequential addressing + first add on load, reduce4
// shuffle for the last warp, reduce5 fully unrolled, reduce6 grid-stride multi-element) and
// lets the caller pick one (ReductionType, inc/testing.cuh:100-106).  The same choice is offered
// here, re-thought for wave64; every variant accumulates in fp64 and leaves one partial per
// block, finished by final_reduce_kernel:
//   3  LDS tree with sequential addressing down to one element (barrier per level)
//   4  LDS tree down to one wave, then wave64 shuffles
//   5  wave64 shuffles first, one LDS slot per wave, first wave finishes (no tree at all)
//   6  grid-stride, 16 B loads per lane, then as 5 — the production reduce
// Variants 3-5 consume 2 elements per thread (first add on load), so their grid grows with n.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ double load2(const T *__restrict__ in, uint64_t n, uint64_t i)
{
    double v = i < n ? static_cast<double>(in[i]) : 0.0;
    if (i + kBlock < n) v += static_cast<double>(in[i + kBlock]);
    return v;
}

template <typename T, int VARIANT>
__global__ __launch_bounds__(kBlock) void reduce_kernel(const T *__restrict__ in, uint64_t n,
                                                        double *__restrict__ partials)
{
    __shared__ double sdata[kBlock];
    const int tid = threadIdx.x;
    double v = 0.0, zero = 0.0;
    if (VARIANT == MCAMD_REDUCE_GRID_STRIDE) {
        constexpr int V = 16 / sizeof(T);
        using VT = T __attribute__((ext_vector_type(V)));
        const uint64_t n_vec = (reinterpret_cast<uintptr_t>(in) % 16 == 0) ? n / V : 0;
        const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
        const uint64_t gtid = static_cast<uint64_t>(blockIdx.x) * kBlock + tid;
        const VT *vin = reinterpret_cast<const VT *>(in);
        for (uint64_t i = gtid; i < n_vec; i += stride) {
            const VT x = vin[i];
#pragma unroll
            for (int j = 0; j < V; ++j) v += static_cast<double>(x[j]);
        }
        for (uint64_t i = n_vec * V + gtid; i < n; i += stride) v += static_cast<double>(in[i]);
        block_sum2<kBlock>(v, zero);
    } else {
        // first add on load: two elements per thread per chunk; the grid is capped at kMaxGrid blocks, so
        // for very large n a block walks several chunks before its tree
        const uint64_t chunk_stride = static_cast<uint64_t>(gridDim.x) * (2 * kBlock);
        for (uint64_t base = static_cast<uint64_t>(blockIdx.x) * (2 * kBlock); base < n; base += chunk_stride)
            v += load2(in, n, base + tid);
        if (VARIANT == MCAMD_REDUCE_SEQUENTIAL) {
            sdata[tid] = v;
            __syncthreads();
            for (int sft = kBlock / 2; sft > 0; sft >>= 1) {
                if (tid < sft) sdata[tid] = v = v + sdata[tid + sft];
                __syncthreads();
            }
        } else if (VARIANT == MCAMD_REDUCE_FIRST_ADD) {
            sdata[tid] = v;
            __syncthreads();
            for (int sft = kBlock / 2; sft >= kWave; sft >>= 1) {
                if (tid < sft) sdata[tid] = v = v + sdata[tid + sft];
                __syncthreads();
            }
            if (tid < kWave) v = wave_sum(v);
        } else {
            block_sum2<kBlock>(v, zero);
        }
    }
    if (tid == 0) {
        partials[2 * blockIdx.x] = v;
        partials[2 * blockIdx.x + 1] = 0.0;
    }
}

// Final pass: sums n_records records of N doubles with ONE workgroup in a fixed order -> deterministic for
// a given launch shape.  1024 threads, four independent accumulator sets per thread so the L2 latency
// of the single workgroup is overlapped.
constexpr int kFinalBlock = 1024;

// n_value >= 0 selects the asynchronous "stats" layout: out[0..5) = {sum, sumsq, sum_c, sum_cc, sum_yc}
// (zeros where N = 2) and out[5] = n_value, so one all-reduce of 6 doubles carries a whole shard.
template <int N>
__global__ __launch_bounds__(kFinalBlock) void final_reduce_kernel(const double *__restrict__ partials,
                                                                  uint32_t n_records, double *__restrict__ out,
                                                                  double n_value)
{
    double s[4][N];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < N; ++k) s[u][k] = 0.0;
    uint32_t i = threadIdx.x;
    for (; i + 3 * kFinalBlock < n_records; i += 4 * kFinalBlock) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < N; ++k) s[u][k] += partials[static_cast<uint64_t>(i + u * kFinalBlock) * N + k];
    }
    for (; i < n_records; i += kFinalBlock)
#pragma unroll
        for (int k = 0; k < N; ++k) s[0][k] += partials[static_cast<uint64_t>(i) * N + k];
    double v[N];
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = (s[0][k] + s[1][k]) + (s[2][k] + s[3][k]);
    block_sumN<kFinalBlock, N>(v);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) out[k] = v[k];
        if (n_value >= 0.0) {
#pragma unroll
            for (int k = N; k < 5; ++k) out[k] = 0.0;
            out[5] = n_value;
        }
    }
}

hipError_t launch_final_reduce(const double *d_partials, uint32_t n_records, int record_doubles, double *d_out,
                               hipStream_t stream, double n_value)
{
    if (record_doubles == 5)
        hipLaunchKernelGGL(final_reduce_kernel<5>, dim3(1), dim3(kFinalBlock), 0, stream, d_partials, n_records, d_out,
                           n_value);
    else if (record_doubles == 3)
        hipLaunchKernelGGL(final_reduce_kernel<3>, dim3(1), dim3(kFinalBlock), 0, stream, d_partials, n_records, d_out,
                           n_value);
    else
        hipLaunchKernelGGL(final_reduce_kernel<2>, dim3(1), dim3(kFinalBlock), 0, stream, d_partials, n_records, d_out,
                           n_value);
    return hipGetLastError();
}

uint32_t reduce_grid(uint64_t n, int variant)
{
    if (variant == MCAMD_REDUCE_GRID_STRIDE) {
        // memory-bound: 256 CUs x 8 blocks, grid-stride the rest
        const uint64_t want = (n + kBlock * 4 - 1) / (kBlock * 4);
        return static_cast<uint32_t>(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    }
    return clamp_grid((n + 2 * kBlock - 1) / (2 * kBlock));
}

template <typename T>
static hipError_t launch_reduce_t(const void *d_in, uint64_t n, int variant, double *d_partials, uint32_t grid,
                                  hipStream_t stream)
{
    const T *in = static_cast<const T *>(d_in);
    const dim3 g(grid), b(kBlock);
    switch (variant) {
        case MCAMD_REDUCE_SEQUENTIAL:
            hipLaunchKernelGGL((reduce_kernel<T, MCAMD_REDUCE_SEQUENTIAL>), g, b, 0, stream, in, n, d_partials);
            break;
        case MCAMD_REDUCE_FIRST_ADD:
            hipLaunchKernelGGL((reduce_kernel<T, MCAMD_REDUCE_FIRST_ADD>), g, b, 0, stream, in, n, d_partials);
            break;
        case MCAMD_REDUCE_UNROLL_LAST:
            hipLaunchKernelGGL((reduce_kernel<T, MCAMD_REDUCE_UNROLL_LAST>), g, b, 0, stream, in, n, d_partials);
            break;
        default:
            hipLaunchKernelGGL((reduce_kernel<T, MCAMD_REDUCE_GRID_STRIDE>), g, b, 0, stream, in, n, d_partials);
            break;
    }
    return hipGetLastError();
}

hipError_t launch_reduce(const void *d_in, uint64_t n, int precision, int variant, double *d_partials, uint32_t grid,
                         hipStream_t stream)
{
    return precision == 32 ? launch_reduce_t<float>(d_in, n, variant, d_partials, grid, stream)
                           : launch_reduce_t<double>(d_in, n, variant, d_partials, grid, stream);
}

}  // namespace mcamd
