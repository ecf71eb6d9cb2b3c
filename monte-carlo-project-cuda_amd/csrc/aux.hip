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
    uint32_t skew;   // (address of normals) mod 128, taken on the host: where the buffer starts inside its cache line
};

// VEC: every path row starts 16-byte aligned (n_sim a multiple of 16 / sizeof(T), aligned buffer).  A row is then
// read as whole 128-byte CACHE LINES of the buffer, not as 128-byte pieces counted from the row's own start: a row
// of 252 floats is 1008 bytes, so row-relative pieces straddle two lines each and every line is requested twice
// (measured 5.05 TB/s).  Tile k of a row is line number k counted from the line that holds the row's first
// element; the 16-byte vectors of a line that belong to the neighbouring rows are simply not consumed (row starts
// and vector boundaries are both multiples of 16 bytes, so a vector is inside the row or outside it, never both).
// A lane fetches 16 bytes, one wave-wide load covers 8 rows x 128 bytes; lanes walk their own step index, which
// differs from row to row by the row's offset inside its first line.
template <typename T, bool WINDOW, bool VEC>
__global__ __launch_bounds__(kBlock) void from_normals_kernel(ArrayArgs<T> a, double *__restrict__ partials)
{
    constexpr int kWaves = kBlock / kWave;
    constexpr int TS = 128 / sizeof(T);        // tile width in elements: one 128 B line per path row
    constexpr int kRowsPerLoad = kWave / TS;   // path rows one wave-wide scalar-element load covers (non-VEC path)
    constexpr int kPad = 16 / sizeof(T);       // one 16-byte vector of padding: rows stay 16-byte aligned
    __shared__ alignas(16) T tile[kWaves][kWave][TS + kPad];
    const StepConsts<T> c = resident(a.c);
    const MathCtx<T> m = MathCtx<T>::init();
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int lcol = lane % TS, lrow = lane / TS;
    const uint64_t n_tiles = (a.n_local + kWave - 1) / kWave;  // one tile row = 64 paths
    const uint64_t wave_stride = static_cast<uint64_t>(gridDim.x) * kWaves;
    double s = 0.0, s2 = 0.0;

    // VEC staging geometry: V elements (16 bytes) per lane per load, 8 lanes per line, 8 rows per load
    constexpr int V = 16 / sizeof(T);
    constexpr int kLanesPerRow = TS / V;                  // 8
    constexpr int kRowsPerVecLoad = kWave / kLanesPerRow; // 8
    constexpr int kVecLoads = kWave / kRowsPerVecLoad;    // 8 wave-wide loads per 64-row x 128-byte tile
    using VT = T __attribute__((ext_vector_type(V)));
    const int vrow = lane / kLanesPerRow, vchunk = lane % kLanesPerRow;
    // all addresses below are byte offsets from a.normals (int64: a row's first line may start up to 127 bytes
    // before the buffer); the buffer's own position inside its cache line is a kernel argument (a.skew), so the
    // kernel never turns the pointer into an integer and its loads stay in the global address space
    const int64_t row_bytes = static_cast<int64_t>(c.n_sim) * sizeof(T);
    const int64_t buf_skew = a.skew;
    const int64_t buf_bytes = static_cast<int64_t>(a.n_local) * row_bytes;
    const char *const buf = reinterpret_cast<const char *>(a.normals);

    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * kWaves + wave; t < n_tiles; t += wave_stride) {
        const uint64_t path0 = t * kWave;
        const uint64_t my_path = path0 + lane;
        PathState<T> ps = PathState<T>::start(c.S_start);
        int32_t count = c.Ik;
        if (VEC) {
            // the lines this lane fetches: row path0 + i * 8 + vrow, i = 0..7; and this lane's own row as a consumer
            int64_t line0[kVecLoads];
            bool row_ok[kVecLoads];
#pragma unroll
            for (int i = 0; i < kVecLoads; ++i) {
                const uint64_t p = path0 + i * kRowsPerVecLoad + vrow;
                row_ok[i] = p < a.n_local;
                const int64_t row0 = static_cast<int64_t>(row_ok[i] ? p : 0) * row_bytes;
                line0[i] = row0 - ((buf_skew + row0) & 127);
            }
            const int64_t my_row0 = static_cast<int64_t>(my_path < a.n_local ? my_path : 0) * row_bytes;
            const int32_t my_skew = static_cast<int32_t>((buf_skew + my_row0) & 127);   // bytes of line 0 before my row
            // lines per row: the widest row of the wave decides (rows differ by at most one)
            const uint32_t my_lines = static_cast<uint32_t>((my_skew + row_bytes + 127) / 128);  // < 2^32: n_sim is 32-bit
            uint32_t n_lines = my_lines;
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) {
                const uint32_t o = __shfl_xor(n_lines, off, kWave);
                n_lines = o > n_lines ? o : n_lines;
            }
            VT pre[kVecLoads];
            auto prefetch = [&](uint32_t k) {
#pragma unroll
                for (int i = 0; i < kVecLoads; ++i) {
                    const int64_t adr = line0[i] + static_cast<int64_t>(k) * 128 + vchunk * 16;
                    const bool ok = row_ok[i] && adr >= 0 && adr + 16 <= buf_bytes;
                    pre[i] = *reinterpret_cast<const VT *>(buf + (ok ? adr : 0));   // out-of-buffer vectors re-read element 0
                }
            };
            prefetch(0);
            for (uint32_t k = 0; k < n_lines; ++k) {
#pragma unroll
                for (int i = 0; i < kVecLoads; ++i)   // one ds_write_b128 per staged load
                    *reinterpret_cast<VT *>(&tile[wave][i * kRowsPerVecLoad + vrow][vchunk * V]) = pre[i];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (k + 1 < n_lines) prefetch(k + 1);
                // the lane's line comes back as 8 16-byte LDS reads; a vector is consumed iff it lies inside the row
                VT zz[kLanesPerRow];
#pragma unroll
                for (int i = 0; i < kLanesPerRow; ++i) zz[i] = *reinterpret_cast<const VT *>(&tile[wave][lane][i * V]);
                const int32_t line_off = static_cast<int32_t>(k) * 128 - my_skew;   // byte offset of this line in my row
#pragma unroll
                for (int i = 0; i < kLanesPerRow; ++i) {
                    const int32_t boff = line_off + i * 16;
                    const bool live = my_path < a.n_local && boff >= 0 && boff < row_bytes;
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        // a dead vector multiplies the price by 2^0
                        const T x = live ? fma_t(zz[i][e], c.vol, c.drift) : T(0);
                        ps.step(x, m);
                        if (WINDOW) count += (live && c.B > ps.value(m)) ? 1 : 0;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        } else {
            for (uint32_t s0 = 0; s0 < c.n_sim; s0 += TS) {
                const uint32_t n_cols = (c.n_sim - s0 < static_cast<uint32_t>(TS)) ? c.n_sim - s0 : TS;
                // stage: global reads are contiguous along a path's row, LDS holds [path][step]
                for (int r = 0; r < kWave; r += kRowsPerLoad) {
                    const uint64_t p = path0 + r + lrow;
                    if (p < a.n_local && static_cast<uint32_t>(lcol) < n_cols)
                        tile[wave][r + lrow][lcol] = a.normals[p * c.n_sim + s0 + lcol];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (my_path < a.n_local) {
                    for (uint32_t j = 0; j < n_cols; ++j) {
                        ps.step(fma_t(tile[wave][lane][j], c.vol, c.drift), m);
                        if (WINDOW) count += (c.B > ps.value(m)) ? 1 : 0;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
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
    ArrayArgs<T> a{make_consts<T>(j), j.n_local, static_cast<const T *>(d_normals), static_cast<T *>(d_payoffs),
                   static_cast<uint32_t>(reinterpret_cast<uintptr_t>(d_normals) & 127)};
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
    const PhiloxKeys key = PhiloxKeys::make(seed);
    const uint64_t n_blocks = (n + NB - 1) / NB;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t k = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; k < n_blocks; k += stride) {
        Normals<T> nrm;
        nrm.fill(m, key, 0, k);
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
// What a variant selects is the in-block schedule.  The load phase is the same for all: 16 B per lane per
// load, and variants 3-5 keep the reference's "first add on load" shape as TWO such loads per thread per chunk
// (the reference's one-float-per-load version, 4 B per lane, ran at 2.9-3.1 TB/s here; r01_aux_kernels.json), so
// their grid grows with n (capped: a block then walks several chunks before its tree).
// ---------------------------------------------------------------------------------------------
// Sum of the two 16-byte vectors a thread owns in the chunk starting at element `base` (vector index base/V + tid
// and + kBlock more); elements past n_vec vectors contribute 0.
template <typename T>
__device__ __forceinline__ double load2v(const T *__restrict__ in, uint64_t n_vec, uint64_t vbase, int tid)
{
    constexpr int V = 16 / sizeof(T);
    using VT = T __attribute__((ext_vector_type(V)));
    const VT *vin = reinterpret_cast<const VT *>(in);
    const uint64_t i0 = vbase + tid, i1 = i0 + kBlock;
    VT a, b;
#pragma unroll
    for (int j = 0; j < V; ++j) a[j] = b[j] = T(0);
    if (i0 < n_vec) a = vin[i0];
    if (i1 < n_vec) b = vin[i1];
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < V; ++j) v += static_cast<double>(a[j]) + static_cast<double>(b[j]);
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
        // four 16-byte loads in flight per lane: a persistent grid of 2048 workgroups needs the extra depth to keep
        // HBM busy (one load per trip: 5.8 TB/s on a 32 GB fp32 input; profiles/r02_aux_kernels.json)
        uint64_t i = gtid;
        for (; i + 3 * stride < n_vec; i += 4 * stride) {
            VT x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = vin[i + u * stride];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < V; ++j) v += static_cast<double>(x[u][j]);
        }
        for (; i < n_vec; i += stride) {
            const VT x = vin[i];
#pragma unroll
            for (int j = 0; j < V; ++j) v += static_cast<double>(x[j]);
        }
        for (uint64_t t = n_vec * V + gtid; t < n; t += stride) v += static_cast<double>(in[t]);
        block_sum2<kBlock>(v, zero);
    } else {
        // first add on load: two 16-byte vectors per thread per chunk; the grid is capped at kMaxGrid blocks, so
        // for very large n a block walks several chunks before its tree.  A misaligned input and the last
        // n % V elements take the scalar tail.
        constexpr int V = 16 / sizeof(T);
        const uint64_t n_vec = (reinterpret_cast<uintptr_t>(in) % 16 == 0) ? n / V : 0;
        const uint64_t chunk_stride = static_cast<uint64_t>(gridDim.x) * (2 * kBlock);
        for (uint64_t vbase = static_cast<uint64_t>(blockIdx.x) * (2 * kBlock); vbase < n_vec; vbase += chunk_stride)
            v += load2v(in, n_vec, vbase, tid);
        const uint64_t gtid = static_cast<uint64_t>(blockIdx.x) * kBlock + tid;
        for (uint64_t i = n_vec * V + gtid; i < n; i += static_cast<uint64_t>(gridDim.x) * kBlock)
            v += static_cast<double>(in[i]);
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

// grid_finish's sum as a launch of its own (one workgroup of kBlock threads, small_final_sum's order): what a kernel
// that finishes itself computes in its last workgroup, for the callers that ask for the separate launch.
template <int N>
__global__ __launch_bounds__(kBlock) void small_final_kernel(const double *__restrict__ partials, uint32_t n_records,
                                                             double *__restrict__ out, double n_value)
{
    double v[N];
    small_final_sum<kBlock, N>(partials, n_records, v);
    if (threadIdx.x == 0) write_final(out, v, N, n_value);
}

hipError_t launch_small_final(const double *d_partials, uint32_t n_records, int record_doubles, double *d_out,
                              hipStream_t stream, double n_value)
{
    if (record_doubles == 2)
        hipLaunchKernelGGL(small_final_kernel<2>, dim3(1), dim3(kBlock), 0, stream, d_partials, n_records, d_out, n_value);
    else if (record_doubles == 5)
        hipLaunchKernelGGL(small_final_kernel<5>, dim3(1), dim3(kBlock), 0, stream, d_partials, n_records, d_out, n_value);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_final_reduce(const double *d_partials, uint32_t n_records, int record_doubles, double *d_out,
                               hipStream_t stream, double n_value)
{
    switch (record_doubles) {
    case 2:
        hipLaunchKernelGGL(final_reduce_kernel<2>, dim3(1), dim3(kFinalBlock), 0, stream, d_partials, n_records, d_out,
                           n_value);
        break;
    case kNmcRecord:
        hipLaunchKernelGGL(final_reduce_kernel<kNmcRecord>, dim3(1), dim3(kFinalBlock), 0, stream, d_partials, n_records,
                           d_out, n_value);
        break;
    case 5:
        hipLaunchKernelGGL(final_reduce_kernel<5>, dim3(1), dim3(kFinalBlock), 0, stream, d_partials, n_records, d_out,
                           n_value);
        break;
    default:
        return hipErrorInvalidValue;   // a record width nobody instantiated must not fall through to another one
    }
    return hipGetLastError();
}

uint32_t reduce_grid(uint64_t n, int variant)
{
    if (variant == MCAMD_REDUCE_GRID_STRIDE) {
        // memory-bound: several rounds of resident workgroups (256 CUs x 8), grid-stride the rest; 2048 workgroups
        // alone left HBM under-subscribed (5.98 vs 6.15 TB/s for the chunked variants at 32 GB)
        const uint64_t want = (n + kBlock * 16 - 1) / (kBlock * 16);
        return static_cast<uint32_t>(want < 1 ? 1 : (want > 16384 ? 16384 : want));
    }
    // variants 3-5: one chunk = 2 x 256 sixteen-byte vectors; sized for fp32 (4 per vector) — an fp64 input just
    // gives every block two chunks
    return clamp_grid((n / 4 + 2 * kBlock - 1) / (2 * kBlock));
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
