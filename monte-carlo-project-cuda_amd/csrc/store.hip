// store.hip — trajectory-store kernel for gfx950: every St (and optionally every running
// barrier count) goes to HBM.
//
// Replaces simulateOptionPriceMultipleBlockGPU (trajectory overload, inc/testing.cuh:46-73:
// prices only) and simulate_outer_trajectories (inc/trajectories.cuh:273-351: prices + counts,
// the nested-MC outer stage).  The reference writes path-major, a[idx * N_STEPS + i]
// (inc/trajectories.cuh:304-305): consecutive lanes are N_STEPS elements apart, so every lane
// touches its own cache line.  Native layout here is step-major, a[step * n_local + path]:
// a thread owns 16 bytes' worth of consecutive paths (4 fp32 / 2 fp64), so each step is one
// 16 B-per-lane store and a wavefront writes 1 KiB contiguous — the widest coalesced store the
// memory pipeline has.  The path-major layout stays available for callers that need the
// reference's indexing (CSV dump, small tests); it is not the fast path.
//
// Same Philox counters and the same exponent arithmetic as price_impl.hpp (subsequence = global path id), so a stored trajectory's
// last row is bit-identical to the in-register path's terminal price.
// Algorithmic HBM traffic: sizeof(T) bytes per path-step (+4 with counts) + sizeof(T) per path
// payoff; no reads.  This is the bandwidth-bound configuration (BASELINE config 3).
#include "path_consts.hpp"

#include "mcamd.h"

namespace mcamd {

template <typename T>
struct StoreArgs {
    StepConsts<T> c;
    uint64_t seed;
    uint64_t path_offset;
    uint64_t n_local;
    T *traj;
    int32_t *counts;
    T *payoffs;
    bool vec_ok;  // rows are 16-byte aligned: vector stores allowed
};

// One step row: the thread's V consecutive paths.  Streaming data: written once, never re-read
// by this kernel, so the stores are non-temporal (measured against plain stores on the full-size
// job: 15.9 vs 16.5 ms, profiles/r02_store_variants.jsonl).  `row` is wave-uniform and `off` is the
// thread's element offset inside a row: with a 32-bit `off` the store takes the scalar-base +
// vector-offset form, and stepping to the next row costs the vector ALU nothing.
template <typename E, int V, bool VEC, typename OFF>
__device__ __forceinline__ void store_row(E *__restrict__ row, OFF off, const E (&val)[V], int n_valid)
{
    // byte offset in OFF's own width: a 32-bit offset zero-extended onto a uniform base is what selects the
    // scalar-base addressing form
    char *const p0 = reinterpret_cast<char *>(row) + static_cast<OFF>(off * static_cast<OFF>(sizeof(E)));
    if (VEC) {
        using VT = E __attribute__((ext_vector_type(V)));
        VT pack;
#pragma unroll
        for (int p = 0; p < V; ++p) pack[p] = val[p];
        __builtin_nontemporal_store(pack, reinterpret_cast<VT *>(p0));
    } else {
#pragma unroll
        for (int p = 0; p < V; ++p)
            if (p < n_valid) __builtin_nontemporal_store(val[p], reinterpret_cast<E *>(p0) + p);
    }
}

// The paths of one thread (V consecutive ones starting at `base`) through all steps.  OFF = uint32_t when a whole
// row is addressable with 32 bits (every realistic trajectory buffer), uint64_t otherwise.
template <typename T, bool WINDOW, int LAYOUT, bool VEC, typename OFF>
__device__ __forceinline__ void store_group(const StoreArgs<T> &a, const StepConsts<T> &c, const MathCtx<T> &m,
                                            const PhiloxKeys &key, uint64_t base, double &s, double &s2)
{
    constexpr int V = 16 / sizeof(T);
    constexpr int NB = Normals<T>::kPerBlock;
    const uint32_t n_full = c.n_sim / NB;         // Philox blocks whose NB steps are all simulated
    const uint32_t rem = c.n_sim - n_full * NB;   // steps of the last, partial block
    const OFF off = static_cast<OFF>(base);
    const int n_valid =
        VEC ? V : ((a.n_local - base >= static_cast<uint64_t>(V)) ? V : static_cast<int>(a.n_local - base));
    T St[V];
    PathState<T> ps[V];
    int32_t cnt[V];
#pragma unroll
    for (int p = 0; p < V; ++p) {
        St[p] = c.S_start;
        ps[p] = PathState<T>::start(c.S_start);
        cnt[p] = c.Ik;
    }
    auto advance = [&](const Exponents<T>(&nrm)[V], int j, uint32_t step) {
#pragma unroll
        for (int p = 0; p < V; ++p) {
            ps[p].step(nrm[p].x[j], m);
            St[p] = ps[p].value(m);
            if (WINDOW) cnt[p] += (c.B > St[p]) ? 1 : 0;
        }
        if (LAYOUT == MCAMD_STEP_MAJOR) {
            const uint64_t row = static_cast<uint64_t>(step) * a.n_local;   // wave-uniform: scalar unit
            store_row<T, V, VEC, OFF>(a.traj + row, off, St, n_valid);
            if (WINDOW && a.counts) store_row<int32_t, V, VEC, OFF>(a.counts + row, off, cnt, n_valid);
        } else {
#pragma unroll
            for (int p = 0; p < V; ++p)
                if (p < n_valid) {
                    const uint64_t idx = (base + p) * c.n_sim + step;
                    a.traj[idx] = St[p];
                    if (WINDOW && a.counts) a.counts[idx] = cnt[p];
                }
        }
    };
    for (uint32_t k = 0; k < n_full; ++k) {
        Exponents<T> nrm[V];
#pragma unroll
        for (int p = 0; p < V; ++p) nrm[p].fill(m, c, key, a.path_offset + base + p, k);
#pragma unroll
        for (int j = 0; j < NB; ++j) advance(nrm, j, k * NB + j);
    }
    if (rem) {
        Exponents<T> nrm[V];
#pragma unroll
        for (int p = 0; p < V; ++p) nrm[p].fill(m, c, key, a.path_offset + base + p, n_full);
#pragma unroll
        for (int j = 0; j < NB - 1; ++j)
            if (static_cast<uint32_t>(j) < rem) advance(nrm, j, n_full * NB + j);
    }
    T pay[V];
#pragma unroll
    for (int p = 0; p < V; ++p) {
        pay[p] = payoff<T, WINDOW>(St[p], cnt[p], c);
        if (p < n_valid) {
            const double pd = static_cast<double>(pay[p]);
            s += pd;
            s2 = __builtin_fma(pd, pd, s2);
        }
    }
    if (a.payoffs) store_row<T, V, VEC, OFF>(a.payoffs, off, pay, n_valid);
}

// VEC: every row is 16-byte aligned and n_local is a multiple of V, so each thread's group is full
// and each step is exactly one 16 B store per lane with no per-lane predicate at all.  Otherwise
// (ragged n_local or unaligned buffers) the same loop runs with guarded scalar stores.
template <typename T, bool WINDOW, int LAYOUT, bool VEC>
__global__ __launch_bounds__(kBlock) void store_kernel(StoreArgs<T> a, double *__restrict__ partials)
{
    constexpr int V = 16 / sizeof(T);
    const MathCtx<T> m = MathCtx<T>::init();
    const PhiloxKeys key = PhiloxKeys::make(a.seed);
    const StepConsts<T> c = resident(a.c);
    const uint64_t n_groups = (a.n_local + V - 1) / V;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    const bool narrow = a.n_local + V <= 0xffffffffull / 8;   // a row's byte offsets (prices or counts) fit 32 bits
    double s = 0.0, s2 = 0.0;
    for (uint64_t g = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; g < n_groups; g += stride) {
        if (narrow) store_group<T, WINDOW, LAYOUT, VEC, uint32_t>(a, c, m, key, g * V, s, s2);
        else store_group<T, WINDOW, LAYOUT, VEC, uint64_t>(a, c, m, key, g * V, s, s2);
    }
    block_sum2<kBlock>(s, s2);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = s;
        partials[2 * blockIdx.x + 1] = s2;
    }
}

// Diagnostic: the store kernel's launch shape and store stream with nothing simulated — every thread writes its 16
// bytes of every step row (non-temporal, scalar row base + 32-bit lane offset, as store_row does) and of the payoff
// row, the value being a counter.  What it measures is the HBM write ceiling of THIS access pattern on THIS device at
// the moment of the call; bench.py runs it beside the real kernel so that the line carries its own yardstick.
template <typename T>
__global__ __launch_bounds__(kBlock) void store_pattern_kernel(T *__restrict__ traj, T *__restrict__ payoffs,
                                                               uint64_t n_local, uint32_t n_sim)
{
    constexpr int V = 16 / sizeof(T);
    using VT = T __attribute__((ext_vector_type(V)));
    const uint64_t n_groups = n_local / V;   // whole 16-byte groups only (the launcher requires n_local % V == 0)
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t g = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; g < n_groups; g += stride) {
        const uint32_t off = static_cast<uint32_t>(g * V);
        VT v;
#pragma unroll
        for (int p = 0; p < V; ++p) v[p] = static_cast<T>(off + p);
        for (uint32_t step = 0; step < n_sim; ++step) {
            T *const row = traj + static_cast<uint64_t>(step) * n_local;   // wave-uniform
            __builtin_nontemporal_store(v, reinterpret_cast<VT *>(reinterpret_cast<char *>(row) + off * static_cast<uint32_t>(sizeof(T))));
#pragma unroll
            for (int p = 0; p < V; ++p) v[p] += T(1);
        }
        if (payoffs)
            __builtin_nontemporal_store(v, reinterpret_cast<VT *>(reinterpret_cast<char *>(payoffs) + off * static_cast<uint32_t>(sizeof(T))));
    }
}

hipError_t launch_store_pattern(uint64_t n_local, uint32_t n_sim, int precision, void *d_traj, void *d_payoffs,
                                uint32_t grid, hipStream_t stream)
{
    const dim3 g(grid), b(kBlock);
    if (precision == 32)
        hipLaunchKernelGGL(store_pattern_kernel<float>, g, b, 0, stream, static_cast<float *>(d_traj),
                           static_cast<float *>(d_payoffs), n_local, n_sim);
    else
        hipLaunchKernelGGL(store_pattern_kernel<double>, g, b, 0, stream, static_cast<double *>(d_traj),
                           static_cast<double *>(d_payoffs), n_local, n_sim);
    return hipGetLastError();
}

uint32_t store_grid(uint64_t n_local, int precision)
{
    const uint64_t v = precision == 32 ? 4 : 2;
    const uint64_t groups = (n_local + v - 1) / v;
    return clamp_grid((groups + kBlock - 1) / kBlock);
}

template <typename T>
static hipError_t launch_store_t(const PathJob &j, int layout, void *d_traj, int32_t *d_counts, void *d_payoffs,
                                 double *d_partials, uint32_t grid, hipStream_t stream)
{
    constexpr uint64_t V = 16 / sizeof(T);
    StoreArgs<T> a;
    a.c = make_consts<T>(j);
    a.seed = j.seed;
    a.path_offset = j.path_offset;
    a.n_local = j.n_local;
    a.traj = static_cast<T *>(d_traj);
    a.counts = d_counts;
    a.payoffs = static_cast<T *>(d_payoffs);
    a.vec_ok = (j.n_local % V == 0) && (reinterpret_cast<uintptr_t>(d_traj) % 16 == 0) &&
               (d_counts == nullptr || reinterpret_cast<uintptr_t>(d_counts) % 16 == 0) &&
               (d_payoffs == nullptr || reinterpret_cast<uintptr_t>(d_payoffs) % 16 == 0);
    const dim3 g(grid), b(kBlock);
#define MCAMD_LAUNCH_STORE(W, L, VEC) \
    hipLaunchKernelGGL((store_kernel<T, W, L, VEC>), g, b, 0, stream, a, d_partials)
    if (layout == MCAMD_STEP_MAJOR) {
        if (a.vec_ok) {
            if (j.window) MCAMD_LAUNCH_STORE(true, MCAMD_STEP_MAJOR, true);
            else MCAMD_LAUNCH_STORE(false, MCAMD_STEP_MAJOR, true);
        } else {
            if (j.window) MCAMD_LAUNCH_STORE(true, MCAMD_STEP_MAJOR, false);
            else MCAMD_LAUNCH_STORE(false, MCAMD_STEP_MAJOR, false);
        }
    } else {
        if (j.window) MCAMD_LAUNCH_STORE(true, MCAMD_PATH_MAJOR, false);
        else MCAMD_LAUNCH_STORE(false, MCAMD_PATH_MAJOR, false);
    }
#undef MCAMD_LAUNCH_STORE
    return hipGetLastError();
}

hipError_t launch_store(const PathJob &j, int layout, void *d_traj, int32_t *d_counts, void *d_payoffs,
                        double *d_partials, uint32_t grid, hipStream_t stream)
{
    return j.precision == 32 ? launch_store_t<float>(j, layout, d_traj, d_counts, d_payoffs, d_partials, grid, stream)
                             : launch_store_t<double>(j, layout, d_traj, d_counts, d_payoffs, d_partials, grid, stream);
}

}  // namespace mcamd
