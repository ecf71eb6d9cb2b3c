// mc_device.hpp — device-side building blocks shared by the gfx950 kernels:
// Philox4x32-10 in registers, rocRAND-convention Box-Muller, GBM step, wave64 reductions.
// gfx950 only: wavefront = 64 lanes, hard-coded.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fast64.hpp"

namespace mcamd {

#define MCAMD_TAB_DECL static __device__ const
#include "tables64.inc"
#undef MCAMD_TAB_DECL

constexpr int kWave = 64;
constexpr int kBlock = 256;  // 4 waves: one per SIMD of a CU

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011).  Counter layout is rocRAND's: (block_lo, block_hi,
// subsequence_lo, subsequence_hi); key = (seed_lo, seed_hi).  Replaces the per-thread
// curandState + setup_kernel of the reference (inc/tool.cuh:192-195): no state lives in HBM.
// The round keys depend only on the seed (a kernel argument), so they stay in SGPRs.
// ---------------------------------------------------------------------------------------------
struct U4 {
    uint32_t x, y, z, w;
};

// A value copied into a vector register, opaque to the optimiser.  Why: on gfx950 a full-rate 32-bit VALU
// instruction (v_xor / v_bitop3 / v_fma_f32 / v_mul_f32 ...) that reads a SCALAR register issues in 4 cycles instead of
// 2 (tools/ubench_bank.hip -> profiles/r02_operand_costs.txt: v_bitop3 v,v,s 4.2 vs v,v,v 2.4; v_xor s,v 4.2 vs 2.2;
// v_fma_f32 v,s,v 4.2 vs 2.4; literals and fp64 / multiply instructions are not affected).  hipcc keeps every
// wave-uniform value in scalar registers, so the constants of the hot full-rate instructions — the Philox round keys,
// the fp32 drift and volatility — are moved to vector registers once per kernel.
__device__ __forceinline__ uint32_t vgpr_resident(uint32_t s)
{
    uint32_t v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}
__device__ __forceinline__ float vgpr_resident(float s)
{
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}

// The ten round-key pairs of Philox4x32-10 for one seed, in vector registers (20 VGPRs, built once per kernel).
struct PhiloxKeys {
    uint32_t k0[10], k1[10];
    uint32_t k1_first;   // k1[0] once more, left to the compiler (a scalar register): round 1 folds it into scalar terms
    __device__ __forceinline__ static PhiloxKeys make(uint64_t seed)
    {
        constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
        PhiloxKeys k;
        uint32_t a = static_cast<uint32_t>(seed), b = static_cast<uint32_t>(seed >> 32);
        k.k1_first = b;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            k.k0[i] = vgpr_resident(a);
            k.k1[i] = vgpr_resident(b);
            a += W0;
            b += W1;
        }
        return k;
    }
};

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, const PhiloxKeys &key)
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    // round 1 with plain xors: in the path kernels c0 (block index) is wave-uniform and c1 (its high word) is zero, so
    // one of the two three-input xors collapses and the compiler sees which inputs are scalar
    {
        const uint64_t p0 = static_cast<uint64_t>(M0) * c0;
        const uint64_t p1 = static_cast<uint64_t>(M1) * c2;
        const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ key.k0[0];
        // hi(M0 c0) and the key are both wave-uniform there: their xor runs on the scalar unit, leaving one vector xor
        const uint32_t n2 = (static_cast<uint32_t>(p0 >> 32) ^ key.k1_first) ^ c3;
        c1 = static_cast<uint32_t>(p1);
        c3 = static_cast<uint32_t>(p0);
        c0 = n0;
        c2 = n2;
    }
#pragma unroll
    for (int i = 1; i < 10; ++i) {
        const uint64_t p0 = static_cast<uint64_t>(M0) * c0;
        const uint64_t p1 = static_cast<uint64_t>(M1) * c2;
        // three-input xor in one full-rate instruction (v_bitop3_b32, truth table 0x96), all three inputs in VGPRs
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32(static_cast<uint32_t>(p1 >> 32), c1, key.k0[i], 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32(static_cast<uint32_t>(p0 >> 32), c3, key.k1[i], 0x96);
        c1 = static_cast<uint32_t>(p1);
        c3 = static_cast<uint32_t>(p0);
        c0 = n0;
        c2 = n2;
    }
    return U4{c0, c1, c2, c3};
}

__device__ __forceinline__ U4 philox_block(const PhiloxKeys &key, uint64_t subsequence, uint64_t block)
{
    return philox4x32_10(static_cast<uint32_t>(block), static_cast<uint32_t>(block >> 32),
                         static_cast<uint32_t>(subsequence), static_cast<uint32_t>(subsequence >> 32), key);
}

// ---------------------------------------------------------------------------------------------
// Per-workgroup math context.  The fp64 path uses the table-driven functions of fast64.hpp; their
// three 512-entry tables (20 KB) are copied into LDS once per workgroup, so the per-lane lookups
// run on the LDS pipe beside the VALU.  The fp32 path needs nothing (hardware transcendentals).
// ---------------------------------------------------------------------------------------------
template <typename T>
struct MathCtx {
    template <bool ROTATED = false>
    __device__ __forceinline__ static MathCtx init() { return MathCtx{}; }
};

template <>
struct MathCtx<double> {
    f64::Tables t;
    double exp_c1;  // f64::kExpC1, resident in vector registers for the step loops
    // Must be called by every thread of the workgroup (contains a barrier).  The copy runs at raised wave priority:
    // the SIMD's arbiter serves the oldest wave first, and beside three or four older waves that keep the vector ALU
    // saturated a newly launched wave otherwise needs ~58 us for these thirty instructions (tools/init_probe.hip:
    // 1.4 us on an idle CU) — during which its workgroup holds LDS and registers without computing.
    // ROTATED: the sin/cos table is copied rotated by an eighth of a turn (entry j = arc j + N/8), which is what the
    // pair-sum loop (PairSum below) looks up; a kernel uses one form or the other, never both.
    template <bool ROTATED = false>
    __device__ __forceinline__ static MathCtx init()
    {
        __builtin_amdgcn_s_setprio(3);
        __shared__ f64::D2 s_log[MCAMD_TAB_N];
        __shared__ f64::D2 s_sincos[MCAMD_TAB_N];
        __shared__ double s_exp_hi[256];
        __shared__ double s_exp_lo[256];
        static_assert(MCAMD_TAB_N == 512, "f64::table_entry's plane variant assumes 512 entries");
        for (int i = threadIdx.x; i < MCAMD_TAB_N; i += blockDim.x) {
            const int j = ROTATED ? ((i + MCAMD_TAB_N / 8) & (MCAMD_TAB_N - 1)) : i;
#if defined(MCAMD_LDS_PLANES)   // experiment: two 8-byte planes per table (fast64.hpp table_entry)
            reinterpret_cast<double *>(s_log)[i] = kLogTab[i][0];
            reinterpret_cast<double *>(s_log)[MCAMD_TAB_N + i] = kLogTab[i][1];
            reinterpret_cast<double *>(s_sincos)[i] = kSinCosTab[j][0];
            reinterpret_cast<double *>(s_sincos)[MCAMD_TAB_N + i] = kSinCosTab[j][1];
#else
            s_log[i] = f64::D2{kLogTab[i][0], kLogTab[i][1]};
            s_sincos[i] = f64::D2{kSinCosTab[j][0], kSinCosTab[j][1]};
#endif
        }
        for (int i = threadIdx.x; i < 256; i += blockDim.x) {
            s_exp_hi[i] = kExpHiTab[i];
            s_exp_lo[i] = kExpLoTab[i];
        }
        __syncthreads();
        __builtin_amdgcn_s_setprio(0);
        return MathCtx{f64::Tables{s_log, s_sincos, s_exp_hi, s_exp_lo}, f64::exp_c1_resident()};
    }
};

// ---------------------------------------------------------------------------------------------
// Box-Muller, rocRAND convention (rocrand_normal.h box_muller / box_muller_double):
//   fp32: u = 2^-32 + x 2^-32, angle = 2 pi (2^-32 + y 2^-32), (sin, cos) * sqrt(-2 ln u)
//   fp64: u = 2^-53 + v1 2^-53 with v1 = x ^ (y << 21); angle = pi * (2^-52 + v2 2^-52)
// fp64 uses the table-driven functions of fast64.hpp (u and the angle's reduction are exact, the elementary
// functions are within 2 ulp of libm).  fp32 uses the hardware transcendental unit directly: v_log_f32 (log2), v_sqrt_f32 and
// v_sin_f32 / v_cos_f32, whose operand is in revolutions, so the angle needs no 2 pi multiply
// and no range reduction.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void box_muller(uint32_t x, uint32_t y, float &a, float &b)
{
    constexpr float k2pow32inv = 2.3283064365386963e-10f;
    const float u = __builtin_fmaf(static_cast<float>(x), k2pow32inv, k2pow32inv);
    const float rev = __builtin_fmaf(static_cast<float>(y), k2pow32inv, k2pow32inv);
    // -2 ln u = (-2 ln 2) log2 u
    const float s = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u));
    a = __builtin_amdgcn_sinf(rev) * s;
    b = __builtin_amdgcn_cosf(rev) * s;
}

__device__ __forceinline__ void box_muller(const U4 &w, const MathCtx<double> &m, double &a, double &b)
{
    const double u = f64::u53(w.x, w.y, 0x1p-53);   // (v1 + 1) 2^-53, exact
    const double s = f64::sqrt_unclamped(f64::neg2log(u, m.t.log_tab));   // -2 ln u > 0 for every u in (0, 1] (fast64.hpp)
    double sn, cs;
    f64::sincos_bits(w.z, w.w, m.t.sincos_tab, sn, cs);   // angle pi (v2 + 1) 2^-52, from the bits
    a = sn * s;
    b = cs * s;
}

// Normals of one Philox block: 4 fp32 (steps 4k..4k+3) or 2 fp64 (steps 2k, 2k+1).
template <typename T>
struct Normals;

template <>
struct Normals<float> {
    static constexpr int kPerBlock = 4;
    float z[4];
    __device__ __forceinline__ void fill(const MathCtx<float> &, const PhiloxKeys &key, uint64_t subsequence,
                                         uint64_t block)
    {
        const U4 w = philox_block(key, subsequence, block);
        box_muller(w.x, w.y, z[0], z[1]);
        box_muller(w.z, w.w, z[2], z[3]);
    }
};

template <>
struct Normals<double> {
    static constexpr int kPerBlock = 2;
    double z[2];
    __device__ __forceinline__ void fill(const MathCtx<double> &m, const PhiloxKeys &key, uint64_t subsequence,
                                         uint64_t block)
    {
        const U4 w = philox_block(key, subsequence, block);
        box_muller(w, m, z[0], z[1]);
    }
};

// ---------------------------------------------------------------------------------------------
// GBM step constants and the running price of a path:  St *= exp(drift + vol * G)
// (inc/trajectories.cuh:146, :224, :302; one-step form :75).  The exponent constants are pre-scaled on the host
// to the unit the precision's exponential consumes: log2(e) for fp32 (v_exp_f32 computes 2^x), 65536 / ln 2 for
// fp64 (f64::ExpAcc splits the exponent at 2^-16 octaves).
// ---------------------------------------------------------------------------------------------
template <typename T>
struct StepConsts {
    T drift;    // (r - v^2/2) dt          [times the exponent scale]
    T vol;      // v sqrt(dt)              [times the exponent scale]
    T vol_bm;   // fp32: vol * sqrt(2 ln 2), so that vol * sqrt(-2 ln u) = vol_bm * sqrt(-log2 u); fp64: = vol
    T K;        // strike
    T B;        // barrier
    T S_start;  // S0, or Sk when restarting
    T logB;     // ln(B / S_start) in exponent units, -inf when B <= 0: barrier test in log space
    T win_delta;  // fp64: half-width (exponent units) of the band where the cheap barrier test defers to the exact one
    int32_t P1, P2, Ik;
    uint32_t n_sim;  // steps to simulate = n_steps - Tk
};

// The step constants as the kernels' inner loops should read them: in fp32 the three constants of the full-rate
// instructions (x = fma(sin, sv, drift), sv = r * vol_bm, the log-space update) go to vector registers — a scalar
// operand would double those instructions' issue time (see vgpr_resident) — in fp64 nothing changes (fp64
// instructions take four cycles whatever their operands).
__device__ __forceinline__ StepConsts<float> resident(StepConsts<float> c)
{
    c.drift = vgpr_resident(c.drift);
    c.vol = vgpr_resident(c.vol);
    c.vol_bm = vgpr_resident(c.vol_bm);
    return c;
}
__device__ __forceinline__ StepConsts<double> resident(const StepConsts<double> &c) { return c; }

// Running price of one path.  fp32: the price itself, one v_exp_f32 and one multiply per step.  fp64: the start
// price and the factored product of the step exponentials (f64::ExpAcc); value() evaluates it.
template <typename T>
struct PathState;

template <>
struct PathState<float> {
    float St;
    __device__ __forceinline__ static PathState start(float S) { return PathState{S}; }
    __device__ __forceinline__ void step(float x, const MathCtx<float> &) { St *= __builtin_amdgcn_exp2f(x); }
    __device__ __forceinline__ float value(const MathCtx<float> &) const { return St; }
    // barrier test B > St (inc/trajectories.cuh:147): the price is at hand
    __device__ __forceinline__ void arm_barrier(float) {}
    __device__ __forceinline__ int32_t below_barrier(const StepConsts<float> &c, const MathCtx<float> &) const { return c.B > St ? 1 : 0; }
};

template <>
struct PathState<double> {
    double S0;
    f64::ExpAcc a;
    double kq;  // k - theta - kExpScale as a double, theta = log2(B / S0) * 65536 (set by arm_barrier; unused otherwise)
    __device__ __forceinline__ static PathState start(double S) { return PathState{S, f64::exp_acc_init(), 0.0}; }
    __device__ __forceinline__ void step(double y, const MathCtx<double> &m) { kq += f64::exp_acc_mul(a, y, m.exp_c1); }
    __device__ __forceinline__ double value(const MathCtx<double> &m) const
    {
        return f64::exp_acc_value(S0, a, m.t.exp_hi_tab, m.t.exp_lo_tab);
    }
    // Barrier test B > St without evaluating St.  In exponent units log2(St / B) * 65536 = k - theta + kExpScale ln P,
    // and ln P = (P - 1) up to c.win_delta (P stays within 1 +- n * 5.3e-6: f64::exp_acc_window_delta), so
    // q = fma(P, kExpScale, kq) decides unless |q| <= win_delta; then — for the whole wavefront, a few times per
    // ten thousand steps — the price is evaluated and compared exactly as the trajectory-store kernel does, so both
    // count the same steps.  theta: ln(B / S0) in exponent units (c.logB when the path starts at c.S_start).
    __device__ __forceinline__ void arm_barrier(double theta) { kq = -theta - f64::kExpScale; }
    // Returns 1 when the price is below the barrier, else 0 (what the count grows by).  On the fast path that is the
    // sign bit of q — one full-rate shift instead of compare + select (8 issue cycles); q cannot be -0 there
    // (|q| > win_delta).
    __device__ __forceinline__ int32_t below_barrier(const StepConsts<double> &c, const MathCtx<double> &m) const
    {
        const double q = __builtin_fma(a.P, f64::kExpScale, kq);
        const bool unsure = !(__builtin_fabs(q) > c.win_delta);   // also true for a NaN
        if (__builtin_amdgcn_ballot_w64(unsure) != 0) {
            // marker for tools/count_valu_slots.py: the basic block holding it runs for a few wavefront-steps in ten
            // thousand and is left out of the static per-step instruction count (the PMC count in profiles/ is the
            // dynamic check)
            asm volatile("; MCAMD_RARE_BLOCK");
            return c.B > value(m) ? 1 : 0;
        }
        return static_cast<int32_t>(f64::hi32(q) >> 31);
    }
};

// Exponents of one Philox block: x[j] = drift + vol * z[j] for the block's normals z (4 fp32 / 2 fp64),
// without materialising z: the Box-Muller radius is multiplied by vol once per pair and the constant
// sqrt(2 ln 2) of the fp32 radius is folded into vol_bm (two multiplies fewer per pair than
// normal-then-scale; the value differs from fma(z, vol, drift) by rounding only).
template <typename T>
struct Exponents;

template <>
struct Exponents<float> {
    static constexpr int kPerBlock = 4;
    float x[4];
    __device__ __forceinline__ void pair(uint32_t a, uint32_t b, const StepConsts<float> &c, float &x0, float &x1)
    {
        constexpr float k2pow32inv = 2.3283064365386963e-10f;
        const float u = __builtin_fmaf(static_cast<float>(a), k2pow32inv, k2pow32inv);
        const float rev = __builtin_fmaf(static_cast<float>(b), k2pow32inv, k2pow32inv);
        const float sv = __builtin_amdgcn_sqrtf(-__builtin_amdgcn_logf(u)) * c.vol_bm;
        x0 = __builtin_fmaf(__builtin_amdgcn_sinf(rev), sv, c.drift);
        x1 = __builtin_fmaf(__builtin_amdgcn_cosf(rev), sv, c.drift);
    }
    __device__ __forceinline__ void fill(const MathCtx<float> &, const StepConsts<float> &c, const PhiloxKeys &key,
                                         uint64_t subsequence, uint64_t block)
    {
        const U4 w = philox_block(key, subsequence, block);
        pair(w.x, w.y, c, x[0], x[1]);
        pair(w.z, w.w, c, x[2], x[3]);
    }
};

template <>
struct Exponents<double> {
    static constexpr int kPerBlock = 2;
    double x[2];
    __device__ __forceinline__ void fill(const MathCtx<double> &m, const StepConsts<double> &c, const PhiloxKeys &key,
                                         uint64_t subsequence, uint64_t block)
    {
        const U4 w = philox_block(key, subsequence, block);
        const double u = f64::u53(w.x, w.y, 0x1p-53);
        const double sv = f64::sqrt_scaled(f64::neg2log(u, m.t.log_tab), c.vol_bm);
        double sn, cs;
        f64::sincos_bits(w.z, w.w, m.t.sincos_tab, sn, cs);
        x[0] = f64::fma_vvs(sn, sv, c.drift);
        x[1] = f64::fma_vvs(cs, sv, c.drift);
    }
};

// Sum of the normals of one Philox block, for the paths that need nothing else of them (a window-less path in log
// space: ln(S_T / S_0) = n drift + vol (z_1 + ... + z_n)).  A Box-Muller pair is (r sin a, r cos a), so its sum is
//     r (sin a + cos a) = sqrt(2) r sin(a + pi/4):
// ONE trigonometric evaluation per pair instead of two, and no product with the radius per normal.  Every step
// still draws its normal from the same Philox words; this is algebra on the pair, exact up to rounding (the sum
// differs from adding rocRAND's two normals by about an ulp).  The sums are returned in units of kUnit so that the
// constant factors fold into ONE multiplication at the end of the path:
//     z_0 + ... + z_{NB-1} = kUnit * (add_block(acc, ...) - acc),
//     fp32: kUnit = sqrt(4 ln 2)  (r = sqrt(2 ln 2) sqrt(-log2 u); v_sin_f32 takes revolutions: + 1/8)
//     fp64: kUnit = sqrt(2)       (the rotated sin/cos table of MathCtx<double>::init<true>)
// head(n) is the sum of the block's FIRST n normals (n < NB: a path's last, partial block) in the same units.
template <typename T>
struct PairSum;

template <>
struct PairSum<float> {
    static constexpr float kUnit = 1.6651092223153954f;   // sqrt(4 ln 2)
    // sqrt(-log2 u) and the angle in revolutions of one pair (rocRAND's uniforms: 2^-32 + x 2^-32)
    __device__ __forceinline__ static void polar(uint32_t a, uint32_t b, float shift, float &t, float &rev)
    {
        constexpr float k2pow32inv = 2.3283064365386963e-10f;
        const float u = __builtin_fmaf(static_cast<float>(a), k2pow32inv, k2pow32inv);
        rev = __builtin_fmaf(static_cast<float>(b), k2pow32inv, k2pow32inv + shift);
        t = __builtin_amdgcn_sqrtf(-__builtin_amdgcn_logf(u));
    }
    __device__ __forceinline__ static float pair(uint32_t a, uint32_t b)
    {
        float t, rev;
        polar(a, b, 0.125f, t, rev);
        return t * __builtin_amdgcn_sinf(rev);
    }
    // acc + the block's sum (two fused multiply-adds)
    __device__ __forceinline__ static float add_block(float acc, const MathCtx<float> &, const PhiloxKeys &key,
                                                       uint64_t subsequence, uint64_t block)
    {
        const U4 w = philox_block(key, subsequence, block);
        float t1, r1, t2, r2;
        polar(w.x, w.y, 0.125f, t1, r1);
        polar(w.z, w.w, 0.125f, t2, r2);
        acc = __builtin_fmaf(t1, __builtin_amdgcn_sinf(r1), acc);
        return __builtin_fmaf(t2, __builtin_amdgcn_sinf(r2), acc);
    }
    __device__ __forceinline__ static float head(const MathCtx<float> &, const PhiloxKeys &key, uint64_t subsequence,
                                                  uint64_t block, uint32_t n)
    {
        constexpr float kInvSqrt2 = 0.70710678118654752f;
        const U4 w = philox_block(key, subsequence, block);
        float t, rev;
        if (n == 1) {   // z0 alone: the sine member of the first pair
            polar(w.x, w.y, 0.0f, t, rev);
            return t * __builtin_amdgcn_sinf(rev) * kInvSqrt2;
        }
        float s = pair(w.x, w.y);
        if (n == 3) {
            polar(w.z, w.w, 0.0f, t, rev);
            s = __builtin_fmaf(t * __builtin_amdgcn_sinf(rev), kInvSqrt2, s);
        }
        return s;
    }
};

template <>
struct PairSum<double> {
    static constexpr double kUnit = 1.4142135623730951;   // sqrt(2)
    // acc + the block's sum (one fused multiply-add)
    __device__ __forceinline__ static double add_block(double acc, const MathCtx<double> &m, const PhiloxKeys &key,
                                                        uint64_t subsequence, uint64_t block)
    {
        const U4 w = philox_block(key, subsequence, block);
        const double u = f64::u53(w.x, w.y, 0x1p-53);
        const double r = f64::sqrt_unclamped(f64::neg2log(u, m.t.log_tab));   // strictly positive argument (fast64.hpp)
        return __builtin_fma(r, f64::sin_bits_rotated<false>(w.z, w.w, m.t.sincos_tab, nullptr), acc);
    }
    // n == 1: z0 = r sin a = r (sin a' - cos a') / sqrt 2 with a' = a + pi/4 the rotated angle
    __device__ __forceinline__ static double head(const MathCtx<double> &m, const PhiloxKeys &key, uint64_t subsequence,
                                                   uint64_t block, uint32_t)
    {
        const U4 w = philox_block(key, subsequence, block);
        const double u = f64::u53(w.x, w.y, 0x1p-53);
        const double r = f64::sqrt_unclamped(f64::neg2log(u, m.t.log_tab));
        double cs;
        const double sn = f64::sin_bits_rotated<true>(w.z, w.w, m.t.sincos_tab, &cs);
        return r * (0.5 * (sn - cs));
    }
};

template <typename T, bool WINDOW>
__device__ __forceinline__ T payoff(T St, int32_t count, const StepConsts<T> &c)
{
    T pay = St - c.K;
    pay = pay > T(0) ? pay : T(0);
    if (WINDOW) pay = (count >= c.P1 && count <= c.P2) ? pay : T(0);
    return pay;
}

__device__ __forceinline__ float exp_of_logreturn(float S, float x, const MathCtx<float> &)
{
    return S * __builtin_amdgcn_exp2f(x);  // fp32 exponent constants carry log2(e)
}

__device__ __forceinline__ double exp_of_logreturn(double S, double y, const MathCtx<double> &m)
{
    PathState<double> ps = PathState<double>::start(S);   // fp64 exponent constants carry 65536 / ln 2
    ps.step(y, m);
    return ps.value(m);
}

// Simulates n_sim steps of one path in registers from (St, count) on the Philox stream
// (seed, subsequence) and returns its undiscounted payoff.  Step loop of
// inc/trajectories.cuh:144-148 (and the inner loops inc/nmc.cuh:55-59, :335-339).
//
// LOGSPACE = false: the reference's recurrence as written, St *= exp(drift + vol G) every step.
// LOGSPACE = true (the default of the in-register kernels): the same scheme carried in the logarithm — every
// step still draws its normal, but the path accumulates ln(St / S_start) and exponentiates once at
// the end; the barrier test B > St becomes ln(B / S_start) > ln(St / S_start).  Same mathematics,
// different rounding (~1e-14 relative in fp64); one add (or add + fma + compare) per step instead
// of an exp.
// WINDOW: the barrier count only grows, so once it is beyond P2 the payoff is 0 whatever follows
// (inc/trajectories.cuh:149); when that holds for every lane of the wavefront, the wavefront leaves the step
// loop (checked once per Philox block: one compare and a scalar branch).  The reference tests this only before
// the loop (inc/nmc.cuh:53, :330); the result is the same, the work is not: with the benchmark's bullet window
// (B = 120, P2 = 50 of 252 steps, hello.cu:11-13) a continuation path is over after ~50 steps.
// log_start: ln(St / c.S_start) in the exponent's units (0 when the path starts at c.S_start); read in WINDOW
// mode by the log-space loop and by the fp64 barrier test, which hold the barrier level as ln(B / c.S_start).
__device__ __forceinline__ float log_ratio(float a, float b) { return __builtin_amdgcn_logf(a / b); }  // log2
__device__ __forceinline__ double log_ratio(double a, double b) { return log(a / b) * f64::kExpScale; }

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// What one sample contributes: its undiscounted payoff and the control variable S_T.  With ANTI the
// sample is the antithetic pair (G, -G) of the same normals: both members are averaged.
template <typename T>
struct Sample {
    T pay;   // payoff (mean of the pair with ANTI)
    T ctrl;  // terminal price S_T (mean of the pair with ANTI); with WINDOW, the price where the loop stopped
    uint32_t steps_run;  // steps the wavefront executed (wave-uniform): n_sim unless the window let it stop early
    uint64_t live_steps; // of steps_run x 64 lane-steps, those of lanes whose window was still open (wave-uniform, WINDOW only)
};

// lanes of the wavefront that can still be paid (barrier count not beyond P2); 0: the window has closed for all
template <bool ANTI>
__device__ __forceinline__ uint32_t window_open_lanes(int32_t count, int32_t count2, int32_t P2)
{
    const bool open = ANTI ? (count <= P2 || count2 <= P2) : (count <= P2);
    return static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(open)));
}

// EARLY: leave the loop when the window has closed for the whole wavefront (off when the caller needs the
// terminal price itself, i.e. for the S_T control variate).
// The window-less log-space path of the pricing kernel: only the sum of a path's normals matters (PairSum).  A thread
// walks NP paths with consecutive subsequences at once, block by block — NP independent chains in one loop body (two
// chains run 1.6-3.2 % faster than one on one box, three and four slower: profiles/r03_pairsum_paths_ab.txt) — and
// acc[p] = (z_1 + ... + z_n) / kUnit of path p.  The kernel's MathCtx must hold the ROTATED sin/cos table
// (MathCtx::init<true>).
#ifndef MCAMD_PAIRSUM_PATHS   // overridable for that same-box comparison only
#define MCAMD_PAIRSUM_PATHS 2
#endif
constexpr int kPairSumPaths = MCAMD_PAIRSUM_PATHS;
static_assert(kPairSumPaths >= 1 && kPairSumPaths <= 4, "MCAMD_PAIRSUM_PATHS must be in [1, 4]");
template <typename T, int NP>
__device__ __forceinline__ void pair_sums_of_paths(const MathCtx<T> &m, const PhiloxKeys &seed, uint64_t subsequence0,
                                                   uint32_t n_sim, T (&acc)[NP])
{
    constexpr int NB = Normals<T>::kPerBlock;
    const uint32_t n_full = n_sim / NB;
    const uint32_t rem = n_sim - n_full * NB;
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = T(0);
    for (uint32_t k = 0; k < n_full; ++k) {
#pragma unroll
        for (int p = 0; p < NP; ++p) acc[p] = PairSum<T>::add_block(acc[p], m, seed, subsequence0 + p, k);
    }
    if (rem) {
#pragma unroll
        for (int p = 0; p < NP; ++p) acc[p] += PairSum<T>::head(m, seed, subsequence0 + p, n_full, rem);
    }
}

// What a window-less path contributes, from its pair-sum total: ln(S_T / S_in) = n drift + (vol kUnit) acc.
template <typename T, bool ANTI>
__device__ __forceinline__ Sample<T> sample_from_pair_sum(const StepConsts<T> &c, const MathCtx<T> &m, T acc, T S_in,
                                                          uint32_t n_sim)
{
    const T nd = c.drift * static_cast<T>(n_sim);
    const T vol_unit = c.vol * PairSum<T>::kUnit;
    const T St = exp_of_logreturn(S_in, fma_t(acc, vol_unit, nd), m);
    Sample<T> out;
    out.steps_run = n_sim;
    out.live_steps = 0;
    out.pay = payoff<T, false>(St, 0, c);
    out.ctrl = St;
    if (ANTI) {
        const T St2 = exp_of_logreturn(S_in, fma_t(-acc, vol_unit, nd), m);
        out.pay = T(0.5) * (out.pay + payoff<T, false>(St2, 0, c));
        out.ctrl = T(0.5) * (St + St2);
    }
    return out;
}

template <typename T, bool WINDOW, bool LOGSPACE, bool ANTI, bool EARLY = WINDOW>
__device__ __forceinline__ Sample<T> simulate_sample(const StepConsts<T> &c, const MathCtx<T> &m, const PhiloxKeys &seed,
                                                     uint64_t subsequence, T St, int32_t count, uint32_t n_sim,
                                                     T log_start = T(0))
{
    constexpr int NB = Normals<T>::kPerBlock;
    const uint32_t n_full = n_sim / NB;
    const uint32_t rem = n_sim - n_full * NB;
    T St2 = St;               // antithetic twin
    int32_t count2 = count;
    uint32_t steps_run = n_sim;
    bool rem_live = true;
    // lanes entering the next block with an open window (EARLY only; every active lane at the start)
    uint32_t open_lanes = EARLY ? static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(true))) : 0u;
    uint64_t live_steps = 0;   // up to 64 x n_sim
    // LOGSPACE: ln(St / S_ref) so far in exponent units
    T acc = WINDOW ? log_start : T(0), acc2 = acc;
    if (LOGSPACE) {
        // ln(St / S_ref) carried in the exponent's units: acc += x with x = drift + vol G the step's exponent (the
        // same Exponents the product form multiplies by; the twin's is 2 drift - x).  The barrier test B > St is
        // ln(B / S_start) > acc.
        Exponents<T> ex;
        const T two_drift = c.drift + c.drift;
        auto step = [&](T x) {
            acc += x;
            if (WINDOW) count += (c.logB > acc) ? 1 : 0;
            if (ANTI) {
                acc2 += two_drift - x;
                if (WINDOW) count2 += (c.logB > acc2) ? 1 : 0;
            }
        };
        for (uint32_t k = 0; k < n_full; ++k) {
            ex.fill(m, c, seed, subsequence, k);
#pragma unroll
            for (int j = 0; j < NB; ++j) step(ex.x[j]);
            if (WINDOW && EARLY) {
                live_steps += open_lanes * NB;
                open_lanes = window_open_lanes<ANTI>(count, count2, c.P2);
                if (open_lanes == 0) {
                    steps_run = (k + 1) * NB;
                    rem_live = false;
                    break;
                }
            }
        }
        if (rem && rem_live) {
            ex.fill(m, c, seed, subsequence, n_full);
#pragma unroll
            for (int j = 0; j < NB - 1; ++j)
                if (static_cast<uint32_t>(j) < rem) step(ex.x[j]);
        }
    } else {
        // the reference's recurrence: St *= exp(drift + vol G); the twin uses drift - vol G = 2 drift - x.
        // The barrier count needs St at every step (value()); a European path only at the end.
        Exponents<T> ex;
        const T two_drift = c.drift + c.drift;
        PathState<T> ps = PathState<T>::start(St);
        if (WINDOW) ps.arm_barrier(c.logB - log_start);   // ln(B / St) = ln(B / S_start) - ln(St / S_start)
        PathState<T> ps2 = ps;
        auto step = [&](T x) {
            ps.step(x, m);
            if (WINDOW) count += ps.below_barrier(c, m);
            if (ANTI) {
                ps2.step(two_drift - x, m);
                if (WINDOW) count2 += ps2.below_barrier(c, m);
            }
        };
        for (uint32_t k = 0; k < n_full; ++k) {
            ex.fill(m, c, seed, subsequence, k);
#pragma unroll
            for (int j = 0; j < NB; ++j) step(ex.x[j]);
            if (WINDOW && EARLY) {
                live_steps += open_lanes * NB;
                open_lanes = window_open_lanes<ANTI>(count, count2, c.P2);
                if (open_lanes == 0) {
                    steps_run = (k + 1) * NB;
                    rem_live = false;
                    break;
                }
            }
        }
        if (rem && rem_live) {
            ex.fill(m, c, seed, subsequence, n_full);
#pragma unroll
            for (int j = 0; j < NB - 1; ++j)
                if (static_cast<uint32_t>(j) < rem) step(ex.x[j]);
        }
        St = ps.value(m);
        if (ANTI) St2 = ps2.value(m);
    }
    if (LOGSPACE) {
        // acc = ln(St / S_ref) in exponent units; S_ref = c.S_start with a window (acc started at log_start), else the
        // start price itself (acc started at 0)
        const T S_ref = WINDOW ? c.S_start : St;
        St = exp_of_logreturn(S_ref, acc, m);
        if (ANTI) St2 = exp_of_logreturn(S_ref, acc2, m);
    }
    Sample<T> out;
    out.steps_run = steps_run;
    out.live_steps = live_steps + ((rem && rem_live) ? static_cast<uint64_t>(open_lanes) * rem : 0ull);
    out.pay = payoff<T, WINDOW>(St, count, c);
    out.ctrl = St;
    if (ANTI) {
        out.pay = T(0.5) * (out.pay + payoff<T, WINDOW>(St2, count2, c));
        out.ctrl = T(0.5) * (St + St2);
    }
    return out;
}

template <typename T, bool WINDOW, bool LOGSPACE>
__device__ __forceinline__ T simulate_path(const StepConsts<T> &c, const MathCtx<T> &m, const PhiloxKeys &seed,
                                           uint64_t subsequence, T St, int32_t count, uint32_t n_sim,
                                           T log_start = T(0), uint64_t *steps_run = nullptr,
                                           uint64_t *live_steps = nullptr)
{
    const Sample<T> s = simulate_sample<T, WINDOW, LOGSPACE, false>(c, m, seed, subsequence, St, count, n_sim, log_start);
    if (steps_run) *steps_run += s.steps_run;
    if (live_steps) *live_steps += s.live_steps;
    return s.pay;
}

// ---------------------------------------------------------------------------------------------
// Reductions.  Replaces the 1024-entry shared-memory tree + warp-32 shuffle inlined into every
// reference kernel (inc/trajectories.cuh:77-111 etc., inc/reduce.cuh): a wave64 shuffle
// reduction, one LDS slot per wave, finished by the first wave.  Inactive lanes contribute 0 —
// the work is predicated, never the reduction (the reference puts barriers inside
// `if (idx < N_PATHS)`).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// Sums (a, b) over the block; the result is valid in thread 0.  BLOCK must be a multiple of 64.
template <int BLOCK>
__device__ __forceinline__ void block_sum2(double &a, double &b)
{
    constexpr int kWaves = BLOCK / kWave;
    __shared__ double lds[2 * kWaves];
    a = wave_sum(a);
    b = wave_sum(b);
    if (kWaves == 1) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    if (lane == 0) {
        lds[2 * wave] = a;
        lds[2 * wave + 1] = b;
    }
    __syncthreads();
    if (wave == 0) {
        a = lane < kWaves ? lds[2 * lane] : 0.0;
        b = lane < kWaves ? lds[2 * lane + 1] : 0.0;
#pragma unroll
        for (int off = kWaves / 2; off > 0; off >>= 1) {
            a += __shfl_down(a, off, kWave);
            b += __shfl_down(b, off, kWave);
        }
    }
}

// Sums N values over the block (same scheme as block_sum2); results valid in thread 0.
template <int BLOCK, int N>
__device__ __forceinline__ void block_sumN(double (&v)[N])
{
    constexpr int kWaves = BLOCK / kWave;
    __shared__ double lds[N * kWaves];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = wave_sum(v[i]);
    if (kWaves == 1) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) lds[N * wave + i] = v[i];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            v[i] = lane < kWaves ? lds[N * lane + i] : 0.0;
#pragma unroll
            for (int off = kWaves / 2; off > 0; off >>= 1) v[i] += __shfl_down(v[i], off, kWave);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Finishing a grid.  The reference finishes inside the simulation kernel with one atomicAdd per block into a float
// (inc/trajectories.cuh:77-111): one launch, but the result depends on the order the blocks arrive in.  Here a block
// leaves one record (N doubles) in `partials`, and the records are summed in ONE fixed order (small_final_sum) by a
// single workgroup — either the last workgroup of the simulation grid to finish (grid_finish: one launch, as the
// reference) or a separate one-workgroup launch.  Same function, same order, same bits either way.
// ---------------------------------------------------------------------------------------------
#ifndef MCAMD_FINISH_ACC   // overridable for the same-box comparison of profiles/r03_finish_acc_ab.txt only
#define MCAMD_FINISH_ACC 4
#endif
static_assert(MCAMD_FINISH_ACC == 1 || MCAMD_FINISH_ACC == 2 || MCAMD_FINISH_ACC == 4 || MCAMD_FINISH_ACC == 8,
              "the accumulator sets are added pairwise: a power of two");
// Sums n_records records of N doubles with the BLOCK threads of one workgroup in a fixed order: thread t takes records
// t, t + BLOCK, ... round-robin into kAcc = 4 independent accumulator sets (more would push the pricing kernels past 80 vector registers, i.e. below six wavefronts per SIMD; four loads in flight per lane hide most of the L2
// latency of the lone workgroup), the sets are added pairwise, block_sumN finishes.  Result valid in thread 0.
template <int BLOCK, int N>
__device__ __forceinline__ void small_final_sum(const double *__restrict__ partials, uint32_t n_records, double (&v)[N])
{
    constexpr int kAcc = MCAMD_FINISH_ACC;
    double s[kAcc][N];
#pragma unroll
    for (int u = 0; u < kAcc; ++u)
#pragma unroll
        for (int k = 0; k < N; ++k) s[u][k] = 0.0;
    uint32_t i = threadIdx.x;
    for (; i + (kAcc - 1) * BLOCK < n_records; i += kAcc * BLOCK) {
#pragma unroll
        for (int u = 0; u < kAcc; ++u)
#pragma unroll
            for (int k = 0; k < N; ++k) s[u][k] += partials[static_cast<uint64_t>(i + u * BLOCK) * N + k];
    }
#pragma unroll
    for (int u = 0; u < kAcc - 1; ++u)   // the last, partial sweep keeps each record in the set the full sweeps give it
        if (i + u * BLOCK < n_records) {
#pragma unroll
            for (int k = 0; k < N; ++k) s[u][k] += partials[static_cast<uint64_t>(i + u * BLOCK) * N + k];
        }
#pragma unroll
    for (int w = 1; w < kAcc; w *= 2)
#pragma unroll
        for (int u = 0; u + w < kAcc; u += 2 * w)
#pragma unroll
            for (int k = 0; k < N; ++k) s[u][k] += s[u + w][k];
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = s[0][k];
    block_sumN<BLOCK, N>(v);
}

// Where a grid's final record goes: N doubles to out[0..N); with n_value >= 0 the 6-double statistics layout
// (out[N..5) = 0, out[5] = n_value) that one all-reduce carries.
__device__ __forceinline__ void write_final(double *__restrict__ out, const double *v, int n, double n_value)
{
    for (int k = 0; k < n; ++k) out[k] = v[k];
    if (n_value >= 0.0) {
        for (int k = n; k < 5; ++k) out[k] = 0.0;
        out[5] = n_value;
    }
}

struct GridFinish {
    double *out;            // final record (device memory, or pinned host memory the device can write)
    unsigned int *ticket;   // arrival counter, zero at launch (the last workgroup leaves it zero again); nullptr: the
                            // kernel only writes its block records and a separate launch sums them
    double n_value;
};

// Every thread of the workgroup calls this once, after its block sum (v valid in thread 0; contains barriers).
// Thread 0 publishes the block's record and takes a ticket; the workgroup whose ticket is the last one sums all
// records.  Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): producer = plain stores by ONE lane, agent-
// scope release fence, s_waitcnt vmcnt(0) (kept in inline asm: the compiler may drop its own wait after the fence),
// relaxed agent-scope atomic add; consumer = the lane whose add came last, agent-scope acquire fence, s_waitcnt,
// workgroup barrier, plain loads.  Per-XCD L2s are not coherent: the release writes the record back, the acquire
// drops what the reading CU may hold of the array from an earlier launch.
template <int BLOCK, int N>
__device__ __forceinline__ void grid_finish(double (&v)[N], double *__restrict__ partials, const GridFinish &f)
{
    __shared__ unsigned int s_last;
    if (threadIdx.x == 0) {
        unsigned int last = 0;
        if (f.ticket != nullptr) {
#pragma unroll
            for (int k = 0; k < N; ++k) partials[static_cast<uint64_t>(N) * blockIdx.x + k] = v[k];
            // (write-through `sc1` record stores + s_waitcnt in place of the fence measured the same: +-0.5 % in a
            // same-box A/B of 1M / 10M / 100M-path jobs, r03)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned int t = __hip_atomic_fetch_add(f.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (t == gridDim.x - 1) ? 1u : 0u;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) partials[static_cast<uint64_t>(N) * blockIdx.x + k] = v[k];
        }
        s_last = last;
    }
    __syncthreads();
    if (s_last == 0) return;   // workgroup-uniform
    double t[N];
    small_final_sum<BLOCK, N>(partials, gridDim.x, t);
    if (threadIdx.x == 0) {
        write_final(f.out, t, N, f.n_value);
        __hip_atomic_store(f.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
}

}  // namespace mcamd
