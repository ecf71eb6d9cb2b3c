// fast64.hpp — hand-written fp64 elementary functions for the pricing kernels' inner loop.
//
// Why: on gfx950 the in-register fp64 path is bound by VALU issue (every fp64 op costs 4 cycles per
// wave64, profiles/r01_valu_issue_costs.json), and the device libm's log / sincospi / exp spend a
// large share of their instructions on cases this loop cannot produce (denormals, NaN/Inf, huge
// arguments) and on table-free polynomials.  These versions use the argument ranges the
// Box-Muller / GBM step actually has, and move part of the work off the VALU onto the LDS pipe:
// three 512-entry tables (20 KB per workgroup) keep every polynomial at degree <= 4.
//
//   neg2log(u)        -2 ln u          u in [2^-53, 1]                 10 fp64 ops
//   sqrt_pos(a)       sqrt(a)          a >= 0 (clamped to >= 1e-300)   v_rsq_f64 + 8 fp64 ops
//   sincos_bits(z,w)  sin, cos(pi t)   t = (v2 + 1) 2^-52 from two Philox words: 11 fp64 ops, no int->fp convert
//   ExpAcc            running product of e^x_i kept as 2^(k/65536) * P: 6 fp64 ops per factor, no table
//                     lookup until the value is needed (exp_acc_value: 2 lookups, 3 multiplies, ldexp)
// Accuracy (tests/test_fast64.py, against long-double libm on 4M random arguments each):
// <= 2 ulp for neg2log, <= 1 ulp sqrt_pos, <= 2.5e-16 absolute for sin / cos, <= 4.5 ulp for a single
// exp factor S e^x (|x| <= 1) and <= 64 ulp (observed 36) for a 252-factor product, whose rounding
// random-walks like any fp64 product recurrence of that length.
//
// The header compiles for the host too (plain g++), with the tables as ordinary arrays and the
// hardware reciprocal square root emulated at reduced precision, so the numerics are unit-tested
// on the CPU without a GPU.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MC_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define MC_HD inline
#endif

namespace mcamd {
namespace f64 {

struct alignas(16) D2 {
    double a, b;
};

// Pointers to the three tables: LDS copies inside a kernel, the static arrays on the host.
struct Tables {
    const D2 *log_tab;        // {-2/c_i, -2 ln c_i}
    const D2 *sincos_tab;     // {sin, cos}(2 pi (j + 1/2) / 512): nodes at the MIDDLE of each of the 512 arcs
    const double *exp_hi_tab; // 2^(i/256),   i in [0, 256)
    const double *exp_lo_tab; // 2^(i/65536), i in [0, 256)
};

// One 16-byte table entry {a, b}.  Experiment (-DMCAMD_LDS_PLANES, profiles/r03_lds_planes_ab.txt): the same table as
// two 8-byte planes, a[0..N) then b[0..N), read with two ds_read_b64 instead of one ds_read_b128 — VERDICT r02 asked
// whether that lowers the LDS bank conflicts of the random lookups enough to show in the time.  Not the shipped layout.
MC_HD D2 table_entry(const D2 *tab, uint32_t byte_offset_of_entry)
{
#if defined(MCAMD_LDS_PLANES)
    const char *base = reinterpret_cast<const char *>(tab);
    const uint32_t half = byte_offset_of_entry >> 1;   // entry i sits at 16 i: plane element i at 8 i
    return D2{*reinterpret_cast<const double *>(base + half),
              *reinterpret_cast<const double *>(base + 512 * 8 + half)};
#else
    return *reinterpret_cast<const D2 *>(reinterpret_cast<const char *>(tab) + byte_offset_of_entry);
#endif
}

MC_HD uint32_t hi32(double x)
{
    uint64_t b;
    __builtin_memcpy(&b, &x, 8);
    return static_cast<uint32_t>(b >> 32);
}

MC_HD uint32_t lo32(double x)
{
    uint64_t b;
    __builtin_memcpy(&b, &x, 8);
    return static_cast<uint32_t>(b);
}

MC_HD double make_double(uint32_t lo, uint32_t hi)
{
    const uint64_t b = (static_cast<uint64_t>(hi) << 32) | lo;
    double x;
    __builtin_memcpy(&x, &b, 8);
    return x;
}

MC_HD double rsq_seed(double a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsq(a);  // v_rsq_f64
#else
    return static_cast<double>(static_cast<float>(1.0 / __builtin_sqrt(a)));  // ~24-bit seed, like the hardware op
#endif
}

// Three-address fused multiply-adds for the places where hipcc (ROCm 7.2) otherwise picks the two-address
// v_fmac_f64 and has to copy a loop-invariant addend into a fresh accumulator first (a v_mov_b64, or two
// v_mov_b32 from SGPRs, per use: ~6 % of the fp64 step loop).  One VALU instruction each, register operands
// only; their inputs never come straight from a transcendental op, so no wait state is owed inside.
//   fma_usv(a, k, c)  a * k + c         k wave-uniform, c a value kept in VGPRs (loop-invariant constant)
//   fma_vvs(a, b, k)  a * b + k         k wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)
MC_HD double fma_usv(double a, double k, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(c));
    return d;
}
MC_HD double fma_vvs(double a, double b, double k)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k));
    return d;
}
//   fma_vvv(a, b, c)  a * b + c         c a loop-invariant constant kept in VGPRs
MC_HD double fma_vvv(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
#else
MC_HD double fma_usv(double a, double k, double c) { return __builtin_fma(a, k, c); }
MC_HD double fma_vvs(double a, double b, double k) { return __builtin_fma(a, b, k); }
MC_HD double fma_vvv(double a, double b, double c) { return __builtin_fma(a, b, c); }
#endif
// Hides a value's provenance from the optimiser (no instruction).  Used where ROCm 7.2's instruction selection
// otherwise rewrites "(lo32(M * c) >> 11) & 0xfff" on a Philox output word into a second 32-bit multiply.
MC_HD uint32_t opaque(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(x));
#endif
    return x;
}

// (v + 1) * c with v = x ^ (y << 21) the 53-bit integer rocRAND builds from two Philox words
// (rocrand_normal.h box_muller_double).  v = hi 2^32 + lo is exact in a double (v < 2^53), and
// fma(v, c, c) is rocRAND's own expression; with c a power of two the result is exact.
//   u = (v+1) 2^-53  -> c = 2^-53      (Box-Muller radius uniform, in (0, 1])
//   q = (v+1) 2^-44  -> c = 2^-44      (256 x the angle uniform t = (v+1) 2^-52 in (0, 2])
// y << 21 and y >> 11 from ONE instruction: the two halves of the 64-bit product y * 2^21 (v_mad_u64_u32, 4 issue
// cycles; v_lshlrev_b32 alone is 4 on gfx950 and v_lshrrev_b32 2 more — profiles/r02_operand_costs.txt).
struct Split21 {
    uint32_t shl21, shr11;
};
MC_HD Split21 split21(uint32_t y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t p, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(carry) : "v"(y), "s"(0x200000u));
    // opaque: otherwise the uint32 -> double conversion of the upper half is widened to a 64-bit conversion
    return Split21{static_cast<uint32_t>(p), opaque(static_cast<uint32_t>(p >> 32))};
#else
    return Split21{y << 21, y >> 11};
#endif
}

MC_HD double u53(uint32_t x, uint32_t y, double c)
{
    const Split21 sy = split21(y);
    const uint32_t lo = x ^ sy.shl21;
    const uint32_t hi = sy.shr11;
    // (lo + 1) c without a conversion: lo sits in the mantissa of 2^52, and fma(2^52 + lo, c, c - 2^52 c) is exact
    const double l = fma_usv(make_double(lo, 0x43300000u), c, c - 0x1p52 * c);
    // + hi 2^32 c: exact too, the sum is (v + 1) c with v + 1 <= 2^53
    return __builtin_fma(static_cast<double>(hi), 0x1p32 * c, l);
}

#include "tables64_consts.inc"

// -2 ln(u) for u in [2^-53, 1].  u = 2^k z, z in [0.6875, 1.375); chunk i (top 9 bits of z's bit pattern
// above 0.6875) selects c_i; t = -2 (z / c_i - 1) is tiny (|t| < 2^-9), and
// -2 ln(1 - t/2) = t + t^2/4 + t^3/12 + t^4/32 + t^5/80 (the dropped t^6/192 is < 2.9e-19 absolute, i.e. below
// 0.7 ulp of the result even in the one chunk, next to u = 1, where the table term is zero).
MC_HD double neg2log(double u, const D2 *tab)
{
    const uint32_t hx = hi32(u);
    const uint32_t tmp = hx - 0x3fe60000u;
    const uint32_t i = (tmp >> 11) & 511u;
    const int32_t k = static_cast<int32_t>(tmp) >> 20;
    const double z = make_double(lo32(u), hx - (tmp & 0xfff00000u));
    const D2 e = table_entry(tab, i << 4);
    const double t = __builtin_fma(z, e.a, 2.0);
    const double w = __builtin_fma(static_cast<double>(k), kM2Ln2, e.b);
    double q = fma_usv(t, 1.0 / 80.0, 1.0 / 32.0);
    q = fma_vvv(t, q, 1.0 / 12.0);
    q = fma_vvv(t, q, 0.25);
    return w + __builtin_fma(t * t, q, t);
}

// sqrt(a), a >= 0: hardware reciprocal-sqrt seed, one coupled Newton step, one correction.
MC_HD double sqrt_pos(double a)
{
    a = __builtin_fmax(a, 1e-300);  // a == 0 (u == 1, probability 2^-53) must not reach rsq
    const double y = rsq_seed(a);
    double g = a * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, a);
    return __builtin_fma(d, h, g);
}

// k * sqrt(a), a > 0, for the Box-Muller radius times the step volatility: hardware reciprocal-sqrt seed y
// (relative error e0 <= 2^-22), e = 1 - a y^2, sqrt(a) = a y (1 - e)^(-1/2) = a y (1 + e/2 + 3 e^2/8 + O(e^3)):
// one cubic step, six fp64 operations including the scaling (sqrt_pos + multiply: eight); <= 1.5 ulp.
// No clamp: a = neg2log(u) is strictly positive for every u in (0, 1] (the table entry that serves u = 1 is biased
// by 4e-18, tools/gen_tables64.py), so the seed is finite.
MC_HD double sqrt_scaled(double a, double k)
{
    const double y = rsq_seed(a);
    const double g = a * y;
    const double e = __builtin_fma(-y, g, 1.0);
    const double t = e * fma_usv(e, 0.375, 0.5);
    const double gk = g * k;
    return __builtin_fma(gk, t, gk);
}

// sqrt(a), a > 0 (same argument range and seed as sqrt_scaled, no scaling): five fp64 operations.
MC_HD double sqrt_unclamped(double a)
{
    const double y = rsq_seed(a);
    const double g = a * y;
    const double e = __builtin_fma(-y, g, 1.0);
    const double t = e * fma_usv(e, 0.375, 0.5);
    return __builtin_fma(g, t, g);
}

// sin and cos of the Box-Muller angle pi * t, t = (v2 + 1) 2^-52 in (0, 2], straight from the two Philox words
// (z, w) rocRAND builds v2 from (v2 = z ^ (w << 21) in the low word, w >> 11 above it: 53 bits).
//   pi t = (2 pi / 512)(j + 1/2 + f),   j = v2 >> 44 (the top 9 bits of w),
//   f = ((v2 mod 2^44) + 1) 2^-44 - 1/2  in (-1/2, 1/2].
// f needs no integer-to-double conversion: the 44 low bits of v2 are dropped into the mantissa of 2^52
// (D = 2^52 + (v2 mod 2^44), one and-or on the high word) and f = fma(D, 2^-44, 2^-44 - 256.5) is exact.
// The table holds sin/cos at the arc MIDPOINTS, so the offset d = f (2 pi / 512) has |d| <= pi/512 and
// sin d = d - d^3/6 + d^5/120, cos d = 1 - d^2/2 + d^4/24 are exact to < 1e-16; the powers of 2 pi / 512 are
// folded into the coefficients (kSinF1.., kCosF2..), so the polynomials run on f itself.
MC_HD void sincos_bits(uint32_t z, uint32_t w, const D2 *tab, double &s, double &c)
{
    const Split21 sw = split21(w);
    const uint32_t lo = z ^ sw.shl21;
#if defined(__HIP_DEVICE_COMPILE__)
    // (a & 0xfff) | 0x43300000 as one full-rate v_bitop3_b32 on vector-register constants (v_and_or_b32 and v_bfe_u32
    // issue in 4 cycles, and so does any full-rate instruction reading a scalar register)
    uint32_t hi;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xea" : "=v"(hi) : "v"(sw.shr11), "v"(0xfffu), "v"(0x43300000u));
#else
    const uint32_t hi = (sw.shr11 & 0xfffu) | 0x43300000u;
#endif
    const double f = fma_usv(make_double(lo, hi), 0x1p-44, 0x1p-44 - 256.5);
    const D2 e = table_entry(tab, (sw.shr11 >> 8) & 0x1ff0u);
    const double ff = f * f;
    double sp = fma_usv(ff, kSinF5, kSinF3);
    sp = fma_vvs(ff, sp, kSinF1);
    const double sd = f * sp;
    const double cp = fma_usv(ff, kCosF4, kCosF2);
    const double cd = __builtin_fma(ff, cp, 1.0);
    s = __builtin_fma(e.a, cd, e.b * sd);
    c = __builtin_fma(e.b, cd, -(e.a * sd));
}

// The sine alone, from a table ROTATED by an eighth of a turn (entry j holds {sin, cos} of arc j + N/8): with it
// this returns sin(pi t + pi/4) = (sin(pi t) + cos(pi t)) / sqrt 2 — all a path needs of a Box-Muller pair when only
// the SUM of its two normals matters (mc_device.hpp PairSum).  Same reduction and polynomials as sincos_bits, one
// multiply and one fma fewer; cos_out (nullable at compile time through the template) is the cosine of that rotated
// angle, for the callers that must split the pair after all (a path's odd last step).
template <bool WANT_COS>
MC_HD double sin_bits_rotated(uint32_t z, uint32_t w, const D2 *rot_tab, double *cos_out)
{
    const Split21 sw = split21(w);
    const uint32_t lo = z ^ sw.shl21;
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t hi;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xea" : "=v"(hi) : "v"(sw.shr11), "v"(0xfffu), "v"(0x43300000u));
#else
    const uint32_t hi = (sw.shr11 & 0xfffu) | 0x43300000u;
#endif
    const double f = fma_usv(make_double(lo, hi), 0x1p-44, 0x1p-44 - 256.5);
    const D2 e = table_entry(rot_tab, (sw.shr11 >> 8) & 0x1ff0u);
    const double ff = f * f;
    double sp = fma_usv(ff, kSinF5, kSinF3);
    sp = fma_vvs(ff, sp, kSinF1);
    const double sd = f * sp;
    const double cp = fma_usv(ff, kCosF4, kCosF2);
    const double cd = __builtin_fma(ff, cp, 1.0);
    if (WANT_COS) *cos_out = __builtin_fma(e.b, cd, -(e.a * sd));
    return __builtin_fma(e.a, cd, e.b * sd);
}

// ---------------------------------------------------------------------------------------------
// Running product of exponentials, S * e^(x_1) * e^(x_2) * ... — the GBM recurrence St *= exp(x)
// (inc/trajectories.cuh:146) — kept in factored form 2^(k / 65536) * P:
//   each factor:  y = x * 65536 / ln 2 (the caller folds that scale into its drift and volatility constants);
//                 n = rint(y), rr = y - n exact, |rr| <= 1/2;  e^x = 2^(n/65536) * e^(rr ln2/65536);
//                 k += n (integer, exact);  P *= 1 + rr (c1 + c2 rr)   with c1 = ln2/65536, c2 = c1^2/2
//                 (|r| = |rr c1| <= 5.3e-6, so the dropped r^3/6 is < 2.5e-17: a quarter ulp);
//   the value:    ldexp(((S * 2^(hi/256)) * 2^(lo/65536)) * P, k >> 16),  hi = bits 15..8 of k, lo = bits 7..0.
// The power-of-two part of every factor is accumulated exactly, in an integer, so a path that only needs its
// terminal price (European payoff) does six fp64 operations per step and looks nothing up until the end; a
// path that needs St at every step (barrier count, trajectory store) evaluates the value each step — the
// same expression, so the last stored price and the in-register terminal price are the same bits.
// Range: |k| < 2^31 over the whole path (the host checks n_steps * max|y|, capi.cpp); ldexp saturates to
// inf / 0 like exp does.
// ---------------------------------------------------------------------------------------------
constexpr int kExpBits = 16;

struct ExpAcc {
    double P;
    int32_t k;
};

MC_HD ExpAcc exp_acc_init() { return ExpAcc{1.0, 0}; }

// multiplies the running product by e^x, given y = x * kExpScale.  c1 = kExpC1, passed in so that a kernel
// can keep it in vector registers across its step loop (exp_c1_resident).  Returns rint(y) as a double (what was
// added to k), for callers that also track the exponent in floating point (the barrier test of mc_device.hpp).
MC_HD double exp_acc_mul(ExpAcc &a, double y, double c1 = kExpC1)
{
    // round-to-nearest by adding 1.5 * 2^52: the integer lands in the low mantissa word (|y| < 2^31)
    const double ks = y + 0x1.8p52;
    const double kd = ks - 0x1.8p52;
    const double rr = y - kd;
    a.k += static_cast<int32_t>(lo32(ks));
    const double t = rr * fma_usv(rr, kExpC2, c1);
    a.P = __builtin_fma(a.P, t, a.P);
    return kd;
}

// Largest |ln P - (P - 1)| * kExpScale a path of n factors can reach: every factor is 1 + t with
// |t| <= (1/2) kExpC1 (1 + 1e-5), so |ln P| <= L = n * 5.295e-6 and |ln P - (P - 1)| <= 0.55 (e^L - 1)^2 for L <= 0.3.
// Used as the half-width of the band in which the cheap barrier test hands over to the exact one; +inf (always
// exact) for paths too long for the bound.
inline double exp_acc_window_delta(uint32_t n_factors)
{
    const double L = static_cast<double>(n_factors) * 5.295e-6;
    if (!(L <= 0.3)) return __builtin_huge_val();
    const double e = __builtin_expm1(L);
    return 0.55 * e * e * kExpScale + 1e-6;
}

// kExpC1 in vector registers, opaque to rematerialisation: without this hipcc (ROCm 7.2) re-creates the constant
// from scalar registers in front of every use inside the step loop (one v_mov_b64 per iteration).
MC_HD double exp_c1_resident()
{
#if defined(__HIP_DEVICE_COMPILE__)
    double c;
    asm volatile("v_mov_b64 %0, %1" : "=v"(c) : "s"(kExpC1));
    return c;
#else
    return kExpC1;
#endif
}

MC_HD double exp_acc_value(double S, const ExpAcc &a, const double *hi_tab, const double *lo_tab)
{
    const uint32_t k = static_cast<uint32_t>(a.k);
    const double th = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(hi_tab) + ((k >> 5) & 0x7f8u));
    const double tl = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(lo_tab) + ((k << 3) & 0x7f8u));
    return __builtin_ldexp(((S * th) * tl) * a.P, a.k >> kExpBits);
}

// S * e^x for one natural-units exponent (the log-space mode's single exponentiation, tests)
MC_HD double mul_exp(double S, double x, const double *hi_tab, const double *lo_tab)
{
    ExpAcc a = exp_acc_init();
    exp_acc_mul(a, x * kExpScale);
    return exp_acc_value(S, a, hi_tab, lo_tab);
}

}  // namespace f64
}  // namespace mcamd
