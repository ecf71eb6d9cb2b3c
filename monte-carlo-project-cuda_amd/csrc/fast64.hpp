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
//   sincos_q(q)       sin, cos(pi q/256) q = 256 t, t in (0, 2]        14 fp64 ops
//   mul_exp(S, x)     S * e^x          |x| < 700                       10 fp64 ops
// Accuracy (tests/test_fast64.py, against long-double libm on 4M random arguments each):
// <= 2 ulp for neg2log / mul_exp, <= 1 ulp sqrt_pos, <= 2e-16 absolute for sin / cos.
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
    const D2 *log_tab;     // {-2/c_i, -2 ln c_i}
    const D2 *sincos_tab;  // {sin, cos}(2 pi j / 512)
    const double *exp_tab; // 2^(j/512)
};

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
//   fma_us(a, k)      a * k + k         k wave-uniform (SGPR pair used twice)
//   fma_usv(a, k, c)  a * k + c         k wave-uniform, c a value kept in VGPRs (loop-invariant constant)
//   fma_vvs(a, b, k)  a * b + k         k wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)
MC_HD double fma_us(double a, double k)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %2" : "=v"(d) : "v"(a), "s"(k));
    return d;
}
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
#else
MC_HD double fma_us(double a, double k) { return __builtin_fma(a, k, k); }
MC_HD double fma_usv(double a, double k, double c) { return __builtin_fma(a, k, c); }
MC_HD double fma_vvs(double a, double b, double k) { return __builtin_fma(a, b, k); }
#endif

// (v + 1) * c with v = x ^ (y << 21) the 53-bit integer rocRAND builds from two Philox words
// (rocrand_normal.h box_muller_double).  v = hi 2^32 + lo is exact in a double (v < 2^53), and
// fma(v, c, c) is rocRAND's own expression; with c a power of two the result is exact.
//   u = (v+1) 2^-53  -> c = 2^-53      (Box-Muller radius uniform, in (0, 1])
//   q = (v+1) 2^-44  -> c = 2^-44      (256 x the angle uniform t = (v+1) 2^-52 in (0, 2])
MC_HD double u53(uint32_t x, uint32_t y, double c)
{
    const uint32_t lo = x ^ (y << 21);
    const uint32_t hi = y >> 11;
    const double v = __builtin_fma(static_cast<double>(hi), 0x1p32, static_cast<double>(lo));
    return fma_us(v, c);
}

#include "tables64_consts.inc"

// -2 ln(u) for u in [2^-53, 1].  u = 2^k z, z in [0.6875, 1.375); chunk i (top 9 bits of z's bit pattern
// above 0.6875) selects c_i; t = -2 (z / c_i - 1) is tiny (|t| < 2^-9), and
// -2 ln(1 - t/2) = t + t^2/4 + t^3/12 + t^4/32 + t^5/80 + t^6/192 (next term < 1.2e-19 relative).
MC_HD double neg2log(double u, const D2 *tab)
{
    const uint32_t hx = hi32(u);
    const uint32_t tmp = hx - 0x3fe60000u;
    const uint32_t i = (tmp >> 11) & 511u;
    const int32_t k = static_cast<int32_t>(tmp) >> 20;
    const double z = make_double(lo32(u), hx - (tmp & 0xfff00000u));
    const D2 e = tab[i];
    const double t = __builtin_fma(z, e.a, 2.0);
    const double w = __builtin_fma(static_cast<double>(k), kM2Ln2, e.b);
    double q = fma_usv(t, 1.0 / 192.0, 1.0 / 80.0);
    q = __builtin_fma(t, q, 1.0 / 32.0);
    q = __builtin_fma(t, q, 1.0 / 12.0);
    q = __builtin_fma(t, q, 0.25);
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

// sin and cos of pi * t given q = 256 t (exact), t in (0, 2]: angle = (2 pi / 512)(j + f),
// j = rint(q), f = q - j in [-1/2, 1/2]; the table gives sin/cos of the node, and for |d| <= pi/512
// sin d = d + d^3 (-1/6 + d^2/120), cos d = 1 + d^2 (-1/2 + d^2/24) are exact to < 1e-16.
MC_HD void sincos_q(double q, const D2 *tab, double &s, double &c)
{
    const double j = __builtin_rint(q);
    const double f = q - j;
    const int32_t ji = static_cast<int32_t>(j);
    const D2 e = tab[ji & 511];
    const double d = f * kTwoPiOverN;
    const double z = d * d;
    const double sp = fma_usv(z, 1.0 / 120.0, -1.0 / 6.0);
    const double sd = __builtin_fma(d * z, sp, d);
    const double cp = __builtin_fma(z, 1.0 / 24.0, -0.5);
    const double cd = __builtin_fma(z, cp, 1.0);
    s = __builtin_fma(e.a, cd, e.b * sd);
    c = __builtin_fma(e.b, cd, -(e.a * sd));
}

// S * exp(x): x = (k / 512) ln 2 + r, |r| <= ln2 / 1024; 2^(k/512) = 2^(k >> 9) * table[k & 511];
// e^r = 1 + r + r^2 (1/2 + r/6 + r^2/24) (next term r^5/120 < 1.3e-18).
MC_HD double mul_exp(double S, double x, const double *tab)
{
    // round-to-nearest by adding 1.5 * 2^52: the integer lands in the low mantissa word (|x| < 2^20)
    const double ks = __builtin_fma(x, kNOverLn2, 0x1.8p52);
    const int32_t ki = static_cast<int32_t>(lo32(ks));
    const double kd = ks - 0x1.8p52;
    double r = __builtin_fma(kd, -kLn2OverN_hi, x);
    r = __builtin_fma(kd, -kLn2OverN_lo, r);
    const double tv = tab[ki & 511];
    const uint32_t bump = (static_cast<uint32_t>(ki) & 0xfffffe00u) << 11;  // (k >> 9) << 20
    const double sc = make_double(lo32(tv), hi32(tv) + bump);
    double p = fma_usv(r, 1.0 / 24.0, 1.0 / 6.0);
    p = __builtin_fma(r, p, 0.5);
    const double tmp = __builtin_fma(r * r, p, r);
    const double Ss = S * sc;
    return __builtin_fma(Ss, tmp, Ss);
}

}  // namespace f64
}  // namespace mcamd
