// host_fast64_check.cpp — CPU unit test of the hand-written fp64 math (csrc/fast64.hpp): compiled with
// g++ and run by tests/test_fast64.py.  Prints one JSON line with the worst errors found against
// long-double libm over random arguments drawn the way the kernels produce them.
#include "fast64.hpp"

#define MCAMD_TAB_DECL static const
#include "tables64.inc"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

using namespace mcamd::f64;

static double ulp_err(double got, long double want)
{
    if (want == 0.0L) return got == 0.0 ? 0.0 : 1e30;
    int e;
    std::frexp(static_cast<double>(want), &e);
    const long double ulp = std::ldexp(1.0L, e - 53);
    return static_cast<double>(fabsl(static_cast<long double>(got) - want) / ulp);
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? std::atol(argv[1]) : 4000000;
    const Tables T{reinterpret_cast<const D2 *>(kLogTab), reinterpret_cast<const D2 *>(kSinCosTab), kExp2Tab};
    std::mt19937_64 gen(12345);
    double e_log = 0, e_sqrt = 0, e_sin = 0, e_cos = 0, e_exp = 0, e_u = 0;
    const long double PI = 3.14159265358979323846264338327950288L;
    for (long i = 0; i < n; ++i) {
        const uint64_t a = gen(), b = gen();
        uint32_t x = static_cast<uint32_t>(a), y = static_cast<uint32_t>(a >> 32);
        const uint32_t z = static_cast<uint32_t>(b), w = static_cast<uint32_t>(b >> 32);
        if (i % 7 == 0) { y = 0xffffffffu; x |= 0xfffff000u; }  // u close to 1
        if (i % 11 == 0) { y &= 0xfffu; }                         // small u
        if (i == 1) { x = 0xffffffffu; y = 0xffffffffu; }         // u == 1 exactly
        if (i == 2) { x = 0; y = 0; }                             // u == 2^-53
        const double u = u53(x, y, 0x1p-53);
        const uint64_t v1 = static_cast<uint64_t>(x) ^ (static_cast<uint64_t>(y) << 21);
        const double u_ref = std::fma(static_cast<double>(v1), 0x1p-53, 0x1p-53);
        if (u != u_ref) e_u = 1;
        const double aa = neg2log(u, T.log_tab);
        const long double want_a = -2.0L * logl(static_cast<long double>(u));
        if (u == 1.0) { if (std::fabs(aa) > 1e-15) e_log = 1e30; }
        else { const double e = ulp_err(aa, want_a); if (e > e_log) e_log = e; }
        const double s = sqrt_pos(std::fmax(aa, 0.0));
        if (aa > 0) { const double e = ulp_err(s, sqrtl(static_cast<long double>(aa))); if (e > e_sqrt) e_sqrt = e; }
        const double q = u53(z, w, 0x1p-44);
        const uint64_t v2 = static_cast<uint64_t>(z) ^ (static_cast<uint64_t>(w) << 21);
        const double t_ref = std::fma(static_cast<double>(v2), 0x1p-52, 0x1p-52);
        if (q != 256.0 * t_ref) e_u = 2;
        double sn, cs;
        sincos_q(q, T.sincos_tab, sn, cs);
        const long double ang = PI * static_cast<long double>(t_ref);
        const double es = std::fabs(static_cast<double>(static_cast<long double>(sn) - sinl(ang)));
        const double ec = std::fabs(static_cast<double>(static_cast<long double>(cs) - cosl(ang)));
        if (es > e_sin) e_sin = es;
        if (ec > e_cos) e_cos = ec;
        // exp arguments: the GBM exponent range, plus a wide sweep
        const double xx = (i & 1) ? (static_cast<double>(static_cast<int64_t>(a)) * 0x1p-63) * 0.2
                                  : (static_cast<double>(static_cast<int64_t>(b)) * 0x1p-63) * 300.0;
        const double S = 37.0 + static_cast<double>(z) * 0x1p-32 * 200.0;
        const double got = mul_exp(S, xx, T.exp_tab);
        const double e = ulp_err(got, static_cast<long double>(S) * expl(static_cast<long double>(xx)));
        if (e > e_exp) e_exp = e;
    }
    std::printf("{\"n\": %ld, \"uniform_mismatch\": %g, \"neg2log_ulp\": %.3f, \"sqrt_ulp\": %.3f, \"sin_abs\": %.3g, "
                "\"cos_abs\": %.3g, \"mul_exp_ulp\": %.3f}\n", n, e_u, e_log, e_sqrt, e_sin, e_cos, e_exp);
    return 0;
}
