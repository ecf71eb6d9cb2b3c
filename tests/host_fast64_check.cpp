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
    const Tables T{reinterpret_cast<const D2 *>(kLogTab), reinterpret_cast<const D2 *>(kSinCosTab), kExpHiTab, kExpLoTab};
    std::mt19937_64 gen(12345);
    // the sin/cos table as the pair-sum loop holds it: rotated by an eighth of a turn (MathCtx<double>::init<true>)
    static D2 rot[MCAMD_TAB_N];
    for (int i = 0; i < MCAMD_TAB_N; ++i) {
        const int j = (i + MCAMD_TAB_N / 8) & (MCAMD_TAB_N - 1);
        rot[i] = D2{kSinCosTab[j][0], kSinCosTab[j][1]};
    }
    double e_squ = 0, e_rot = 0, e_rotc = 0, e_pair = 0;
    double e_log = 0, e_sqrt = 0, e_sin = 0, e_cos = 0, e_exp = 0, e_u = 0, e_prod = 0, e_wide = 0, e_sqs = 0, e_band = 0, e_pos = 0;
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
        if (!(aa > 0.0)) e_pos = 1.0;   // the radius path divides by sqrt(aa): strictly positive for every u in (0, 1]
        const long double want_a = -2.0L * logl(static_cast<long double>(u));
        if (u == 1.0) { if (std::fabs(aa) > 1e-15) e_log = 1e30; }
        else { const double e = ulp_err(aa, want_a); if (e > e_log) e_log = e; }
        const double s = sqrt_pos(std::fmax(aa, 0.0));
        if (aa > 0) { const double e = ulp_err(s, sqrtl(static_cast<long double>(aa))); if (e > e_sqrt) e_sqrt = e; }
        if (aa > 0) {
            const double kk = 0.0126 * (1.0 + static_cast<double>(z & 1023u));   // a step volatility, any binade
            const double e = ulp_err(sqrt_scaled(aa, kk), static_cast<long double>(kk) * sqrtl(static_cast<long double>(aa)));
            if (e > e_sqs) e_sqs = e;
        }
        if (aa > 0) { const double e = ulp_err(sqrt_unclamped(aa), sqrtl(static_cast<long double>(aa))); if (e > e_squ) e_squ = e; }
        uint32_t zz = z, ww = w;
        if (i % 13 == 0) { zz = 0xffffffffu; ww |= 0x7fffffu; }   // top of an arc: f = +1/2
        if (i % 17 == 0) { zz = 0; ww &= ~0x7fffffu; }            // bottom of an arc
        if (i == 3) { zz = 0xffffffffu; ww = 0xffffffffu; }       // t == 2 exactly
        const uint64_t v2 = static_cast<uint64_t>(zz) ^ (static_cast<uint64_t>(ww) << 21);
        const double t_ref = std::fma(static_cast<double>(v2), 0x1p-52, 0x1p-52);   // rocRAND's angle uniform
        double sn, cs;
        sincos_bits(zz, ww, T.sincos_tab, sn, cs);
        const long double ang = PI * static_cast<long double>(t_ref);
        const double es = std::fabs(static_cast<double>(static_cast<long double>(sn) - sinl(ang)));
        const double ec = std::fabs(static_cast<double>(static_cast<long double>(cs) - cosl(ang)));
        if (es > e_sin) e_sin = es;
        if (ec > e_cos) e_cos = ec;
        // the rotated-table sine: sin(a + pi/4) = (sin a + cos a) / sqrt 2, and the cosine of the same rotated angle
        double rc;
        const double rs = sin_bits_rotated<true>(zz, ww, rot, &rc);
        const long double SQ2 = 1.41421356237309504880168872420969808L;
        const double er = std::fabs(static_cast<double>(static_cast<long double>(rs) - (sinl(ang) + cosl(ang)) / SQ2));
        const double erc = std::fabs(static_cast<double>(static_cast<long double>(rc) - (cosl(ang) - sinl(ang)) / SQ2));
        if (er > e_rot) e_rot = er;
        if (erc > e_rotc) e_rotc = erc;
        if (sin_bits_rotated<false>(zz, ww, rot, nullptr) != rs) e_rot = 1.0;
        // a whole pair sum as PairSum<double> forms it, against the sum of the two normals, relative to the radius
        if (aa > 0) {
            const double r = sqrt_unclamped(aa);
            const long double want = sqrtl(want_a) * (sinl(ang) + cosl(ang));
            const double ep = std::fabs(static_cast<double>(static_cast<long double>(r * rs) * SQ2 - want)) / (1.0 + static_cast<double>(sqrtl(want_a)));
            if (ep > e_pair) e_pair = ep;
        }
        // exp arguments: the GBM exponent range, plus a wide sweep
        const double xx = (i & 1) ? (static_cast<double>(static_cast<int64_t>(a)) * 0x1p-63) * 0.2
                                  : (static_cast<double>(static_cast<int64_t>(b)) * 0x1p-63) * 300.0;
        const double S = 37.0 + static_cast<double>(z) * 0x1p-32 * 200.0;
        const double got = mul_exp(S, xx, T.exp_hi_tab, T.exp_lo_tab);
        const double e = ulp_err(got, static_cast<long double>(S) * expl(static_cast<long double>(xx)));
        if (std::fabs(xx) <= 1.0) { if (e > e_exp) e_exp = e; }
        else if (e / std::fabs(xx) > e_wide) e_wide = e / std::fabs(xx);   // the exponent's own rounding scales with |x|
        // a 252-factor running product, the GBM recurrence: exponents of the benchmark's size, sum kept in long double
        if (i % 64 == 0) {
            ExpAcc acc = exp_acc_init();
            long double sum = 0.0L;
            std::mt19937_64 g2(a);
            for (int k = 0; k < 252; ++k) {
                const double xk = (static_cast<double>(static_cast<int64_t>(g2())) * 0x1p-63) * 0.06 + 1.5e-4;
                const double yk = xk * kExpScale;
                exp_acc_mul(acc, yk);
                sum += static_cast<long double>(yk);   // the product is exact in y: compare in the same units
            }
            const double gotp = exp_acc_value(S, acc, T.exp_hi_tab, T.exp_lo_tab);
            const long double wantp = static_cast<long double>(S) *
                                      expl(sum * (0.693147180559945309417232121458176568L / 65536.0L));
            const double ep = ulp_err(gotp, wantp);
            if (ep > e_prod) e_prod = ep;
            // the cheap barrier test of mc_device.hpp reads log2(product) * 65536 as k + (P - 1) kExpScale; its
            // distance from the true value must stay inside the band exp_acc_window_delta() hands to the exact test
            const long double true_units = sum;                                         // log2(prod) * 65536, exactly
            const long double approx_units = static_cast<long double>(acc.k) + (static_cast<long double>(acc.P) - 1.0L) * kExpScale;
            const double band = static_cast<double>(fabsl(approx_units - true_units)) / exp_acc_window_delta(252);
            if (band > e_band) e_band = band;
        }
    }
    // the top of the uniform's range, one by one: u = 1 - j 2^-53, j = 0 .. 2^16 (the chunk that holds u = 1 and its
    // neighbour), and the values around every power of two, where the chunk index wraps
    for (uint64_t j = 0; j < 65536; ++j) {
        const uint64_t v = (1ull << 53) - 1 - j;
        const double u = std::fma(static_cast<double>(v), 0x1p-53, 0x1p-53);
        const double aa = neg2log(u, T.log_tab);
        if (!(aa > 0.0) || !(sqrt_scaled(aa, 0.0126) >= 0.0)) e_pos = 1.0;
    }
    for (int e = 1; e <= 52; ++e)
        for (int64_t d = -4; d <= 4; ++d) {
            const int64_t v = (int64_t(1) << e) + d;
            if (v < 0) continue;
            const double u = std::fma(static_cast<double>(v), 0x1p-53, 0x1p-53);
            if (!(neg2log(u, T.log_tab) > 0.0)) e_pos = 1.0;
        }
    // saturation instead of wrap-around: a huge exponent gives inf / 0, never a finite wrong value
    {
        const double big = mul_exp(1.0, 800.0, T.exp_hi_tab, T.exp_lo_tab), tiny = mul_exp(1.0, -800.0, T.exp_hi_tab, T.exp_lo_tab);
        if (!std::isinf(big) || tiny != 0.0) e_exp = 1e30;
    }
    std::printf("{\"n\": %ld, \"uniform_mismatch\": %g, \"neg2log_ulp\": %.3f, \"sqrt_ulp\": %.3f, \"sin_abs\": %.3g, "
                "\"cos_abs\": %.3g, \"mul_exp_ulp\": %.3f, \"product252_ulp\": %.3f, \"mul_exp_wide_ulp_per_unit_x\": %.3f, \"sqrt_scaled_ulp\": %.3f, \"barrier_band_used\": %.4f, \"neg2log_nonpositive\": %g, "
                "\"sqrt_unclamped_ulp\": %.3f, \"sin_rotated_abs\": %.3g, \"cos_rotated_abs\": %.3g, \"pair_sum_rel\": %.3g}\n",
                n, e_u, e_log, e_sqrt, e_sin, e_cos, e_exp, e_prod, e_wide, e_sqs, e_band, e_pos, e_squ, e_rot, e_rotc, e_pair);
    return 0;
}
