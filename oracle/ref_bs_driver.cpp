// Build recipe glue for oracle/_ref/libref_bs.so — TEST INFRASTRUCTURE ONLY.
// Compiles the reference's closed form from where it lies
// (/root/reference/inc/BlackandScholes.hpp, plain C++, needs only <cmath>) and
// exports it with C linkage so tests can check the oracle restatement against it.
// No reference source is copied: the header is found through -I at build time.
#include "BlackandScholes.hpp"

extern "C" float ref_CND(float x) { return CND(x); }

extern "C" float ref_black_scholes_CPU(float x0, float strike, float T, float r, float sigma)
{
    float call = 0.0f;
    black_scholes_CPU(call, x0, strike, T, r, sigma);
    return call;
}

// Timing aid for BASELINE configs[0] ("1M evals over a (K, sigma) grid"): the reference's black_scholes_CPU called
// n times over a sqrt(n) x sqrt(n) grid of strikes 50..150 and volatilities 0.05..0.55; returns the sum of the
// prices (a checksum, and what keeps the loop from being optimised away).
extern "C" double ref_black_scholes_grid(int n)
{
    int side = 1;
    while ((side + 1) * (side + 1) <= n) ++side;
    double acc = 0.0;
    for (int i = 0; i < side; ++i)
        for (int j = 0; j < side; ++j) {
            float call = 0.0f;
            black_scholes_CPU(call, 100.0f, 50.0f + 100.0f * i / side, 1.0f, 0.1f, 0.05f + 0.5f * j / side);
            acc += call;
        }
    return acc;
}
