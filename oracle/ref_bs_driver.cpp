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
