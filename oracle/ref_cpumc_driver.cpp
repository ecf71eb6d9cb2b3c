// Build recipe glue for oracle/_ref/libref_cpumc.so — TEST INFRASTRUCTURE ONLY.
// The reference's CPU Monte Carlo lives in .cuh headers whose include lists need
// the CUDA toolkit (inc/tool.cuh:6-8), so the headers cannot be included whole.
// build_ref.sh streams the host-only line ranges below out of the reference
// files into a throw-away directory under /tmp (never into this repo), this
// driver includes them from there, and only the resulting .so lands in
// oracle/_ref/.  Nothing is stubbed: the ranges use <cmath>/<random>/<algorithm>
// only.
//   ref_optiondata.inc : inc/tool.cuh:13-26     (struct OptionData)
//   ref_cpumc.inc      : inc/tool.cuh:104-173   (simulateOptionPriceCPU, simulateBulletOptionPriceCPU)
//   ref_arraycpu.inc   : inc/testing.cuh:75-91  (array-driven simulateOptionPriceCPU)
#include <algorithm>
#include <cmath>
#include <iostream>
#include <random>
using namespace std;  // as inc/tool.cuh:10

#include "ref_optiondata.inc"
#include "ref_cpumc.inc"
#include "ref_arraycpu.inc"

static OptionData make_od(float S0, float T, float K, float r, float v, float B, int P1, int P2,
                          int N_PATHS, int N_PATHS_INNER, int N_STEPS)
{
    OptionData od;
    od.S0 = S0; od.T = T; od.K = K; od.r = r; od.v = v; od.B = B;
    od.P1 = P1; od.P2 = P2;
    od.N_PATHS = N_PATHS; od.N_PATHS_INNER = N_PATHS_INNER; od.N_STEPS = N_STEPS;
    od.step = od.T / static_cast<float>(od.N_STEPS);  // hello.cu:17
    return od;
}

extern "C" int ref_sizeof_OptionData() { return (int)sizeof(OptionData); }

extern "C" float ref_simulateOptionPriceCPU(float S0, float T, float K, float r, float v, int N_PATHS)
{
    float out = 0.0f;
    simulateOptionPriceCPU(&out, make_od(S0, T, K, r, v, 0.0f, 0, 0, N_PATHS, 0, 1));
    return out;
}

extern "C" float ref_simulateBulletOptionPriceCPU(float S0, float T, float K, float r, float v, float B,
                                                  int P1, int P2, int N_PATHS, int N_STEPS)
{
    float out = 0.0f;
    simulateBulletOptionPriceCPU(&out, make_od(S0, T, K, r, v, B, P1, P2, N_PATHS, 0, N_STEPS));
    return out;
}

extern "C" float ref_simulateOptionPriceCPU_array(int N_PATHS, int N_STEPS, float *normals, float S0,
                                                  float sigma, float sqrdt, float r, float K, float dt,
                                                  float *payoffs)
{
    float out = 0.0f;
    simulateOptionPriceCPU(&out, N_PATHS, N_STEPS, normals, S0, sigma, sqrdt, r, K, dt, payoffs);
    return out;
}

// std::normal_distribution<float> over std::mt19937(seed): the input stream SURVEY.md 8c uses
// for the array-driven golden vector (libstdc++'s algorithm is implementation-defined, so the
// fixture stores the normals themselves).
extern "C" void ref_mt19937_normals(unsigned seed, int n, float *out)
{
    mt19937 gen(seed);
    normal_distribution<float> dist(0.0f, 1.0f);
    for (int i = 0; i < n; ++i) out[i] = dist(gen);
}
