/*
 * oracle.h — CPU restatement of the reference's Monte Carlo pricing path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker / the reported CPU baseline.  The shipped
 * library (libmcamd.so) never links, loads or calls anything in this directory.
 *
 * Every function cites the reference lines (relative to /root/reference) whose
 * arithmetic it restates.  The reference's GPU random stream (closed-source
 * cuRAND XORWOW) cannot be reproduced here, so RNG-driven functions consume the
 * build's own counter-based stream ("MCAMD stream v1", identical to
 * rocrand_init(seed, subsequence = global path id, offset = 0) followed by
 * rocrand_normal4 / rocrand_normal_double2 calls) and are pinned by
 *   - the rocRAND Philox4x32-10 known-answer words,
 *   - the reference's closed form (inc/BlackandScholes.hpp, also compiled as
 *     oracle/_ref/libref_bs.so) and its golden values,
 *   - the reference's array-driven CPU pricer golden vector (inc/testing.cuh:75-91).
 * Parity status: pinned for closed form + array-driven path; "parity unpinned"
 * at the cuRAND boundary (the reference's own tests hold no numbers there).
 */
#ifndef MCAMD_ORACLE_H
#define MCAMD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Option + simulation parameters.  Mirrors the fields of the reference's
 * OptionData (inc/tool.cuh:13-26) in double precision with 64-bit counts. */
typedef struct oracle_params {
    double S0, T, K, r, v, B;
    int32_t P1, P2;
    uint64_t n_paths;        /* paths in the whole job (the N in price = e^{-rT} sum / N) */
    uint32_t n_steps;
    uint32_t n_paths_inner;
    uint64_t seed;
    int32_t use_window;      /* 0: European call (no barrier test); 1: bullet window P1<=count<=P2 */
    /* restart triple (inc/trajectories.cuh:116-117,140-143): count starts at Ik, price at Sk
     * (Sk == 0 means S0), and only n_steps - Tk steps are simulated */
    int32_t Ik;
    double Sk;
    int32_t Tk;
    /* time step of the multi-step loops: the reference kernels read OptionData.step (inc/tool.cuh:25,
     * inc/trajectories.cuh:131, inc/nmc.cuh:28; the CPU pricer takes it as an argument, inc/tool.cuh:133);
     * 0 means T / n_steps, which is what hello.cu:17 stores there */
    double dt;
} oracle_params;

/* ---- RNG: Philox4x32-10 (Random123 constants; rocRAND counter convention) ---- */
/* counter = (block_lo, block_hi, subseq_lo, subseq_hi), key = (seed_lo, seed_hi). */
void oracle_philox4x32_10(uint64_t seed, uint64_t subsequence, uint64_t block, uint32_t out[4]);
/* rocRAND's Box-Muller (host branch) on one Philox block: 4 floats / 2 doubles. */
void oracle_normal4_f32(uint64_t seed, uint64_t subsequence, uint64_t block, float out[4]);
void oracle_normal2_f64(uint64_t seed, uint64_t subsequence, uint64_t block, double out[2]);
/* bulk fill: out[i] = i-th normal of the stream (subsequence = i / per_block... see .c) */
void oracle_generate_normals_f32(uint64_t seed, uint64_t n, float *out);
void oracle_generate_normals_f64(uint64_t seed, uint64_t n, double *out);

/* ---- closed form (inc/BlackandScholes.hpp) ---- */
float oracle_cnd_f32(float x);                                        /* :8-30  */
float oracle_bs_call_f32(float x0, float K, float T, float r, float sigma); /* :34-43 */
double oracle_bs_call_f64(double x0, double K, double T, double r, double sigma); /* exact, erfc */
/* sum of oracle_bs_call_f32 over a sqrt(n) x sqrt(n) grid of strikes 50..150 and volatilities 0.05..0.55
 * (BASELINE configs[0]'s "1M evals over a (K, sigma) grid" timing; same grid as oracle/ref_bs_driver.cpp) */
double oracle_bs_call_f32_grid(int n);

/* ---- array-driven pricer (inc/testing.cuh:75-91; kernels inc/trajectories.cuh:14-52) ---- */
/* normals[path * n_steps + step]; writes per-path undiscounted payoffs; returns their mean. */
float oracle_price_from_normals_f32(const float *normals, uint64_t n_paths, uint32_t n_steps,
                                    float S0, float sigma, float sqrdt, float r, float K, float dt,
                                    float *payoffs);
double oracle_price_from_normals_f64(const double *normals, uint64_t n_paths, uint32_t n_steps,
                                     double S0, double sigma, double sqrdt, double r, double K, double dt,
                                     double *payoffs);

/* ---- RNG-driven MC on the MCAMD stream ---- */
/* Simulates global path ids [path_lo, path_lo + n_local).  precision: 32 or 64.
 * payoffs (optional, may be NULL): n_local undiscounted payoffs as double.
 * trajectories (optional): step-major [n_steps_sim][n_local] as double (value of St after each step).
 * counts (optional): step-major running barrier counts, int32.
 * sum/sumsq: fp64 sums of the undiscounted payoffs.
 * n_steps == 1 reproduces the exact one-step pricer (inc/trajectories.cuh:58-76, inc/tool.cuh:104-130);
 * otherwise the step loop of inc/trajectories.cuh:144-148 / inc/tool.cuh:157-166.
 * threads: OpenMP threads to use (<=1: serial). */
void oracle_mc_paths(const oracle_params *p, int precision, uint64_t path_lo, uint64_t n_local,
                     double *payoffs, double *trajectories, int32_t *counts,
                     double *sum, double *sumsq, int threads);

/* Opt-in variance-reduced estimator (antithetic pairs and/or S_T control variate): five raw sums
 * {sum y, sum y^2, sum c, sum c^2, sum y c}, c = S_T - control_mean. */
void oracle_mc_paths_vr(const oracle_params *p, int precision, uint64_t path_lo, uint64_t n_local, int antithetic,
                        double control_mean, double sums[5], int threads);

/* Nested MC inner price of one stored point (inc/nmc.cuh:47-66,100-103):
 * n_paths_inner continuation paths of n_steps-1-step steps from (St, count), windowed payoff,
 * mean, discounted by e^{-rT}.  Stream: seed = p->seed, subsequence = point_id * n_paths_inner + j. */
double oracle_nmc_point(const oracle_params *p, int precision, uint64_t point_id, uint32_t step,
                        double St, int32_t count);

/* Discount + mean + standard error (inc/wrappers.cuh:51,85; SE/CI are new capability). */
void oracle_finalize(double sum, double sumsq, uint64_t n, double r, double T,
                     double *price, double *std_err, double *ci_lo, double *ci_hi);

/* fp64 sum of a float / double array (truth for the reduce kernels, inc/testing.cuh:161-174). */
double oracle_sum_f32(const float *x, uint64_t n);
double oracle_sum_f64(const double *x, uint64_t n);

int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
