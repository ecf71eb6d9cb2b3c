/*
 * oracle.c — CPU restatement of the reference's Monte Carlo pricing path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Plain C, strict IEEE arithmetic
 * (built with -ffp-contract=off), libm transcendentals.
 *
 * Citations are file:line under /root/reference.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* Philox4x32-10.  Published algorithm (Salmon et al., SC'11; Random123);
 * the reference itself uses cuRAND XORWOW (inc/tool.cuh:192-195), which is
 * closed source — this is the build's replacement stream.  The counter/key
 * convention is rocRAND's: key = seed, counter.xy = block index within the
 * subsequence, counter.zw = subsequence.  Known-answer words: SURVEY.md 8c. */
/* ------------------------------------------------------------------ */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void oracle_philox4x32_10(uint64_t seed, uint64_t subsequence, uint64_t block, uint32_t out[4])
{
    uint32_t c0 = (uint32_t)block, c1 = (uint32_t)(block >> 32);
    uint32_t c2 = (uint32_t)subsequence, c3 = (uint32_t)(subsequence >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* rocRAND's Box-Muller, host branch (rocrand_normal.h box_muller / box_muller_double).
 * The multiply-adds are written as explicit fma so that host and device agree on u, v. */
static void box_muller_f32(uint32_t x, uint32_t y, float *a, float *b)
{
    const float two_pow32_inv = 2.3283064e-10f;
    const float two_pow32_inv_2pi = 1.46291807e-09f;
    float u = fmaf((float)x, two_pow32_inv, two_pow32_inv);
    float v = fmaf((float)y, two_pow32_inv_2pi, two_pow32_inv_2pi);
    float s = sqrtf(-2.0f * logf(u));
    *a = sinf(v) * s;
    *b = cosf(v) * s;
}

static void box_muller_f64(const uint32_t w4[4], double *a, double *b)
{
    const double two_pow53_inv = 1.1102230246251565e-16;
    const double pi = 3.1415926535897932;
    uint64_t v1 = (uint64_t)w4[0] ^ ((uint64_t)w4[1] << 21);
    uint64_t v2 = (uint64_t)w4[2] ^ ((uint64_t)w4[3] << 21);
    double u = fma((double)v1, two_pow53_inv, two_pow53_inv);
    double w = fma((double)v2, two_pow53_inv * 2.0, two_pow53_inv * 2.0);
    double s = sqrt(-2.0 * log(u));
    *a = sin(w * pi) * s;
    *b = cos(w * pi) * s;
}

void oracle_normal4_f32(uint64_t seed, uint64_t subsequence, uint64_t block, float out[4])
{
    uint32_t w[4];
    oracle_philox4x32_10(seed, subsequence, block, w);
    box_muller_f32(w[0], w[1], &out[0], &out[1]);
    box_muller_f32(w[2], w[3], &out[2], &out[3]);
}

void oracle_normal2_f64(uint64_t seed, uint64_t subsequence, uint64_t block, double out[2])
{
    uint32_t w[4];
    oracle_philox4x32_10(seed, subsequence, block, w);
    box_muller_f64(w, &out[0], &out[1]);
}

/* Bulk fill = one rocRAND sequence (subsequence 0) consumed front to back:
 * out[4k..4k+3] = normal4(block k)   /   out[2k..2k+1] = normal2(block k).
 * Stands in for the cuRAND host-API call at inc/testing.cuh:17-24. */
void oracle_generate_normals_f32(uint64_t seed, uint64_t n, float *out)
{
    for (uint64_t k = 0; 4 * k < n; ++k) {
        float z[4];
        oracle_normal4_f32(seed, 0, k, z);
        for (int j = 0; j < 4 && 4 * k + j < n; ++j) out[4 * k + j] = z[j];
    }
}

void oracle_generate_normals_f64(uint64_t seed, uint64_t n, double *out)
{
    for (uint64_t k = 0; 2 * k < n; ++k) {
        double z[2];
        oracle_normal2_f64(seed, 0, k, z);
        for (int j = 0; j < 2 && 2 * k + j < n; ++j) out[2 * k + j] = z[j];
    }
}

/* ------------------------------------------------------------------ */
/* Closed form                                                         */
/* ------------------------------------------------------------------ */

/* inc/BlackandScholes.hpp:8-30 — Abramowitz & Stegun 26.2.17, fp32 throughout. */
float oracle_cnd_f32(float x)
{
    const float p = 0.2316419f;
    const float b1 = 0.31938153f, b2 = -0.356563782f, b3 = 1.781477937f;
    const float b4 = -1.821255978f, b5 = 1.330274429f;
    const float inv_sqrt_2pi = 0.39894228f;
    float ax = x >= 0.0f ? x : -x;
    float t = 1.0f / (1.0f + p * ax);
    float poly = t * (t * (t * (t * b5 + b4) + b3) + b2) + b1;
    float tail = inv_sqrt_2pi * expf(-x * x / 2.0f) * t * poly;
    return x >= 0.0f ? 1.0f - tail : tail;
}

/* inc/BlackandScholes.hpp:34-43.  The reference mixes precisions: `0.5 * v * v`
 * and `exp(-r * T)` are evaluated in double (double literal / double exp) and the
 * results narrowed to float; that is kept. */
float oracle_bs_call_f32(float x0, float K, float T, float r, float sigma)
{
    float sqrtT = sqrtf(T);
    float d1 = (float)(((double)logf(x0 / K) + ((double)r + 0.5 * (double)sigma * (double)sigma) * (double)T)
                       / (double)(sigma * sqrtT));
    float d2 = d1 - sigma * sqrtT;
    float n1 = oracle_cnd_f32(d1);
    float n2 = oracle_cnd_f32(d2);
    return (float)((double)(x0 * n1) - (double)K * exp((double)(-r * T)) * (double)n2);
}

double oracle_bs_call_f32_grid(int n)
{
    int side = 1;
    while ((side + 1) * (side + 1) <= n) ++side;
    double acc = 0.0;
    for (int i = 0; i < side; ++i)
        for (int j = 0; j < side; ++j)
            acc += oracle_bs_call_f32(100.0f, 50.0f + 100.0f * i / side, 1.0f, 0.1f, 0.05f + 0.5f * j / side);
    return acc;
}

double oracle_bs_call_f64(double x0, double K, double T, double r, double sigma)
{
    double sqrtT = sqrt(T);
    double d1 = (log(x0 / K) + (r + 0.5 * sigma * sigma) * T) / (sigma * sqrtT);
    double d2 = d1 - sigma * sqrtT;
    double n1 = 0.5 * erfc(-d1 / sqrt(2.0));
    double n2 = 0.5 * erfc(-d2 / sqrt(2.0));
    return x0 * n1 - K * exp(-r * T) * n2;
}

/* ------------------------------------------------------------------ */
/* Array-driven pricer                                                  */
/* ------------------------------------------------------------------ */

/* inc/testing.cuh:75-91: St *= expf((r - sigma^2/2) dt + sigma sqrdt G) for each step,
 * per-path payoff max(St-K,0), returns the UNDISCOUNTED mean accumulated in fp32 (:90). */
float oracle_price_from_normals_f32(const float *normals, uint64_t n_paths, uint32_t n_steps,
                                    float S0, float sigma, float sqrdt, float r, float K, float dt,
                                    float *payoffs)
{
    float acc = 0.0f;
    for (uint64_t i = 0; i < n_paths; ++i) {
        float St = S0;
        for (uint32_t j = 0; j < n_steps; ++j) {
            float G = normals[i * n_steps + j];
            St *= expf((r - (sigma * sigma) / 2) * dt + sigma * sqrdt * G);
        }
        float pay = St - K > 0.0f ? St - K : 0.0f;
        if (payoffs) payoffs[i] = pay;
        acc += pay;
    }
    return n_paths ? acc / (float)n_paths : 0.0f;
}

/* fp64 analogue of the same recurrence (what inc/trajectories.cuh:14-32 computes
 * in the exponent with its double-promoted `exp`, carried through in double). */
double oracle_price_from_normals_f64(const double *normals, uint64_t n_paths, uint32_t n_steps,
                                     double S0, double sigma, double sqrdt, double r, double K, double dt,
                                     double *payoffs)
{
    double acc = 0.0;
    for (uint64_t i = 0; i < n_paths; ++i) {
        double St = S0;
        for (uint32_t j = 0; j < n_steps; ++j) {
            double G = normals[i * n_steps + j];
            St *= exp((r - (sigma * sigma) / 2) * dt + sigma * sqrdt * G);
        }
        double pay = St - K > 0.0 ? St - K : 0.0;
        if (payoffs) payoffs[i] = pay;
        acc += pay;
    }
    return n_paths ? acc / (double)n_paths : 0.0;
}

/* ------------------------------------------------------------------ */
/* RNG-driven MC                                                        */
/* ------------------------------------------------------------------ */

/* One path in fp32.  Step loop: inc/trajectories.cuh:144-148 (GPU) and
 * inc/tool.cuh:157-166 (CPU); payoff window: inc/trajectories.cuh:149-153;
 * one-step exact form (n_steps == 1): inc/trajectories.cuh:74-76. */
static double path_f32_ex(const oracle_params *p, uint64_t seed, uint64_t subseq, uint32_t nsim,
                          float St, int32_t count, double *traj, int32_t *cnts, uint64_t stride, float sign,
                          double *ST_out)
{
    const float r = (float)p->r, sigma = (float)p->v, K = (float)p->K, B = (float)p->B;
    const float dt = p->dt > 0.0 ? (float)p->dt : (float)p->T / (float)p->n_steps;
    const float sqrdt = sqrtf(dt);
    const float drift = (r - (sigma * sigma) / 2) * dt;
    const float vol = sigma * sqrdt;
    float z[4];
    for (uint32_t i = 0; i < nsim; ++i) {
        if ((i & 3u) == 0) oracle_normal4_f32(seed, subseq, i >> 2, z);
        float G = sign * z[i & 3u];
        St *= expf(drift + vol * G);
        if (p->use_window && B > St) count += 1;
        if (traj) traj[(uint64_t)i * stride] = (double)St;
        if (cnts) cnts[(uint64_t)i * stride] = count;
    }
    if (ST_out) *ST_out = (double)St;
    if (p->use_window && !(count >= p->P1 && count <= p->P2)) return 0.0;
    float pay = St - K > 0.0f ? St - K : 0.0f;
    return (double)pay;
}

static double path_f32(const oracle_params *p, uint64_t seed, uint64_t subseq, uint32_t nsim,
                       float St, int32_t count, double *traj, int32_t *cnts, uint64_t stride)
{
    return path_f32_ex(p, seed, subseq, nsim, St, count, traj, cnts, stride, 1.0f, NULL);
}

static double path_f64_ex(const oracle_params *p, uint64_t seed, uint64_t subseq, uint32_t nsim,
                          double St, int32_t count, double *traj, int32_t *cnts, uint64_t stride, double sign,
                          double *ST_out)
{
    const double r = p->r, sigma = p->v, K = p->K, B = p->B;
    const double dt = p->dt > 0.0 ? p->dt : p->T / (double)p->n_steps;
    const double sqrdt = sqrt(dt);
    const double drift = (r - (sigma * sigma) / 2) * dt;
    const double vol = sigma * sqrdt;
    double z[2];
    for (uint32_t i = 0; i < nsim; ++i) {
        if ((i & 1u) == 0) oracle_normal2_f64(seed, subseq, i >> 1, z);
        double G = sign * z[i & 1u];
        St *= exp(drift + vol * G);
        if (p->use_window && B > St) count += 1;
        if (traj) traj[(uint64_t)i * stride] = St;
        if (cnts) cnts[(uint64_t)i * stride] = count;
    }
    if (ST_out) *ST_out = St;
    if (p->use_window && !(count >= p->P1 && count <= p->P2)) return 0.0;
    return St - K > 0.0 ? St - K : 0.0;
}

static double path_f64(const oracle_params *p, uint64_t seed, uint64_t subseq, uint32_t nsim,
                       double St, int32_t count, double *traj, int32_t *cnts, uint64_t stride)
{
    return path_f64_ex(p, seed, subseq, nsim, St, count, traj, cnts, stride, 1.0, NULL);
}

void oracle_mc_paths(const oracle_params *p, int precision, uint64_t path_lo, uint64_t n_local,
                     double *payoffs, double *trajectories, int32_t *counts,
                     double *sum, double *sumsq, int threads)
{
    const uint32_t nsim = p->n_steps - (uint32_t)p->Tk;
    const double S_start = (p->Sk == 0.0) ? p->S0 : p->Sk;
    double s = 0.0, s2 = 0.0;
    (void)threads;
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel for reduction(+ : s, s2) num_threads(threads) schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)n_local; ++i) {
        uint64_t gid = path_lo + (uint64_t)i;
        double *tr = trajectories ? trajectories + i : NULL;
        int32_t *cn = counts ? counts + i : NULL;
        double pay = (precision == 32)
            ? path_f32(p, p->seed, gid, nsim, (float)S_start, p->Ik, tr, cn, n_local)
            : path_f64(p, p->seed, gid, nsim, S_start, p->Ik, tr, cn, n_local);
        if (payoffs) payoffs[i] = pay;
        s += pay;
        s2 += pay * pay;
    }
    *sum = s;
    *sumsq = s2;
}

/* Variance-reduced estimator (new capability, no reference counterpart; checks the engine's opt-in
 * MCAMD_FLAG_ANTITHETIC / MCAMD_FLAG_CONTROL_VARIATE modes).  A sample is one path or, with
 * antithetic != 0, the pair (G, -G) of one path's normals averaged.  sums = {sum y, sum y^2, sum c,
 * sum c^2, sum y c} with c = S_T - control_mean. */
void oracle_mc_paths_vr(const oracle_params *p, int precision, uint64_t path_lo, uint64_t n_local, int antithetic,
                        double control_mean, double sums[5], int threads)
{
    const uint32_t nsim = p->n_steps - (uint32_t)p->Tk;
    const double S_start = (p->Sk == 0.0) ? p->S0 : p->Sk;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    (void)threads;
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel for reduction(+ : s0, s1, s2, s3, s4) num_threads(threads) schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)n_local; ++i) {
        uint64_t gid = path_lo + (uint64_t)i;
        double ST = 0.0, ST2 = 0.0, y, c;
        if (precision == 32) {
            y = path_f32_ex(p, p->seed, gid, nsim, (float)S_start, p->Ik, NULL, NULL, 0, 1.0f, &ST);
            c = ST;
            if (antithetic) {
                double y2 = path_f32_ex(p, p->seed, gid, nsim, (float)S_start, p->Ik, NULL, NULL, 0, -1.0f, &ST2);
                y = (double)(0.5f * ((float)y + (float)y2));
                c = (double)(0.5f * ((float)ST + (float)ST2));
            }
        } else {
            y = path_f64_ex(p, p->seed, gid, nsim, S_start, p->Ik, NULL, NULL, 0, 1.0, &ST);
            c = ST;
            if (antithetic) {
                double y2 = path_f64_ex(p, p->seed, gid, nsim, S_start, p->Ik, NULL, NULL, 0, -1.0, &ST2);
                y = 0.5 * (y + y2);
                c = 0.5 * (ST + ST2);
            }
        }
        c -= control_mean;
        s0 += y; s1 += y * y; s2 += c; s3 += c * c; s4 += y * c;
    }
    sums[0] = s0; sums[1] = s1; sums[2] = s2; sums[3] = s3; sums[4] = s4;
}

/* inc/nmc.cuh:47-66 + :100-103 (one_block_per_point) and :319-343,:378-381 + inc/wrappers.cuh:318
 * (optimal): remaining = N_STEPS - (step + 1) steps from the stored (St, count); a point whose
 * count already exceeds P2 prices to 0 (:53, :330); mean over N_PATHS_INNER; discounted by
 * exp(-r T) with the FULL maturity (:101, :379).  Each inner path restarts from the stored
 * (St, count) — the reference's carry-over between successive inner paths of one thread
 *  is a defect and is not reproduced. */
double oracle_nmc_point(const oracle_params *p, int precision, uint64_t point_id, uint32_t step,
                        double St, int32_t count)
{
    const uint32_t remaining = p->n_steps - (step + 1);
    if (p->use_window && count > p->P2) return 0.0;
    double s = 0.0;
    for (uint32_t j = 0; j < p->n_paths_inner; ++j) {
        uint64_t subseq = point_id * (uint64_t)p->n_paths_inner + j;
        s += (precision == 32)
            ? path_f32(p, p->seed, subseq, remaining, (float)St, count, NULL, NULL, 0)
            : path_f64(p, p->seed, subseq, remaining, St, count, NULL, NULL, 0);
    }
    return s * exp(-p->r * p->T) / (double)p->n_paths_inner;
}

/* inc/wrappers.cuh:51,85: price = exp(-rT) * sum / N.  Sample variance / standard error / 95% CI
 * are new capability (the reference computes none). */
void oracle_finalize(double sum, double sumsq, uint64_t n, double r, double T,
                     double *price, double *std_err, double *ci_lo, double *ci_hi)
{
    const double disc = exp(-r * T);
    const double N = (double)n;
    const double mean = n ? sum / N : 0.0;
    double var = (n > 1) ? (sumsq - N * mean * mean) / (N - 1.0) : 0.0;
    if (var < 0.0) var = 0.0;
    const double se = n ? disc * sqrt(var / N) : 0.0;
    *price = disc * mean;
    *std_err = se;
    *ci_lo = *price - 1.959963984540054 * se;
    *ci_hi = *price + 1.959963984540054 * se;
}

double oracle_sum_f32(const float *x, uint64_t n)
{
    double s = 0.0;
    for (uint64_t i = 0; i < n; ++i) s += (double)x[i];
    return s;
}

double oracle_sum_f64(const double *x, uint64_t n)
{
    double s = 0.0;
    for (uint64_t i = 0; i < n; ++i) s += x[i];
    return s;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
