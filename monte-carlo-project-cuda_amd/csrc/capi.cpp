// capi.cpp — the C ABI declared in include/mcamd.h: argument checking, context/scratch
// ownership, HIP-event timing, final host-side statistics.  All device work is enqueued through
// the launchers of launch.hpp on the context's stream; there is no CPU fallback of any kind — a
// call either runs the gfx950 kernels or returns an error.
#include "launch.hpp"

#include "mcamd.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>

static_assert(sizeof(mcamd_option) == 88 && sizeof(mcamd_sim) == 48 && sizeof(mcamd_result) == 128 &&
                  sizeof(mcamd_device_info) == 384,
              "C ABI struct layout changed: bump MCAMD_ABI_VERSION");

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            (void)hipGetLastError();                                                                   \
            return fail(e_ == hipErrorOutOfMemory ? MCAMD_ERR_NOMEM : MCAMD_ERR_HIP, "%s: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                             \
        }                                                                                              \
    } while (0)

}  // namespace

// used by group.cpp: records the calling thread's last error message
int mcamd_set_error_(int code, const char *msg)
{
    g_last_error = msg ? msg : "";
    return code;
}

struct mcamd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    double *d_partials = nullptr;  // one record (2 or 5 doubles) per block
    uint64_t partial_capacity = 0; // in doubles
    double *d_out = nullptr;       // 8 doubles
    unsigned long long *d_queue = nullptr;  // task counter of the wave-per-point nested-MC kernel
    unsigned int *d_ticket = nullptr;       // arrival counter of kernels that finish their own sum (in d_queue's allocation)
    double *h_out_dev = nullptr;            // h_out as the device addresses it (a self-finishing kernel writes there)
    uint32_t compute_units = 0;
    double *h_out = nullptr;       // pinned, 8 doubles
    // asynchronous calls: a ring of event pairs around the simulation kernel of the last kRing enqueues
    static constexpr uint32_t kRing = 64;
    hipEvent_t ring0[kRing] = {}, ring1[kRing] = {};
    uint64_t n_enqueued = 0;
};

namespace {

int ensure_partials(mcamd_ctx *ctx, uint32_t records, int record_doubles = 2)
{
    const uint64_t need = static_cast<uint64_t>(records) * record_doubles;
    if (need <= ctx->partial_capacity) return MCAMD_OK;
    if (ctx->d_partials) HIP_TRY(hipFree(ctx->d_partials));
    ctx->d_partials = nullptr;
    ctx->partial_capacity = 0;
    HIP_TRY(hipMalloc(&ctx->d_partials, need * sizeof(double)));
    ctx->partial_capacity = need;
    return MCAMD_OK;
}

int check_common(const mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, const mcamd_result *res)
{
    if (!ctx) return fail(MCAMD_ERR_INVALID, "ctx is NULL");
    if (!opt || !sim || !res) return fail(MCAMD_ERR_INVALID, "opt, sim and res must be non-NULL");
    if (sim->precision != MCAMD_F32 && sim->precision != MCAMD_F64)
        return fail(MCAMD_ERR_INVALID, "precision must be MCAMD_F32 (32) or MCAMD_F64 (64), got %d", sim->precision);
    if (sim->n_steps == 0) return fail(MCAMD_ERR_INVALID, "n_steps must be >= 1");
    if (opt->Tk < 0 || static_cast<uint32_t>(opt->Tk) >= sim->n_steps)
        return fail(MCAMD_ERR_INVALID, "restart Tk=%d must satisfy 0 <= Tk < n_steps=%u", opt->Tk, sim->n_steps);
    if (!(opt->T > 0.0) || !(opt->v >= 0.0) || !std::isfinite(opt->S0) || !std::isfinite(opt->K) ||
        !std::isfinite(opt->r) || !std::isfinite(opt->T) || !std::isfinite(opt->v))
        return fail(MCAMD_ERR_INVALID, "option parameters must be finite with T > 0 and v >= 0");
    if (!(opt->dt >= 0.0) || !std::isfinite(opt->dt))
        return fail(MCAMD_ERR_INVALID, "dt must be 0 (= T / n_steps) or a positive finite step, got %g", opt->dt);
    if (sim->flags & ~(MCAMD_FLAG_LOG_SPACE | MCAMD_FLAG_ANTITHETIC | MCAMD_FLAG_CONTROL_VARIATE | MCAMD_FLAG_SEPARATE_REDUCE |
                       MCAMD_FLAG_PRODUCT_FORM))
        return fail(MCAMD_ERR_INVALID, "unknown bits in flags: %d", sim->flags);
    if ((sim->flags & MCAMD_FLAG_LOG_SPACE) && (sim->flags & MCAMD_FLAG_PRODUCT_FORM))
        return fail(MCAMD_ERR_INVALID, "MCAMD_FLAG_LOG_SPACE and MCAMD_FLAG_PRODUCT_FORM exclude each other");
    if (sim->path_offset + sim->n_paths_local < sim->path_offset)
        return fail(MCAMD_ERR_INVALID, "path_offset + n_paths_local overflows 64 bits");
    if (sim->precision == MCAMD_F64) {
        // fp64 keeps a path's exponent as an int32 count of 2^-16 octaves (fast64.hpp ExpAcc): bound the largest
        // log-return a path can reach (|z| <= 8.6 for the 53-bit Box-Muller uniform).  A job beyond this has prices
        // outside the range of a double anyway (e^709); the fp32 path saturates in hardware.
        const double dt = opt->dt > 0.0 ? opt->dt : opt->T / static_cast<double>(sim->n_steps);
        const double per_step = std::fabs((opt->r - 0.5 * opt->v * opt->v) * dt) + 8.6 * opt->v * std::sqrt(dt);
        if (!(per_step < 700.0) || !(per_step * static_cast<double>(sim->n_steps) < 20000.0))
            return fail(MCAMD_ERR_INVALID,
                        "fp64 path: |drift| + 8.6 vol = %.3g per step over %u steps exceeds the exponent range "
                        "(per step < 700, per path < 20000)", per_step, sim->n_steps);
    }
    return MCAMD_OK;
}

mcamd::PathJob make_job(const mcamd_option *opt, const mcamd_sim *sim)
{
    mcamd::PathJob j;
    const double dt = opt->dt > 0.0 ? opt->dt : opt->T / static_cast<double>(sim->n_steps);
    j.drift = (opt->r - 0.5 * opt->v * opt->v) * dt;
    j.vol = opt->v * std::sqrt(dt);
    j.K = opt->K;
    j.B = opt->B;
    j.S_start = (opt->Sk == 0.0) ? opt->S0 : opt->Sk;
    j.P1 = opt->P1;
    j.P2 = opt->P2;
    j.Ik = opt->Ik;
    j.n_sim = sim->n_steps - static_cast<uint32_t>(opt->Tk);
    j.n_steps = sim->n_steps;
    j.seed = sim->seed;
    j.path_offset = sim->path_offset;
    j.n_local = sim->n_paths_local;
    j.window = opt->use_window != 0;
    // the in-register kernels (pricing with or without a window, nested-MC inner stage) carry ln(St / S0) unless the
    // caller asks for the product form; the kernels that must produce St at every step ignore this
    j.logspace = (sim->flags & MCAMD_FLAG_PRODUCT_FORM) == 0;
    j.vr = ((sim->flags & MCAMD_FLAG_ANTITHETIC) ? 1 : 0) | ((sim->flags & MCAMD_FLAG_CONTROL_VARIATE) ? 2 : 0);
    // E[S_T] under the simulated dynamics: S_start exp(r * remaining time)
    j.control_mean = j.S_start * std::exp(opt->r * dt * static_cast<double>(j.n_sim));
    j.precision = sim->precision;
    return j;
}

void zero_result(mcamd_result *res)
{
    std::memset(res, 0, sizeof *res);
}

// How the block records of the kernel just enqueued reach the host.
enum class Finish {
    kFolded,    // the kernel summed them itself and wrote the final record into ctx->h_out (pinned host memory)
    kSmall,     // separate launch of the same sum (launch_small_final), then a copy
    kReduce     // separate 1024-thread reduction (launch_final_reduce), then a copy
};

// final reduce of the block records -> host, with event timing; fills the raw sums + timings
int finish(mcamd_ctx *ctx, uint32_t records, mcamd_result *res, int record_doubles = 2, Finish how = Finish::kReduce)
{
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    if (how != Finish::kFolded) {
        if (how == Finish::kSmall)
            HIP_TRY(mcamd::launch_small_final(ctx->d_partials, records, record_doubles, ctx->d_out, ctx->stream));
        else
            HIP_TRY(mcamd::launch_final_reduce(ctx->d_partials, records, record_doubles, ctx->d_out, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->h_out, ctx->d_out, record_doubles * sizeof(double), hipMemcpyDeviceToHost,
                               ctx->stream));
        HIP_TRY(hipEventRecord(ctx->ev2, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipEventElapsedTime(&res->kernel_ms, ctx->ev0, ctx->ev1));
    if (how == Finish::kFolded) res->total_ms = res->kernel_ms;   // one launch is the whole call
    else HIP_TRY(hipEventElapsedTime(&res->total_ms, ctx->ev0, ctx->ev2));
    res->sum = ctx->h_out[0];
    res->sumsq = ctx->h_out[1];
    if (record_doubles == 5) {
        res->sum_c = ctx->h_out[2];
        res->sum_cc = ctx->h_out[3];
        res->sum_yc = ctx->h_out[4];
    }
    if (record_doubles == mcamd::kNmcRecord) {
        res->work_steps = 64.0 * ctx->h_out[2];  // wave-steps x 64 lanes
        res->live_steps = ctx->h_out[3];
    }
    return MCAMD_OK;
}

void finalize_cv_into(const double s[5], uint64_t n, double r, double T, mcamd_result *res);
void finalize_into(double sum, double sumsq, uint64_t n, double r, double T, mcamd_result *res);

// Common tail of the pricing calls: final reduce + copy + sync, then price / SE / CI from the shard's sums,
// keeping the event timings and the launch shape in the result.
int finish_pricing(mcamd_ctx *ctx, uint32_t grid, int record_doubles, const mcamd_option *opt, const mcamd_sim *sim,
                   mcamd_result *res, Finish how = Finish::kReduce)
{
    if (int rc = finish(ctx, grid, res, record_doubles, how)) return rc;
    const float kms = res->kernel_ms, tms = res->total_ms;
    if (record_doubles == 5) {
        const double sums[5] = {res->sum, res->sumsq, res->sum_c, res->sum_cc, res->sum_yc};
        finalize_cv_into(sums, sim->n_paths_local, opt->r, opt->T, res);
    } else {
        finalize_into(res->sum, res->sumsq, sim->n_paths_local, opt->r, opt->T, res);
    }
    res->kernel_ms = kms;
    res->total_ms = tms;
    res->grid = grid;
    res->block = 256;
    return MCAMD_OK;
}

void finalize_cv_into(const double s[5], uint64_t n, double r, double T, mcamd_result *res)
{
    const double disc = std::exp(-r * T);
    const double N = static_cast<double>(n);
    res->sum = s[0]; res->sumsq = s[1]; res->sum_c = s[2]; res->sum_cc = s[3]; res->sum_yc = s[4];
    res->n = n;
    if (n < 2) {
        res->price = n ? disc * s[0] : 0.0;
        res->std_err = 0.0;
        res->ci_lo = res->ci_hi = res->price;
        return;
    }
    const double ybar = s[0] / N, cbar = s[2] / N;
    const double var_y = std::fmax((s[1] - N * ybar * ybar) / (N - 1.0), 0.0);
    const double var_c = std::fmax((s[3] - N * cbar * cbar) / (N - 1.0), 0.0);
    const double cov = (s[4] - N * ybar * cbar) / (N - 1.0);
    const double beta = var_c > 0.0 ? cov / var_c : 0.0;
    const double rho = (var_c > 0.0 && var_y > 0.0) ? cov / std::sqrt(var_c * var_y) : 0.0;
    const double var_res = std::fmax(var_y - beta * cov, 0.0);
    res->cv_beta = beta;
    res->cv_rho = rho;
    res->price = disc * (ybar - beta * cbar);
    res->std_err = disc * std::sqrt(var_res / N);
    res->ci_lo = res->price - 1.959963984540054 * res->std_err;
    res->ci_hi = res->price + 1.959963984540054 * res->std_err;
}

void finalize_into(double sum, double sumsq, uint64_t n, double r, double T, mcamd_result *res)
{
    const double disc = std::exp(-r * T);
    const double N = static_cast<double>(n);
    const double mean = n ? sum / N : 0.0;
    double var = n > 1 ? (sumsq - N * mean * mean) / (N - 1.0) : 0.0;
    if (var < 0.0) var = 0.0;
    const double se = n ? disc * std::sqrt(var / N) : 0.0;
    res->sum = sum;
    res->sumsq = sumsq;
    res->n = n;
    res->price = disc * mean;
    res->std_err = se;
    res->ci_lo = res->price - 1.959963984540054 * se;
    res->ci_hi = res->price + 1.959963984540054 * se;
}

// argument checks and job of the trajectory store (shared by the synchronous call and the enqueue form)
int prepare_store(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, int layout, const void *d_traj,
                         const int32_t *d_counts, mcamd::PathJob *job)
{
    mcamd_result dummy;
    if (int rc = check_common(ctx, opt, sim, &dummy)) return rc;
    if (sim->flags & (MCAMD_FLAG_ANTITHETIC | MCAMD_FLAG_CONTROL_VARIATE))
        return fail(MCAMD_ERR_INVALID, "variance-reduction flags apply to mcamd_price_paths only");
    if (layout != MCAMD_STEP_MAJOR && layout != MCAMD_PATH_MAJOR)
        return fail(MCAMD_ERR_INVALID, "layout must be MCAMD_STEP_MAJOR or MCAMD_PATH_MAJOR");
    if (sim->n_paths_local == 0) return MCAMD_OK;
    if (!d_traj) return fail(MCAMD_ERR_INVALID, "d_traj is NULL");
    *job = make_job(opt, sim);
    if (d_counts && !job->window) {
        // counts requested for a European payoff: count against B but let every count pay
        job->window = true;
        job->P1 = INT32_MIN;
        job->P2 = INT32_MAX;
    }
    return MCAMD_OK;
}

// argument checks and job of the nested-MC calls (inner stage and fused; shared by the synchronous calls and the
// enqueue forms).  fused: d_prices / d_counts are outputs and outer_seed must differ from the inner seed.
int prepare_nmc(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, int layout, int variant, bool fused,
                       uint64_t outer_seed, const void *d_prices, const int32_t *d_counts, const void *d_point_prices,
                       mcamd::NmcJob *job)
{
    mcamd_result dummy;
    if (int rc = check_common(ctx, opt, sim, &dummy)) return rc;
    if (sim->flags & (MCAMD_FLAG_ANTITHETIC | MCAMD_FLAG_CONTROL_VARIATE))
        return fail(MCAMD_ERR_INVALID, "variance-reduction flags apply to mcamd_price_paths only");
    if (layout != MCAMD_STEP_MAJOR && layout != MCAMD_PATH_MAJOR)
        return fail(MCAMD_ERR_INVALID, "layout must be MCAMD_STEP_MAJOR or MCAMD_PATH_MAJOR");
    if (!fused && variant != MCAMD_NMC_WAVE_PER_POINT && variant != MCAMD_NMC_BLOCK_PER_POINT &&
        variant != MCAMD_NMC_BLOCK_PER_POINT_PLAIN)
        return fail(MCAMD_ERR_INVALID, "unknown nested-MC variant %d", variant);
    if (opt->Tk != 0) return fail(MCAMD_ERR_INVALID, "nested MC expects outer trajectories stored from step 0 (Tk = 0)");
    if (fused && outer_seed == sim->seed)
        return fail(MCAMD_ERR_INVALID, "outer_seed must differ from the inner seed (sim->seed): equal seeds would make "
                                       "outer path p and inner path p draw the same Philox stream");
    if (sim->n_paths_inner == 0) return fail(MCAMD_ERR_INVALID, "n_paths_inner must be >= 1");
    // inner path j of point q draws Philox subsequence q * n_paths_inner + j, q = global_path * n_steps + step: the
    // largest one of the shard must fit 64 bits
    const long double top = static_cast<long double>(sim->path_offset + sim->n_paths_local) * sim->n_steps * sim->n_paths_inner;
    if (top >= 18446744073709551615.0L)
        return fail(MCAMD_ERR_INVALID, "(path_offset + n_paths_local) * n_steps * n_paths_inner overflows the 64-bit "
                                       "Philox subsequence");
    if (sim->n_paths_local == 0) return MCAMD_OK;
    if (!d_prices || !d_point_prices) return fail(MCAMD_ERR_INVALID, "d_prices and d_point_prices must be non-NULL");
    if (opt->use_window && !d_counts) return fail(MCAMD_ERR_INVALID, "bullet window needs d_counts");
    const uint64_t n_points = sim->n_paths_local * static_cast<uint64_t>(sim->n_steps);
    if (n_points / sim->n_steps != sim->n_paths_local) return fail(MCAMD_ERR_INVALID, "point count overflows");
    job->path = make_job(opt, sim);
    job->n_inner = sim->n_paths_inner;
    job->discount = std::exp(-opt->r * opt->T);
    job->n_points = n_points;
    job->compute_units = ctx->compute_units;
    return MCAMD_OK;
}

// host tail of the nested-MC calls: the scalar diagnostic of the wrappers (inc/wrappers.cuh:185-189,316-321)
void fill_nmc_result(mcamd_result *res, uint64_t n_points, uint32_t grid)
{
    res->n = n_points;
    res->price = n_points ? res->sum / static_cast<double>(n_points) : 0.0;  // mean point price (diagnostic)
    res->grid = grid;
    res->block = 256;
}

// Asynchronous calls: an empty shard leaves all-zero statistics, still ordered on the stream.
int enqueue_empty(mcamd_ctx *ctx, double *d_stats)
{
    const uint32_t slot = static_cast<uint32_t>(ctx->n_enqueued % mcamd_ctx::kRing);
    HIP_TRY(hipMemsetAsync(d_stats, 0, 6 * sizeof(double), ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ring0[slot], ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ring1[slot], ctx->stream));
    ctx->n_enqueued++;
    return MCAMD_OK;
}

// Asynchronous calls: `launch` enqueues the simulation kernel (grid blocks, one record of rec doubles each, into
// ctx->d_partials) between a pair of ring events; the final reduce then leaves {record, zeros.., n_value} — the
// 6-double statistics layout — in d_stats.  Nothing waits on the host.
template <typename Launch>
int enqueue_with_stats(mcamd_ctx *ctx, uint32_t grid, int rec, double n_value, double *d_stats, Finish how, Launch launch)
{
    const uint32_t slot = static_cast<uint32_t>(ctx->n_enqueued % mcamd_ctx::kRing);
    // growing the scratch buffer frees the old one: wait for work that may still read it
    if (static_cast<uint64_t>(grid) * rec > ctx->partial_capacity) HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (int rc = ensure_partials(ctx, grid, rec)) return rc;
    HIP_TRY(hipEventRecord(ctx->ring0[slot], ctx->stream));
    HIP_TRY(launch());
    HIP_TRY(hipEventRecord(ctx->ring1[slot], ctx->stream));
    if (how == Finish::kSmall) HIP_TRY(mcamd::launch_small_final(ctx->d_partials, grid, rec, d_stats, ctx->stream, n_value));
    else if (how == Finish::kReduce) HIP_TRY(mcamd::launch_final_reduce(ctx->d_partials, grid, rec, d_stats, ctx->stream, n_value));
    ctx->n_enqueued++;
    return MCAMD_OK;
}

}  // namespace

extern "C" {

int mcamd_abi_version(void)
{
    return MCAMD_ABI_VERSION;
}

const char *mcamd_last_error(void)
{
    return g_last_error.c_str();
}

#ifndef MCAMD_BUILD_ID
#define MCAMD_BUILD_ID "unknown"
#endif
const char *mcamd_build_id(void)
{
    return MCAMD_BUILD_ID;
}

int mcamd_device_count(int *count)
{
    if (!count) return fail(MCAMD_ERR_INVALID, "count is NULL");
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *count = 0;
        return fail(MCAMD_ERR_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    return MCAMD_OK;
}

int mcamd_ctx_create(int device, void *hip_stream, mcamd_ctx **out)
{
    if (!out) return fail(MCAMD_ERR_INVALID, "ctx out-pointer is NULL");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        (void)hipGetLastError();
        return fail(MCAMD_ERR_NODEVICE, "no HIP device visible: this engine has no CPU fallback");
    }
    if (device < 0 || device >= count) return fail(MCAMD_ERR_INVALID, "device %d out of range [0, %d)", device, count);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MCAMD_ERR_NODEVICE, "device %d is %s; the kernels are built for gfx950 only", device,
                    prop.gcnArchName);
    mcamd_ctx *ctx = new (std::nothrow) mcamd_ctx;
    if (!ctx) return fail(MCAMD_ERR_NOMEM, "out of host memory");
    ctx->device = device;
    if (hip_stream) {
        ctx->stream = static_cast<hipStream_t>(hip_stream);
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            return fail(MCAMD_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
        }
        ctx->own_stream = true;
    }
    hipError_t e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev2);
    for (uint32_t i = 0; i < mcamd_ctx::kRing && e == hipSuccess; ++i) {
        e = hipEventCreate(&ctx->ring0[i]);
        if (e == hipSuccess) e = hipEventCreate(&ctx->ring1[i]);
    }
    if (e == hipSuccess) e = hipMalloc(&ctx->d_out, 8 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&ctx->d_queue, 64);
    if (e == hipSuccess) e = hipMemset(ctx->d_queue, 0, 64);   // the ticket must be zero at a kernel's first launch
    if (e == hipSuccess) ctx->d_ticket = reinterpret_cast<unsigned int *>(ctx->d_queue + 2);
    ctx->compute_units = static_cast<uint32_t>(prop.multiProcessorCount);
    if (e == hipSuccess) e = hipHostMalloc(&ctx->h_out, 8 * sizeof(double), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->h_out_dev), ctx->h_out, 0);
    if (e != hipSuccess) {
        mcamd_ctx_destroy(ctx);
        return fail(MCAMD_ERR_HIP, "context setup: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return MCAMD_OK;
}

int mcamd_ctx_destroy(mcamd_ctx *ctx)
{
    if (!ctx) return MCAMD_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_partials) (void)hipFree(ctx->d_partials);
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    if (ctx->d_queue) (void)hipFree(ctx->d_queue);
    if (ctx->h_out) (void)hipHostFree(ctx->h_out);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev2) (void)hipEventDestroy(ctx->ev2);
    for (uint32_t i = 0; i < mcamd_ctx::kRing; ++i) {
        if (ctx->ring0[i]) (void)hipEventDestroy(ctx->ring0[i]);
        if (ctx->ring1[i]) (void)hipEventDestroy(ctx->ring1[i]);
    }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return MCAMD_OK;
}

int mcamd_get_device_info(mcamd_ctx *ctx, mcamd_device_info *info)
{
    if (!ctx || !info) return fail(MCAMD_ERR_INVALID, "ctx and info must be non-NULL");
    std::memset(info, 0, sizeof *info);
    HIP_TRY(hipSetDevice(ctx->device));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, ctx->device));
    std::snprintf(info->name, sizeof info->name, "%s", p.name);
    std::snprintf(info->arch, sizeof info->arch, "%s", p.gcnArchName);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    info->total_mem = total_b;
    info->free_mem = free_b;
    info->compute_units = p.multiProcessorCount;
    info->wavefront_size = p.warpSize;
    info->max_threads_per_block = p.maxThreadsPerBlock;
    info->clock_khz = p.clockRate;
    info->mem_clock_khz = p.memoryClockRate;
    info->mem_bus_bits = p.memoryBusWidth;
    info->lds_per_block = static_cast<int32_t>(p.sharedMemPerBlock);
    info->regs_per_block = p.regsPerBlock;
    info->l2_bytes = p.l2CacheSize;
    info->device_index = ctx->device;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    info->device_count = count;
    return MCAMD_OK;
}

int mcamd_device_malloc(mcamd_ctx *ctx, uint64_t bytes, void **d_ptr)
{
    if (!ctx || !d_ptr) return fail(MCAMD_ERR_INVALID, "ctx and d_ptr must be non-NULL");
    *d_ptr = nullptr;
    if (bytes == 0) return MCAMD_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMalloc(d_ptr, bytes));
    return MCAMD_OK;
}

int mcamd_device_free(mcamd_ctx *ctx, void *d_ptr)
{
    if (!ctx) return fail(MCAMD_ERR_INVALID, "ctx is NULL");
    if (!d_ptr) return MCAMD_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(d_ptr));
    return MCAMD_OK;
}

int mcamd_memcpy_to_host(mcamd_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes)
{
    if (!ctx) return fail(MCAMD_ERR_INVALID, "ctx is NULL");
    if (bytes == 0) return MCAMD_OK;
    if (!h_dst || !d_src) return fail(MCAMD_ERR_INVALID, "copy pointers must be non-NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MCAMD_OK;
}

int mcamd_memcpy_to_device(mcamd_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes)
{
    if (!ctx) return fail(MCAMD_ERR_INVALID, "ctx is NULL");
    if (bytes == 0) return MCAMD_OK;
    if (!d_dst || !h_src) return fail(MCAMD_ERR_INVALID, "copy pointers must be non-NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MCAMD_OK;
}

int mcamd_price_paths(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, mcamd_result *res)
{
    if (int rc = check_common(ctx, opt, sim, res)) return rc;
    zero_result(res);
    if (sim->n_paths_local == 0) return MCAMD_OK;  // empty shard: all-zero statistics
    HIP_TRY(hipSetDevice(ctx->device));
    const mcamd::PathJob job = make_job(opt, sim);
    const int rec = (job.vr & 2) ? 5 : 2;
    const uint32_t grid = mcamd::price_grid(job, ctx->compute_units);
    if (int rc = ensure_partials(ctx, grid, rec)) return rc;
    // few records: the kernel's last workgroup sums them and writes the result straight into pinned host memory —
    // one launch and no copy per call (the reference's shape, inc/trajectories.cuh:77-111 + one cudaMemcpy)
    const Finish how = grid > mcamd::kFoldMaxRecords ? Finish::kReduce
                       : (sim->flags & MCAMD_FLAG_SEPARATE_REDUCE) ? Finish::kSmall : Finish::kFolded;
    mcamd::FinishSpec fs;
    if (how == Finish::kFolded) {
        fs.out = ctx->h_out_dev;
        fs.ticket = ctx->d_ticket;
    }
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    HIP_TRY(mcamd::launch_price(job, ctx->compute_units, ctx->d_partials, ctx->d_queue, grid, fs, ctx->stream));
    return finish_pricing(ctx, grid, rec, opt, sim, res, how);
}

int mcamd_price_paths_enqueue(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, double *d_stats)
{
    mcamd_result dummy;
    if (int rc = check_common(ctx, opt, sim, &dummy)) return rc;
    if (!d_stats) return fail(MCAMD_ERR_INVALID, "d_stats is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    if (sim->n_paths_local == 0) return enqueue_empty(ctx, d_stats);
    const mcamd::PathJob job = make_job(opt, sim);
    const int rec = (job.vr & 2) ? 5 : 2;
    const uint32_t grid = mcamd::price_grid(job, ctx->compute_units);
    const Finish how = grid > mcamd::kFoldMaxRecords ? Finish::kReduce
                       : (sim->flags & MCAMD_FLAG_SEPARATE_REDUCE) ? Finish::kSmall : Finish::kFolded;
    mcamd::FinishSpec fs;
    if (how == Finish::kFolded) {   // the kernel leaves the statistics record in d_stats itself
        fs.out = d_stats;
        fs.ticket = ctx->d_ticket;
        fs.n_value = static_cast<double>(sim->n_paths_local);
    }
    return enqueue_with_stats(ctx, grid, rec, static_cast<double>(sim->n_paths_local), d_stats, how, [&] {
        return mcamd::launch_price(job, ctx->compute_units, ctx->d_partials, ctx->d_queue, grid, fs, ctx->stream);
    });
}

int mcamd_simulate_trajectories_enqueue(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, int layout,
                                        void *d_traj, int32_t *d_counts, void *d_payoffs, double *d_stats)
{
    mcamd::PathJob job;
    if (int rc = prepare_store(ctx, opt, sim, layout, d_traj, d_counts, &job)) return rc;
    if (!d_stats) return fail(MCAMD_ERR_INVALID, "d_stats is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    if (sim->n_paths_local == 0) return enqueue_empty(ctx, d_stats);
    const uint32_t grid = mcamd::store_grid(job.n_local, job.precision);
    return enqueue_with_stats(ctx, grid, 2, static_cast<double>(sim->n_paths_local), d_stats, Finish::kReduce, [&] {
        return mcamd::launch_store(job, layout, d_traj, d_counts, d_payoffs, ctx->d_partials, grid, ctx->stream);
    });
}

int mcamd_nmc_inner_enqueue(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, int layout, int variant,
                            const void *d_prices, const int32_t *d_counts, void *d_point_prices, double *d_stats)
{
    mcamd::NmcJob job;
    if (int rc = prepare_nmc(ctx, opt, sim, layout, variant, false, 0, d_prices, d_counts, d_point_prices, &job)) return rc;
    if (!d_stats) return fail(MCAMD_ERR_INVALID, "d_stats is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    if (sim->n_paths_local == 0) return enqueue_empty(ctx, d_stats);
    const uint32_t grid = mcamd::nmc_grid(job, variant);
    return enqueue_with_stats(ctx, grid, mcamd::kNmcRecord, static_cast<double>(job.n_points), d_stats, Finish::kReduce, [&] {
        return mcamd::launch_nmc_inner(job, layout, variant, d_prices, d_counts, d_point_prices, ctx->d_partials,
                                       ctx->d_queue, grid, ctx->stream);
    });
}

int mcamd_nmc_fused_enqueue(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, uint64_t outer_seed, int layout,
                            void *d_prices, int32_t *d_counts, void *d_point_prices, double *d_stats)
{
    mcamd::NmcJob job;
    if (int rc = prepare_nmc(ctx, opt, sim, layout, 0, true, outer_seed, d_prices, d_counts, d_point_prices, &job)) return rc;
    if (!d_stats) return fail(MCAMD_ERR_INVALID, "d_stats is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    if (sim->n_paths_local == 0) return enqueue_empty(ctx, d_stats);
    const uint32_t grid = mcamd::nmc_fused_grid(job);
    return enqueue_with_stats(ctx, grid, mcamd::kNmcRecord, static_cast<double>(job.n_points), d_stats, Finish::kReduce, [&] {
        return mcamd::launch_nmc_fused(job, outer_seed, layout, d_prices, d_counts, d_point_prices, ctx->d_partials,
                                       ctx->d_queue, grid, ctx->stream);
    });
}

int mcamd_finalize_nmc_stats(const double stats[6], mcamd_result *res)
{
    if (!stats || !res) return fail(MCAMD_ERR_INVALID, "stats and res must be non-NULL");
    zero_result(res);
    res->sum = stats[0];
    res->sumsq = stats[1];
    res->work_steps = 64.0 * stats[2];   // wave-steps x 64 lanes
    res->live_steps = stats[3];
    res->n = static_cast<uint64_t>(std::llround(stats[5]));
    res->price = res->n ? res->sum / static_cast<double>(res->n) : 0.0;
    return MCAMD_OK;
}

int mcamd_enqueued_kernel_ms(mcamd_ctx *ctx, uint32_t n_last, float *ms)
{
    if (!ctx || !ms) return fail(MCAMD_ERR_INVALID, "ctx and ms must be non-NULL");
    if (n_last > mcamd_ctx::kRing || n_last > ctx->n_enqueued)
        return fail(MCAMD_ERR_INVALID, "only the last min(%u, enqueued) calls are kept", mcamd_ctx::kRing);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (uint32_t i = 0; i < n_last; ++i) {
        const uint32_t slot = static_cast<uint32_t>((ctx->n_enqueued - n_last + i) % mcamd_ctx::kRing);
        HIP_TRY(hipEventElapsedTime(&ms[i], ctx->ring0[slot], ctx->ring1[slot]));
    }
    return MCAMD_OK;
}

int mcamd_finalize_stats(const double stats[6], double r, double T, int control_variate, mcamd_result *res)
{
    if (!stats || !res) return fail(MCAMD_ERR_INVALID, "stats and res must be non-NULL");
    zero_result(res);
    const uint64_t n = static_cast<uint64_t>(std::llround(stats[5]));
    if (control_variate) finalize_cv_into(stats, n, r, T, res);
    else finalize_into(stats[0], stats[1], n, r, T, res);
    return MCAMD_OK;
}

int mcamd_simulate_trajectories(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, int layout,
                                void *d_traj, int32_t *d_counts, void *d_payoffs, mcamd_result *res)
{
    if (!res) return fail(MCAMD_ERR_INVALID, "opt, sim and res must be non-NULL");
    mcamd::PathJob job;
    if (int rc = prepare_store(ctx, opt, sim, layout, d_traj, d_counts, &job)) return rc;
    zero_result(res);
    if (sim->n_paths_local == 0) return MCAMD_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t grid = mcamd::store_grid(job.n_local, job.precision);
    if (int rc = ensure_partials(ctx, grid)) return rc;
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    HIP_TRY(mcamd::launch_store(job, layout, d_traj, d_counts, d_payoffs, ctx->d_partials, grid, ctx->stream));
    return finish_pricing(ctx, grid, 2, opt, sim, res);
}

int mcamd_diag_store_pattern(mcamd_ctx *ctx, uint64_t n_paths_local, uint32_t n_steps, int precision, void *d_traj,
                             void *d_payoffs, float *kernel_ms)
{
    if (!ctx || !kernel_ms) return fail(MCAMD_ERR_INVALID, "ctx and kernel_ms must be non-NULL");
    if (precision != MCAMD_F32 && precision != MCAMD_F64) return fail(MCAMD_ERR_INVALID, "bad precision %d", precision);
    *kernel_ms = 0.0f;
    const uint64_t v = precision == MCAMD_F32 ? 4 : 2;
    if (n_paths_local == 0 || n_steps == 0) return MCAMD_OK;
    if (!d_traj) return fail(MCAMD_ERR_INVALID, "d_traj is NULL");
    if (n_paths_local % v != 0 || reinterpret_cast<uintptr_t>(d_traj) % 16 != 0 || reinterpret_cast<uintptr_t>(d_payoffs) % 16 != 0 ||
        n_paths_local + v > 0xffffffffull / 8)
        return fail(MCAMD_ERR_INVALID, "the store pattern is the vector store path's: n_paths_local a multiple of %llu (< 2^29), "
                                       "16-byte aligned buffers", static_cast<unsigned long long>(v));
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t grid = mcamd::store_grid(n_paths_local, precision);
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    HIP_TRY(mcamd::launch_store_pattern(n_paths_local, n_steps, precision, d_traj, d_payoffs, grid, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipEventElapsedTime(kernel_ms, ctx->ev0, ctx->ev1));
    return MCAMD_OK;
}

int mcamd_price_from_normals(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, const void *d_normals,
                             void *d_payoffs, mcamd_result *res)
{
    if (int rc = check_common(ctx, opt, sim, res)) return rc;
    if (sim->flags & (MCAMD_FLAG_ANTITHETIC | MCAMD_FLAG_CONTROL_VARIATE))
        return fail(MCAMD_ERR_INVALID, "variance-reduction flags apply to mcamd_price_paths only");
    zero_result(res);
    if (sim->n_paths_local == 0) return MCAMD_OK;
    if (!d_normals) return fail(MCAMD_ERR_INVALID, "d_normals is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    const mcamd::PathJob job = make_job(opt, sim);
    const uint32_t grid = mcamd::array_grid(job.n_local);
    if (int rc = ensure_partials(ctx, grid)) return rc;
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    HIP_TRY(mcamd::launch_from_normals(job, d_normals, d_payoffs, ctx->d_partials, grid, ctx->stream));
    return finish_pricing(ctx, grid, 2, opt, sim, res);
}

int mcamd_generate_normals(mcamd_ctx *ctx, uint64_t seed, uint64_t n, int precision, void *d_out, float *kernel_ms)
{
    if (!ctx) return fail(MCAMD_ERR_INVALID, "ctx is NULL");
    if (precision != MCAMD_F32 && precision != MCAMD_F64) return fail(MCAMD_ERR_INVALID, "bad precision %d", precision);
    if (kernel_ms) *kernel_ms = 0.0f;
    if (n == 0) return MCAMD_OK;
    if (!d_out) return fail(MCAMD_ERR_INVALID, "d_out is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    HIP_TRY(mcamd::launch_generate_normals(seed, n, precision, d_out, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, ctx->ev0, ctx->ev1));
    return MCAMD_OK;
}

int mcamd_reduce_sum(mcamd_ctx *ctx, const void *d_in, uint64_t n, int precision, int variant, double *sum,
                     float *kernel_ms)
{
    if (!ctx || !sum) return fail(MCAMD_ERR_INVALID, "ctx and sum must be non-NULL");
    if (precision != MCAMD_F32 && precision != MCAMD_F64) return fail(MCAMD_ERR_INVALID, "bad precision %d", precision);
    if (variant < MCAMD_REDUCE_SEQUENTIAL || variant > MCAMD_REDUCE_GRID_STRIDE)
        return fail(MCAMD_ERR_INVALID, "reduce variant must be 3..6 (ReductionType), got %d", variant);
    *sum = 0.0;
    if (kernel_ms) *kernel_ms = 0.0f;
    if (n == 0) return MCAMD_OK;
    if (!d_in) return fail(MCAMD_ERR_INVALID, "d_in is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t grid = mcamd::reduce_grid(n, variant);
    if (int rc = ensure_partials(ctx, grid)) return rc;
    mcamd_result tmp;
    zero_result(&tmp);
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    HIP_TRY(mcamd::launch_reduce(d_in, n, precision, variant, ctx->d_partials, grid, ctx->stream));
    if (int rc = finish(ctx, grid, &tmp)) return rc;
    *sum = tmp.sum;
    if (kernel_ms) *kernel_ms = tmp.kernel_ms;
    return MCAMD_OK;
}

int mcamd_nmc_inner(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, int layout, int variant,
                    const void *d_prices, const int32_t *d_counts, void *d_point_prices, mcamd_result *res)
{
    if (!res) return fail(MCAMD_ERR_INVALID, "opt, sim and res must be non-NULL");
    mcamd::NmcJob job;
    if (int rc = prepare_nmc(ctx, opt, sim, layout, variant, false, 0, d_prices, d_counts, d_point_prices, &job)) return rc;
    zero_result(res);
    if (sim->n_paths_local == 0) return MCAMD_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t grid = mcamd::nmc_grid(job, variant);
    if (int rc = ensure_partials(ctx, grid, mcamd::kNmcRecord)) return rc;
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    HIP_TRY(mcamd::launch_nmc_inner(job, layout, variant, d_prices, d_counts, d_point_prices, ctx->d_partials,
                                    ctx->d_queue, grid, ctx->stream));
    if (int rc = finish(ctx, grid, res, mcamd::kNmcRecord)) return rc;
    fill_nmc_result(res, job.n_points, grid);
    return MCAMD_OK;
}

int mcamd_nmc_fused(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, uint64_t outer_seed, int layout,
                    void *d_prices, int32_t *d_counts, void *d_point_prices, mcamd_result *res)
{
    if (!res) return fail(MCAMD_ERR_INVALID, "opt, sim and res must be non-NULL");
    mcamd::NmcJob job;
    if (int rc = prepare_nmc(ctx, opt, sim, layout, 0, true, outer_seed, d_prices, d_counts, d_point_prices, &job)) return rc;
    zero_result(res);
    if (sim->n_paths_local == 0) return MCAMD_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t grid = mcamd::nmc_fused_grid(job);
    if (int rc = ensure_partials(ctx, grid, mcamd::kNmcRecord)) return rc;
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    HIP_TRY(mcamd::launch_nmc_fused(job, outer_seed, layout, d_prices, d_counts, d_point_prices, ctx->d_partials,
                                    ctx->d_queue, grid, ctx->stream));
    if (int rc = finish(ctx, grid, res, mcamd::kNmcRecord)) return rc;
    fill_nmc_result(res, job.n_points, grid);
    return MCAMD_OK;
}

int mcamd_reduce_partials(mcamd_ctx *ctx, const void *d_in, uint64_t n, int precision, int variant, uint32_t n_blocks,
                          double *h_partials, float *kernel_ms)
{
    if (!ctx || !h_partials) return fail(MCAMD_ERR_INVALID, "ctx and h_partials must be non-NULL");
    if (precision != MCAMD_F32 && precision != MCAMD_F64) return fail(MCAMD_ERR_INVALID, "bad precision %d", precision);
    if (variant < MCAMD_REDUCE_SEQUENTIAL || variant > MCAMD_REDUCE_GRID_STRIDE)
        return fail(MCAMD_ERR_INVALID, "reduce variant must be 3..6 (ReductionType), got %d", variant);
    if (n_blocks == 0 || n_blocks > mcamd::kMaxGrid) return fail(MCAMD_ERR_INVALID, "n_blocks must be in [1, 2^20]");
    if (kernel_ms) *kernel_ms = 0.0f;
    for (uint32_t b = 0; b < n_blocks; ++b) h_partials[b] = 0.0;
    if (n == 0) return MCAMD_OK;
    if (!d_in) return fail(MCAMD_ERR_INVALID, "d_in is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    if (int rc = ensure_partials(ctx, n_blocks)) return rc;
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    HIP_TRY(mcamd::launch_reduce(d_in, n, precision, variant, ctx->d_partials, n_blocks, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    // the kernels leave records of two doubles {partial, 0}: strided copy of the first of each
    HIP_TRY(hipMemcpy2DAsync(h_partials, sizeof(double), ctx->d_partials, 2 * sizeof(double), sizeof(double), n_blocks,
                             hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, ctx->ev0, ctx->ev1));
    return MCAMD_OK;
}

int mcamd_cpu_mc_f32(const mcamd_option *opt, uint64_t n_paths, uint32_t n_steps, uint64_t seed, int from_random_device,
                     float *price, float *payoff_sum)
{
    if (!opt || !price) return fail(MCAMD_ERR_INVALID, "opt and price must be non-NULL");
    if (n_steps == 0) return fail(MCAMD_ERR_INVALID, "n_steps must be >= 1");
    *price = 0.0f;
    if (payoff_sum) *payoff_sum = 0.0f;
    if (n_paths == 0) return MCAMD_OK;
    // fp32 throughout, the operation order of inc/tool.cuh:104-173
    const float K = static_cast<float>(opt->K), r = static_cast<float>(opt->r), sigma = static_cast<float>(opt->v);
    const float S0 = static_cast<float>(opt->S0), T = static_cast<float>(opt->T), B = static_cast<float>(opt->B);
    const float dt = opt->dt > 0.0 ? static_cast<float>(opt->dt) : T / static_cast<float>(n_steps);
    const float sqrdt = sqrtf(dt);
    const float P1 = static_cast<float>(opt->P1), P2 = static_cast<float>(opt->P2);   // the reference compares as floats
    std::mt19937 generator(from_random_device ? std::random_device{}() : static_cast<std::mt19937::result_type>(seed));
    std::normal_distribution<float> distribution(0.0f, 1.0f);
    float sum = 0.0f;
    for (uint64_t i = 0; i < n_paths; ++i) {
        float St = S0;
        int count = 0;
        for (uint32_t j = 0; j < n_steps; ++j) {
            const float G = distribution(generator);
            St *= expf((r - (sigma * sigma) / 2) * dt + sigma * sqrdt * G);
            if (opt->use_window && St < B) count++;
        }
        if (!opt->use_window || (count >= P1 && count <= P2)) sum += std::fmax(St - K, 0.0f);
    }
    if (payoff_sum) *payoff_sum = sum;
    *price = expf(-r * T) * sum / static_cast<float>(n_paths);
    return MCAMD_OK;
}

int mcamd_finalize(double sum, double sumsq, uint64_t n, double r, double T, mcamd_result *res)
{
    if (!res) return fail(MCAMD_ERR_INVALID, "res is NULL");
    zero_result(res);
    finalize_into(sum, sumsq, n, r, T, res);
    return MCAMD_OK;
}

int mcamd_finalize_cv(const double sums[5], uint64_t n, double r, double T, mcamd_result *res)
{
    if (!res || !sums) return fail(MCAMD_ERR_INVALID, "sums and res must be non-NULL");
    zero_result(res);
    finalize_cv_into(sums, n, r, T, res);
    return MCAMD_OK;
}

// Closed form, host.  Same operation order and the same mixed precision as the reference:
// `0.5 * v * v` and `exp(-r * T)` are evaluated in double and narrowed (inc/BlackandScholes.hpp:37,42).
float mcamd_cnd_f32(float x)
{
    const float p = 0.2316419f;
    const float b1 = 0.31938153f, b2 = -0.356563782f, b3 = 1.781477937f, b4 = -1.821255978f, b5 = 1.330274429f;
    const float one_over_sqrt_twopi = 0.39894228f;
    const float t = 1.0f / (1.0f + p * std::fabs(x));
    const float tail = one_over_sqrt_twopi * expf(-x * x / 2.0f) * t * (t * (t * (t * (t * b5 + b4) + b3) + b2) + b1);
    return x >= 0.0f ? 1.0f - tail : tail;
}

float mcamd_bs_call_f32(float x0, float strike, float T, float r, float sigma)
{
    const float sqrtT = sqrtf(T);
    const float d1 = static_cast<float>(
        (static_cast<double>(logf(x0 / strike)) + (static_cast<double>(r) + 0.5 * sigma * sigma) * T) /
        static_cast<double>(sigma * sqrtT));
    const float d2 = d1 - sigma * sqrtT;
    const float n1 = mcamd_cnd_f32(d1), n2 = mcamd_cnd_f32(d2);
    return static_cast<float>(static_cast<double>(x0 * n1) -
                              static_cast<double>(strike) * std::exp(static_cast<double>(-r * T)) * n2);
}

double mcamd_bs_call_f64(double x0, double strike, double T, double r, double sigma)
{
    const double sqrtT = std::sqrt(T);
    const double d1 = (std::log(x0 / strike) + (r + 0.5 * sigma * sigma) * T) / (sigma * sqrtT);
    const double d2 = d1 - sigma * sqrtT;
    return x0 * 0.5 * std::erfc(-d1 / std::sqrt(2.0)) - strike * std::exp(-r * T) * 0.5 * std::erfc(-d2 / std::sqrt(2.0));
}

}  // extern "C"
