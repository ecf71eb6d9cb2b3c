// wrappers.hpp — the reference's host wrappers (inc/wrappers.cuh:10-340) on top of the mcamd C ABI.
// Same names, same arguments, same printed labels, same return convention (price as float, -1 on a
// failed launch).  threadsPerBlock / number_of_blocks are accepted and ignored: the engine picks its
// own wave64 launch shapes.  Seeds are the reference's hard-coded 1234 (outer / single level) and
// 1235 (inner), inc/wrappers.cuh:41,66,102,151,163.  Paths are simulated in fp32 like the reference
// (sums in fp64); the *_f64 variants at the bottom are new.
//
//   wrapper_cpu_option_vanilla                         :10-20
//   wrapper_cpu_bullet_option                          :21-31
//   wrapper_gpu_option_vanilla                         :33-57
//   wrapper_gpu_bullet_option                          :59-93
//   wrapper_gpu_bullet_option_atomic                   :95-125
//   wrapper_gpu_bullet_option_nmc_one_point_one_block  :128-206
//   wrapper_gpu_bullet_option_nmc_one_kernel           :209-266
//   wrapper_gpu_bullet_option_nmc_optimal              :268-340
//
// Nested MC scalar: the reference averages its per-point array inconsistently (one wrapper includes
// the raw outer-payoff slot, another divides by one more than it sums: SURVEY 2.3/2.4-6).  Here all
// three return the plain mean of the N_PATHS * N_STEPS per-point prices; the per-point array itself
// is available through nmc_point_prices().
#pragma once

#include "tool.hpp"

#include <vector>

namespace mcamd_shim {

inline float report(const char *label, float value)
{
    if (verbose()) std::cout << label << value << std::endl << std::endl;
    return value;
}

inline float failed()
{
    std::fprintf(stderr, "mcamd error: %s\n", mcamd_last_error());
    return -1.0f;
}

// In-register pricing of od's option; n_steps == 1 is the exact one-step vanilla pricer.
inline float price(const OptionData &od, uint32_t n_steps, bool window, int precision, mcamd_result *out = nullptr)
{
    mcamd_ctx *ctx = context();
    if (!ctx) return -1.0f;
    const mcamd_option o = to_option(od, window, n_steps > 1);
    const mcamd_sim s = to_sim(static_cast<uint64_t>(od.N_PATHS), n_steps, 1234, precision);
    mcamd_result r;
    if (mcamd_price_paths(ctx, &o, &s, &r) != MCAMD_OK) return failed();
    if (out) *out = r;
    return static_cast<float>(r.price);
}

constexpr int kFusedNmc = -1;  // selects mcamd_nmc_fused in nested()

// Outer trajectories (seed 1234) + inner stage (seed 1235); returns the mean per-point price and, if
// asked, the per-point array in the reference's path-major indexing [path * N_STEPS + step].
inline float nested(const OptionData &od, int variant, std::vector<float> *points = nullptr)
{
    mcamd_ctx *ctx = context();
    if (!ctx) return -1.0f;
    const uint64_t n = static_cast<uint64_t>(od.N_PATHS), steps = static_cast<uint64_t>(od.N_STEPS);
    const uint64_t n_points = n * steps;
    void *d_prices = nullptr, *d_counts = nullptr, *d_points = nullptr;
    auto release = [&]() {
        mcamd_device_free(ctx, d_prices);
        mcamd_device_free(ctx, d_counts);
        mcamd_device_free(ctx, d_points);
    };
    if (mcamd_device_malloc(ctx, n_points * 4, &d_prices) || mcamd_device_malloc(ctx, n_points * 4, &d_counts) ||
        mcamd_device_malloc(ctx, n_points * 4, &d_points)) {
        release();
        return failed();
    }
    const mcamd_option o = to_option(od, true);
    const mcamd_sim outer = to_sim(n, od.N_STEPS, 1234, MCAMD_F32);
    const mcamd_sim inner = to_sim(n, od.N_STEPS, 1235, MCAMD_F32, static_cast<uint32_t>(od.N_PATHS_INNER));
    mcamd_result r;
    int rc;
    if (variant == kFusedNmc) {
        rc = mcamd_nmc_fused(ctx, &o, &inner, outer.seed, MCAMD_STEP_MAJOR, d_prices, static_cast<int32_t *>(d_counts),
                             d_points, &r);
    } else {
        rc = mcamd_simulate_trajectories(ctx, &o, &outer, MCAMD_STEP_MAJOR, d_prices, static_cast<int32_t *>(d_counts),
                                         nullptr, &r);
        if (!rc) rc = mcamd_nmc_inner(ctx, &o, &inner, MCAMD_STEP_MAJOR, variant, d_prices,
                                      static_cast<const int32_t *>(d_counts), d_points, &r);
    }
    if (!rc && points) {
        std::vector<float> step_major(n_points);
        rc = mcamd_memcpy_to_host(ctx, step_major.data(), d_points, n_points * 4);
        points->resize(n_points);
        for (uint64_t s = 0; s < steps; ++s)
            for (uint64_t p = 0; p < n; ++p) (*points)[p * steps + s] = step_major[s * n + p];
    }
    release();
    if (rc) return failed();
    return static_cast<float>(r.price);
}

}  // namespace mcamd_shim

inline float wrapper_cpu_option_vanilla(OptionData option_data, int /*threadsPerBlock*/)
{
    float price = 0.0f;
    simulateOptionPriceCPU(&price, option_data);
    if (mcamd_shim::verbose()) std::cout << std::endl;
    return mcamd_shim::report("Average CPU Vanilla Option: ", price);
}

inline float wrapper_cpu_bullet_option(OptionData option_data, int /*threadsPerBlock*/)
{
    float price = 0.0f;
    simulateBulletOptionPriceCPU(&price, option_data);
    if (mcamd_shim::verbose()) std::cout << std::endl;
    return mcamd_shim::report("Monte Carlo CPU Bullet Option Price : ", price);
}

inline float wrapper_gpu_option_vanilla(OptionData option_data, int /*threadsPerBlock*/)
{
    const float p = mcamd_shim::price(option_data, 1, false, MCAMD_F32);
    return p < 0 ? p : mcamd_shim::report("Average GPU : ", p);
}

inline float wrapper_gpu_bullet_option(OptionData option_data, int /*threadsPerBlock*/)
{
    const float p = mcamd_shim::price(option_data, static_cast<uint32_t>(option_data.N_STEPS), true, MCAMD_F32);
    return p < 0 ? p : mcamd_shim::report("Average GPU bullet option : ", p);
}

// The reference's second bullet wrapper differs only in how block sums are combined (float atomicAdd
// instead of per-block partials).  The engine has one deterministic fp64 reduction; both names give
// the same number.
inline float wrapper_gpu_bullet_option_atomic(OptionData option_data, int /*threadsPerBlock*/)
{
    const float p = mcamd_shim::price(option_data, static_cast<uint32_t>(option_data.N_STEPS), true, MCAMD_F32);
    return p < 0 ? p : mcamd_shim::report("Average GPU bullet option atomic : ", p);
}

inline float wrapper_gpu_bullet_option_nmc_one_point_one_block(OptionData option_data, int /*threadsPerBlock*/,
                                                               int /*number_of_blocks*/)
{
    const float p = mcamd_shim::nested(option_data, MCAMD_NMC_BLOCK_PER_POINT);
    return p < 0 ? p : mcamd_shim::report("Average GPU bullet option nmc one point per block : ", p);
}

// The reference fuses the outer and inner stages into one launch here; so does mcamd_nmc_fused.  With
// counter-based streams the fusion changes no number: the result equals the two-launch strategies'.
inline float wrapper_gpu_bullet_option_nmc_one_kernel(OptionData option_data, int /*threadsPerBlock*/,
                                                      int /*number_of_blocks*/)
{
    const float p = mcamd_shim::nested(option_data, mcamd_shim::kFusedNmc);
    return p < 0 ? p : mcamd_shim::report("Average GPU bullet option nmc one kernel : ", p);
}

inline float wrapper_gpu_bullet_option_nmc_optimal(OptionData option_data, int /*threadsPerBlock*/,
                                                   int /*number_of_blocks*/)
{
    const float p = mcamd_shim::nested(option_data, MCAMD_NMC_WAVE_PER_POINT);
    return p < 0 ? p : mcamd_shim::report("Average GPU bullet option nmc optimal : ", p);
}

// ---- new capability behind the same surface ----

// Per-point nested-MC prices, indexed [path * N_STEPS + step] like the reference's d_option_prices.
inline std::vector<float> nmc_point_prices(OptionData option_data, int variant = MCAMD_NMC_WAVE_PER_POINT)
{
    std::vector<float> pts;
    if (mcamd_shim::nested(option_data, variant, &pts) < 0) pts.clear();
    return pts;
}

// fp64 paths with standard error and 95% confidence interval.
inline mcamd_result wrapper_gpu_option_vanilla_f64(OptionData option_data, int n_steps = 1)
{
    mcamd_result r{};
    mcamd_shim::price(option_data, static_cast<uint32_t>(n_steps), false, MCAMD_F64, &r);
    return r;
}

inline mcamd_result wrapper_gpu_bullet_option_f64(OptionData option_data)
{
    mcamd_result r{};
    mcamd_shim::price(option_data, static_cast<uint32_t>(option_data.N_STEPS), true, MCAMD_F64, &r);
    return r;
}
