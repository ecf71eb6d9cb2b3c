// testing.hpp — the reference's test harness (inc/testing.cuh) on top of the mcamd C ABI.
//   generate_random_array / init_random_array   inc/testing.cuh:17-42  (bulk N(0,1) fill + host copy)
//   simulateOptionPriceCPU (array overload)     inc/testing.cuh:75-91  (deterministic CPU pricer)
//   ReductionType                               inc/testing.cuh:100-106
//   class Simulation                            inc/testing.cuh:108-405
// Differences: device buffers are released in the destructor (the reference leaks them); test_reduction
// returns one partial per block like the reference (inc/testing.cuh:227-234; testing.cu:82-88 prints them
// one by one), but the partials always add up to the sum of the whole array, whatever n_blocks is (the
// reference's reduce3..5 blocks read 2 * blockDim elements each and nothing beyond); trajectories come back
// in the reference's path-major order [trajectory * n_steps + step].
#pragma once

#include "tool.hpp"

#include <algorithm>
#include <cstdlib>
#include <vector>

inline void generate_random_array(float *d_randomData, float *h_randomData, int length,
                                  unsigned long long seed = 1234ULL)
{
    mcamd_ctx *ctx = mcamd_shim::context();
    if (!ctx || mcamd_generate_normals(ctx, seed, static_cast<uint64_t>(length), MCAMD_F32, d_randomData, nullptr) ||
        mcamd_memcpy_to_host(ctx, h_randomData, d_randomData, static_cast<uint64_t>(length) * sizeof(float)))
        std::fprintf(stderr, "mcamd error: %s\n", mcamd_last_error());
}

inline void init_random_array(float **d_randomData, float **h_randomData, size_t length, long seed = 1234)
{
    mcamd_ctx *ctx = mcamd_shim::context();
    void *d = nullptr;
    if (!ctx || mcamd_device_malloc(ctx, length * sizeof(float), &d)) {
        std::fprintf(stderr, "mcamd error: %s\n", mcamd_last_error());
        *d_randomData = nullptr;
        *h_randomData = nullptr;
        return;
    }
    *d_randomData = static_cast<float *>(d);
    *h_randomData = static_cast<float *>(std::malloc(length * sizeof(float)));
    generate_random_array(*d_randomData, *h_randomData, static_cast<int>(length),
                          static_cast<unsigned long long>(seed));
}

// Array-driven CPU pricer: per-path undiscounted payoff, returns the undiscounted mean through
// optionPriceCPU (fp32 accumulation, as inc/testing.cuh:90).
inline void simulateOptionPriceCPU(float *optionPriceCPU, int N_PATHS, int N_STEPS, float *h_randomData, float S0,
                                   float sigma, float sqrdt, float r, float K, float dt, float *simulated_paths_cpu)
{
    const float drift = (r - (sigma * sigma) / 2) * dt;
    float total = 0.0f;
    for (int p = 0; p < N_PATHS; ++p) {
        const float *g = h_randomData + static_cast<size_t>(p) * N_STEPS;
        float s = S0;
        for (int k = 0; k < N_STEPS; ++k) s *= expf(drift + sigma * sqrdt * g[k]);
        const float pay = std::max(s - K, 0.0f);
        simulated_paths_cpu[p] = pay;
        total += pay;
    }
    *optionPriceCPU = total / N_PATHS;
}

enum ReductionType {
    SequentialAddressing = 3,
    FirstAddDuringLoad = 4,
    UnrollLastWarp = 5,
    CompletelyUnrolled = 6
};

class Simulation {
public:
    size_t n_trajectories;
    size_t n_steps;
    float *d_random_array = nullptr;
    float *h_random_array = nullptr;

    Simulation(size_t n_trajectories = 10, size_t n_steps = 100, float volatilty = 0.2f, float risk_free_rate = 0.1f,
               float initial_spot_price = 100.0f, float contract_strike = 100.0f, float contract_maturity = 1.0f,
               float barrier = 0.0f, float P1 = 0.0f, float P2 = 0.0f)
        : n_trajectories{n_trajectories}, n_steps{n_steps}, sigma{volatilty}, r{risk_free_rate},
          x_0{initial_spot_price}, K{contract_strike}, T{contract_maturity}, B{barrier}, P1{P1}, P2{P2}
    {
        initialize_random_array();
    }

    Simulation(const Simulation &) = delete;
    Simulation &operator=(const Simulation &) = delete;

    ~Simulation() { release_random_array(); }

    // No RNG state exists in this engine (counter-based Philox): kept for source compatibility.
    void initialize_rng_state(size_t /*threads_per_block*/, uint64_t /*seed*/ = 1234) {}

    float sum_random_array()
    {
        float out = 0.0f;
        for (size_t i = 0; i < length(); ++i) out += h_random_array[i];
        return out;
    }

    // Block sums of the device random array with the chosen schedule on n_blocks workgroups: one partial per
    // block, as the reference returns (inc/testing.cuh:185-235); their sum is the sum of the whole array.
    // n_threads_per_block is accepted and ignored (the engine's workgroups are 256 threads).
    std::vector<float> test_reduction(size_t n_blocks, size_t /*n_threads_per_block*/, int reduction)
    {
        if (n_blocks == 0) n_blocks = 1;
        std::vector<double> partials(n_blocks, 0.0);
        mcamd_ctx *ctx = mcamd_shim::context();
        if (!ctx || mcamd_reduce_partials(ctx, d_random_array, length(), MCAMD_F32, reduction,
                                          static_cast<uint32_t>(n_blocks), partials.data(), nullptr))
            std::fprintf(stderr, "mcamd error: %s\n", mcamd_last_error());
        return std::vector<float>(partials.begin(), partials.end());
    }

    std::vector<float> simulate_trajectory_cpu()
    {
        float option_price = 0.0f;
        std::vector<float> payoffs(n_trajectories, 0.0f);
        simulateOptionPriceCPU(&option_price, static_cast<int>(n_trajectories), static_cast<int>(n_steps),
                               h_random_array, x_0, sigma, sqrt_dt(), r, K, dt(), payoffs.data());
        return payoffs;
    }

    // Same computation on the GPU from the same device random array (array-driven kernel).
    std::vector<float> simulate_trajectory_gpu()
    {
        std::vector<float> payoffs(n_trajectories, 0.0f);
        mcamd_ctx *ctx = mcamd_shim::context();
        void *d_pay = nullptr;
        mcamd_option o{};
        o.S0 = x_0; o.T = T; o.K = K; o.r = r; o.v = sigma;
        const mcamd_sim s = mcamd_shim::to_sim(n_trajectories, static_cast<uint32_t>(n_steps), 0, MCAMD_F32);
        mcamd_result res;
        if (!ctx || mcamd_device_malloc(ctx, n_trajectories * sizeof(float), &d_pay) ||
            mcamd_price_from_normals(ctx, &o, &s, d_random_array, d_pay, &res) ||
            mcamd_memcpy_to_host(ctx, payoffs.data(), d_pay, n_trajectories * sizeof(float)))
            std::fprintf(stderr, "mcamd error: %s\n", mcamd_last_error());
        if (ctx) mcamd_device_free(ctx, d_pay);
        return payoffs;
    }

    // Outer trajectories on the GPU, returned path-major: out[trajectory * n_steps + step].
    std::vector<float> simulate_outer_trajectories(size_t /*n_threads_per_block*/, uint64_t seed)
    {
        std::vector<float> out(length(), 0.0f);
        mcamd_ctx *ctx = mcamd_shim::context();
        void *d_traj = nullptr;
        mcamd_option o{};
        o.S0 = x_0; o.T = T; o.K = K; o.r = r; o.v = sigma;
        const mcamd_sim s = mcamd_shim::to_sim(n_trajectories, static_cast<uint32_t>(n_steps), seed, MCAMD_F32);
        mcamd_result res;
        if (!ctx || mcamd_device_malloc(ctx, length() * sizeof(float), &d_traj) ||
            mcamd_simulate_trajectories(ctx, &o, &s, MCAMD_PATH_MAJOR, d_traj, nullptr, nullptr, &res) ||
            mcamd_memcpy_to_host(ctx, out.data(), d_traj, length() * sizeof(float)))
            std::fprintf(stderr, "mcamd error: %s\n", mcamd_last_error());
        if (ctx) mcamd_device_free(ctx, d_traj);
        return out;
    }

    size_t length() { return n_steps * n_trajectories; }

    void initialize_random_array(size_t seed = 1234ULL)
    {
        release_random_array();
        init_random_array(&d_random_array, &h_random_array, length(), static_cast<long>(seed));
    }

    float &volatility() { return sigma; }
    float &risk_free_rate() { return r; }
    float &initial_spot_price() { return x_0; }
    float &contract_strike() { return K; }
    float &contract_maturity() { return T; }
    float &barrier() { return B; }
    float dt() { return T / n_steps; }
    float sqrt_dt() { return std::sqrt(dt()); }

    float sigma;  // volatility
    float r;      // risk-free rate
    float x_0;    // initial spot price
    float K;      // contract strike
    float T;      // contract maturity
    float B;      // barrier
    float P1;
    float P2;

private:
    void release_random_array()
    {
        if (d_random_array) mcamd_device_free(mcamd_shim::context(), d_random_array);
        std::free(h_random_array);
        d_random_array = nullptr;
        h_random_array = nullptr;
    }
};
