// tool.hpp — host-side mirror of the reference's inc/tool.cuh on top of the mcamd C ABI.
// Same names and argument meaning; the GPU work behind them runs in libmcamd.so (gfx950).
//   OptionData                     inc/tool.cuh:13-26   (same fields, same order, 48 bytes)
//   printOptionData                inc/tool.cuh:29-44
//   getDeviceProperty              inc/tool.cuh:56-88   (hipDeviceProp via mcamd_get_device_info)
//   simulateOptionPriceCPU         inc/tool.cuh:104-130 (serial CPU MC, one exact step)
//   simulateBulletOptionPriceCPU   inc/tool.cuh:133-173 (serial CPU MC, N_STEPS steps + window)
//   CHECK_MALLOC                   inc/tool.cuh:47-53
//   testCUDA                       inc/tool.cuh:92-100  (takes an mcamd status code)
//   get_max_blocks                 inc/tool.cuh:176-188
//   isPow2 / nextPow2              inc/tool.cuh:200-210
// setup_kernel (inc/tool.cuh:192-195) has no counterpart: the engine's Philox counters live in
// registers, there is no RNG state to initialise.  Errors never exit() the process: a failed
// engine call makes the wrapper return -1.0f (the reference's launch-error value,
// inc/wrappers.cuh:77) and leaves the message in mcamd_last_error().
#pragma once

#include "mcamd.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <random>

struct OptionData {
    float S0;
    float T;
    float K;
    float r;
    float v;
    float B;
    int P1;
    int P2;
    int N_PATHS;
    int N_PATHS_INNER;
    int N_STEPS;
    float step;
};
static_assert(sizeof(OptionData) == 48, "OptionData must keep the reference's 48-byte layout");

// Host allocation check with the reference's name and behaviour (inc/tool.cuh:47-53): message + exit.  It guards
// the CALLER's own malloc results; the engine itself never exits (errors come back as status codes).
#ifndef CHECK_MALLOC
#define CHECK_MALLOC(ptr)                                                                                  \
    do {                                                                                                   \
        if ((ptr) == NULL) {                                                                               \
            std::fprintf(stderr, "Memory allocation failed for %s at %s:%d\n", #ptr, __FILE__, __LINE__); \
            std::exit(EXIT_FAILURE);                                                                       \
        }                                                                                                  \
    } while (0)
#endif

// Caller-side status check with the reference's name and behaviour (inc/tool.cuh:92-100): message to stderr, then
// exit.  It takes an mcamd status code (MCAMD_OK plays cudaSuccess).  Like CHECK_MALLOC it is the CALLING program's
// policy: the engine itself never exits, it returns the code and keeps the text in mcamd_last_error().
inline void testCUDA(int status, const char *file, int line)
{
    if (status != MCAMD_OK) {
        std::cerr << "mcamd Error: " << mcamd_last_error() << " in file " << file << " at line " << line << std::endl;
        std::exit(EXIT_FAILURE);
    }
}
#define testCUDA(status) (testCUDA((status), __FILE__, __LINE__))

namespace mcamd_shim {

// Print switch: the reference's wrappers always print their result (inc/wrappers.cuh:16,28,52,...);
// the shim does too unless this is cleared.
inline bool &verbose()
{
    static bool v = true;
    return v;
}

// One lazily created context on device 0, shared by every shim call of the process.
inline mcamd_ctx *context()
{
    static mcamd_ctx *ctx = nullptr;
    if (!ctx && mcamd_ctx_create(0, nullptr, &ctx) != MCAMD_OK) {
        std::fprintf(stderr, "mcamd: %s\n", mcamd_last_error());
        ctx = nullptr;
    }
    return ctx;
}

// multi_step: the kernels that loop over N_STEPS read the time step from OptionData.step
// (inc/trajectories.cuh:131, inc/nmc.cuh:28), so it travels as mcamd_option.dt; the one-step vanilla pricer
// uses T itself (inc/trajectories.cuh:58-76) and passes 0.
inline mcamd_option to_option(const OptionData &od, bool window, bool multi_step = true)
{
    mcamd_option o{};
    o.S0 = od.S0; o.T = od.T; o.K = od.K; o.r = od.r; o.v = od.v; o.B = od.B;
    o.P1 = od.P1; o.P2 = od.P2;
    o.use_window = window ? 1 : 0;
    o.dt = (multi_step && od.step > 0.0f) ? static_cast<double>(od.step) : 0.0;
    return o;
}

inline mcamd_sim to_sim(uint64_t n_paths, uint32_t n_steps, uint64_t seed, int precision, uint32_t n_inner = 0)
{
    mcamd_sim s{};
    s.n_paths = n_paths; s.path_offset = 0; s.n_paths_local = n_paths;
    s.n_steps = n_steps; s.n_paths_inner = n_inner; s.seed = seed; s.precision = precision;
    return s;
}

// Serial CPU Monte Carlo (inc/tool.cuh:104-173) through mcamd_cpu_mc_f32: fp32, std::mt19937 +
// std::normal_distribution<float>.  seed == nullptr seeds from std::random_device like the reference (not
// reproducible run to run); a seed makes the run repeatable, so a driver can compare GPU and CPU prices at a
// fixed tolerance.  n_steps == 1 with window == false is the one-step vanilla pricer.
inline float cpu_monte_carlo(const OptionData &od, int n_steps, float dt, bool window, const uint64_t *seed = nullptr)
{
    mcamd_option o = to_option(od, window, n_steps > 1);
    o.dt = dt;
    float price = -1.0f;
    if (mcamd_cpu_mc_f32(&o, static_cast<uint64_t>(od.N_PATHS), static_cast<uint32_t>(n_steps), seed ? *seed : 0,
                         seed == nullptr, &price, nullptr) != MCAMD_OK)
        std::fprintf(stderr, "mcamd error: %s\n", mcamd_last_error());
    return price;
}

}  // namespace mcamd_shim

inline void printOptionData(OptionData od)
{
    std::cout << "\nS0 : " << od.S0 << "\nT : " << od.T << "\nK : " << od.K << "\nr : " << od.r << "\nv : " << od.v
              << "\nB : " << od.B << "\nP1 : " << od.P1 << "\nP2 : " << od.P2 << "\nN_PATHS : " << od.N_PATHS
              << "\nN_PATHS_INNER : " << od.N_PATHS_INNER << "\nN_STEPS : " << od.N_STEPS << "\nstep : " << od.step
              << "\n\n";
}

inline void getDeviceProperty()
{
    mcamd_ctx *ctx = mcamd_shim::context();
    mcamd_device_info di;
    if (!ctx || mcamd_get_device_info(ctx, &di) != MCAMD_OK) {
        std::fprintf(stderr, "mcamd: %s\n", mcamd_last_error());
        return;
    }
    const double GIGA = 1024.0 * 1024.0 * 1024.0;
    std::printf("The number of devices available is %d GPUs \n", di.device_count);
    std::printf("Name: %s (%s)\n", di.name[0] ? di.name : "AMD GPU", di.arch);  // the marketing name needs libdrm's id table
    std::printf("Global memory size in bytes: %fGB (free %fGB)\n", di.total_mem / GIGA, di.free_mem / GIGA);
    std::printf("LDS size per block: %d\n", di.lds_per_block);
    std::printf("Number of registers per block: %d\n", di.regs_per_block);
    std::printf("Number of threads in a wavefront: %d\n", di.wavefront_size);
    std::printf("Maximum number of threads that can be launched per block: %d\n", di.max_threads_per_block);
    std::printf("Clock rate: %d kHz, memory clock %d kHz, memory bus %d bits\n", di.clock_khz, di.mem_clock_khz,
                di.mem_bus_bits);
    std::printf("L2 size: %d\n", di.l2_bytes);
    std::printf("Number of compute units: %d\n", di.compute_units);
}

// The reference's signatures, plus an optional seed (nullptr = std::random_device, as the reference).
inline void simulateOptionPriceCPU(float *optionPriceCPU, OptionData option_data, const uint64_t *seed = nullptr)
{
    *optionPriceCPU = mcamd_shim::cpu_monte_carlo(option_data, 1, option_data.T, false, seed);
}

inline void simulateBulletOptionPriceCPU(float *optionPriceCPU, OptionData option_data, const uint64_t *seed = nullptr)
{
    *optionPriceCPU = mcamd_shim::cpu_monte_carlo(option_data, option_data.N_STEPS, option_data.step, true, seed);
}

// The reference sizes its grid by how many curandState fit in 90% of free memory.  There is no
// state array here; the free/total print is kept and the return value is what the same formula
// gives for a 64-byte state, so callers that only print or cap with it keep working.
inline size_t get_max_blocks(int threads_per_block)
{
    mcamd_ctx *ctx = mcamd_shim::context();
    mcamd_device_info di;
    if (!ctx || mcamd_get_device_info(ctx, &di) != MCAMD_OK) return 0;
    const double GIGA = 1024.0 * 1024.0 * 1024.0;
    std::printf("free_mem: %7.3fGB, total_mem: %7.3fGB\n", di.free_mem / GIGA, di.total_mem / GIGA);
    return static_cast<size_t>(di.free_mem * 0.90) / (64u * static_cast<size_t>(threads_per_block));
}

inline bool isPow2(unsigned int x) { return (x & (x - 1)) == 0; }

inline unsigned int nextPow2(unsigned int x)
{
    if (x <= 1) return 1;
    return 1u << (32 - __builtin_clz(x - 1));
}
