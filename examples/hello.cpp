// hello.cpp — what the reference's main driver does (hello.cu:3-48: the default option, CPU pricers, the three
// GPU pricers, the three nested-MC strategies, the closed form), expressed on the shim.  Same defaults, same
// wrapper functions, same printed labels; no cudaMemcpyToSymbol step (hello.cu:22) because parameters travel
// with each call.  Build: make -C examples ; run on an MI355X.
#include "monte_carlo.hpp"

#include <array>

namespace {

// hello.cu:5-17 — S0, T, K, r, v, B, P1, P2, N_PATHS, N_PATHS_INNER, N_STEPS, step
OptionData default_option()
{
    OptionData od{100.0f, 1.0f, 100.0f, 0.1f, 0.2f, 120.0f, 10, 50, 100000, 1000, 100, 0.0f};
    od.step = od.T / static_cast<float>(od.N_STEPS);
    return od;
}

using SingleLevel = float (*)(OptionData, int);
using Nested = float (*)(OptionData, int, int);

}  // namespace

int main()
{
    const OptionData od = default_option();
    constexpr int kThreadsPerBlock = 1024;   // accepted and ignored by the engine (hello.cu:19)
    constexpr int kBlocks = 5000;            // likewise (hello.cu:38-40)

    printOptionData(od);
    int n_devices = 0;
    testCUDA(mcamd_device_count(&n_devices));   // the reference's error macro (inc/tool.cuh:92-100) on an engine status code
    getDeviceProperty();

    const std::array<SingleLevel, 5> single = {wrapper_cpu_option_vanilla, wrapper_cpu_bullet_option,
                                               wrapper_gpu_option_vanilla, wrapper_gpu_bullet_option,
                                               wrapper_gpu_bullet_option_atomic};
    for (SingleLevel price : single) price(od, kThreadsPerBlock);

    const std::array<Nested, 3> nested = {wrapper_gpu_bullet_option_nmc_one_point_one_block,
                                          wrapper_gpu_bullet_option_nmc_one_kernel,
                                          wrapper_gpu_bullet_option_nmc_optimal};
    for (Nested price : nested) price(od, kThreadsPerBlock, kBlocks);

    float closed_form = 0.0f;
    black_scholes_CPU(closed_form, od.S0, od.K, od.T, od.r, od.v);
    std::cout << "\ncall Black Scholes : " << closed_form << std::endl;

    // beyond the reference: the CPU pricers with a seed (the reference's are seeded from std::random_device), so that a
    // test can hold the GPU prices against them at a fixed tolerance
    const uint64_t cpu_seed = 1234;
    float cpu_vanilla = 0.0f, cpu_bullet = 0.0f;
    simulateOptionPriceCPU(&cpu_vanilla, od, &cpu_seed);
    simulateBulletOptionPriceCPU(&cpu_bullet, od, &cpu_seed);
    std::cout << "seeded CPU vanilla : " << cpu_vanilla << std::endl;
    std::cout << "seeded CPU bullet : " << cpu_bullet << std::endl;

    // beyond the reference: fp64 paths with a standard error and a confidence interval
    const mcamd_result r = wrapper_gpu_option_vanilla_f64(od, 1);
    std::cout << "fp64 vanilla : " << r.price << " +- " << r.std_err << "  95% CI [" << r.ci_lo << ", " << r.ci_hi
              << "]" << std::endl;
    return 0;
}
