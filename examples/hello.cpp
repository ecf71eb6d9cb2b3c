// hello.cpp — the reference's main driver (hello.cu:3-48) re-created on the shim: same parameters,
// same order of calls, same printed labels.  Build: make -C examples ; run on an MI355X.
// No cudaMemcpyToSymbol step is needed (hello.cu:22): parameters travel with each call.
#include "monte_carlo.hpp"

int main()
{
    OptionData option_data;
    option_data.S0 = 100.0f;
    option_data.T = 1.0f;
    option_data.K = 100.0f;
    option_data.r = 0.1f;
    option_data.v = 0.2f;
    option_data.B = 120.0f;
    option_data.P1 = 10;
    option_data.P2 = 50;
    option_data.N_PATHS = 100000;
    option_data.N_PATHS_INNER = 1000;
    option_data.N_STEPS = 100;
    option_data.step = option_data.T / static_cast<float>(option_data.N_STEPS);

    const int threadsPerBlock = 1024;

    printOptionData(option_data);
    getDeviceProperty();

    wrapper_cpu_option_vanilla(option_data, threadsPerBlock);
    wrapper_cpu_bullet_option(option_data, threadsPerBlock);

    wrapper_gpu_option_vanilla(option_data, threadsPerBlock);
    wrapper_gpu_bullet_option(option_data, threadsPerBlock);
    wrapper_gpu_bullet_option_atomic(option_data, threadsPerBlock);

    wrapper_gpu_bullet_option_nmc_one_point_one_block(option_data, threadsPerBlock, 5000);
    wrapper_gpu_bullet_option_nmc_one_kernel(option_data, threadsPerBlock, 5000);
    wrapper_gpu_bullet_option_nmc_optimal(option_data, threadsPerBlock, 5000);

    float callResult = 0.0f;
    black_scholes_CPU(callResult, option_data.S0, option_data.K, option_data.T, option_data.r, option_data.v);
    std::cout << std::endl << "call Black Scholes : " << callResult << std::endl;

    // new: fp64 paths with a confidence interval
    const mcamd_result r = wrapper_gpu_option_vanilla_f64(option_data, 1);
    std::cout << "fp64 vanilla : " << r.price << " +- " << r.std_err << "  95% CI [" << r.ci_lo << ", " << r.ci_hi
              << "]" << std::endl;
    return 0;
}
