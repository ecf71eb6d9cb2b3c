// multi_gpu.cpp — a C++ host pricing one European call on every visible MI355X of the node from ONE process:
// path shards per device, one RCCL all-reduce of the payoff statistics over xGMI (mcamd_group_*).
// Usage: multi_gpu [n_paths] [n_steps] [n_devices (0 = all)]      Build: make -C examples multi_gpu
#include "mcamd.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv)
{
    const unsigned long long n_paths = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 100000000ULL;
    const unsigned n_steps = argc > 2 ? static_cast<unsigned>(std::atoi(argv[2])) : 252u;
    const int n_devices = argc > 3 ? std::atoi(argv[3]) : 0;

    mcamd_group *group = nullptr;
    if (mcamd_group_create(n_devices, nullptr, &group) != MCAMD_OK) {
        std::fprintf(stderr, "mcamd: %s\n", mcamd_last_error());
        return 1;
    }
    int R = 0;
    mcamd_group_size(group, &R);

    mcamd_option opt{};
    opt.S0 = 100.0; opt.T = 1.0; opt.K = 100.0; opt.r = 0.1; opt.v = 0.2;   // hello.cu:6-10
    mcamd_sim sim{};
    sim.n_paths = n_paths; sim.path_offset = 0; sim.n_paths_local = n_paths;
    sim.n_steps = n_steps; sim.seed = 1234; sim.precision = MCAMD_F64;

    mcamd_result res;
    if (mcamd_group_price_paths(group, &opt, &sim, &res) != MCAMD_OK) {   // warm-up: RCCL rings, code objects
        std::fprintf(stderr, "mcamd (warm-up): %s\n", mcamd_last_error());
        mcamd_group_destroy(group);
        return 1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (mcamd_group_price_paths(group, &opt, &sim, &res) != MCAMD_OK) {
        std::fprintf(stderr, "mcamd: %s\n", mcamd_last_error());
        mcamd_group_destroy(group);
        return 1;
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double bs = mcamd_bs_call_f64(opt.S0, opt.K, opt.T, opt.r, opt.v);
    std::printf("devices %d  paths %llu x %u steps  price %.6f +- %.6f  (closed form %.6f, |err| %.2e)  %.3f s  %.3e paths/s\n",
                R, n_paths, n_steps, res.price, res.std_err, bs, res.price > bs ? res.price - bs : bs - res.price, secs,
                static_cast<double>(n_paths) / secs);
    mcamd_group_destroy(group);
    return 0;
}
