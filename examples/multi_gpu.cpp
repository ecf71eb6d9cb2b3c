// multi_gpu.cpp — a C++ host pricing one European call on every visible MI355X of the node from ONE process:
// path shards per device, one RCCL all-reduce of the payoff statistics over xGMI (mcamd_group_*).  Then the
// reference's nested-MC host (wrapper_gpu_bullet_option_nmc_one_kernel, inc/wrappers.cuh:209-266) the same way:
// every device simulates and prices its share of the outer paths into its own buffers (mcamd_group_nmc_fused),
// per-point prices stay on the owning device, one all-reduce of the statistics record.
// Usage: multi_gpu [n_paths] [n_steps] [n_devices (0 = all)]      Build: make -C examples multi_gpu
#include "mcamd.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

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

    // ---- nested Monte Carlo, sharded by outer path (hello.cu's bullet option on a smaller grid of points) ----
    mcamd_option bullet = opt;
    bullet.B = 120.0; bullet.P1 = 10; bullet.P2 = 50; bullet.use_window = 1;     // hello.cu:11-13
    mcamd_sim nmc{};
    nmc.n_paths = 4096; nmc.n_paths_local = 4096; nmc.n_steps = 100; nmc.n_paths_inner = 200;
    nmc.seed = 1235; nmc.precision = MCAMD_F64;                                  // inner seed (inc/wrappers.cuh:163)
    std::vector<void *> traj(R, nullptr), pts(R, nullptr);
    std::vector<int32_t *> cnt(R, nullptr);
    int rc = MCAMD_OK;
    for (int i = 0; i < R && rc == MCAMD_OK; ++i) {   // one buffer per device, sized for that device's shard
        uint64_t lo = 0, n = 0;
        mcamd_ctx *c = nullptr;
        rc = mcamd_group_shard(group, &nmc, i, &lo, &n);
        if (rc == MCAMD_OK) rc = mcamd_group_ctx(group, i, &c);
        void *counts = nullptr;
        if (rc == MCAMD_OK) rc = mcamd_device_malloc(c, n * nmc.n_steps * sizeof(double), &traj[i]);
        if (rc == MCAMD_OK) rc = mcamd_device_malloc(c, n * nmc.n_steps * sizeof(int32_t), &counts);
        if (rc == MCAMD_OK) rc = mcamd_device_malloc(c, n * nmc.n_steps * sizeof(double), &pts[i]);
        cnt[i] = static_cast<int32_t *>(counts);
    }
    mcamd_result rn{};
    if (rc == MCAMD_OK)
        rc = mcamd_group_nmc_fused(group, &bullet, &nmc, /*outer_seed*/ 1234, MCAMD_STEP_MAJOR, traj.data(), cnt.data(),
                                   pts.data(), &rn);
    if (rc == MCAMD_OK)
        std::printf("nested MC on %d device(s): %llu points x %u inner paths, mean point price %.12f, lane efficiency %.3f, %.3f ms\n",
                    R, static_cast<unsigned long long>(rn.n), nmc.n_paths_inner, rn.price, rn.live_steps / rn.work_steps,
                    rn.kernel_ms);
    else
        std::fprintf(stderr, "mcamd (nested MC): %s\n", mcamd_last_error());
    for (int i = 0; i < R; ++i) {
        mcamd_ctx *c = nullptr;
        if (mcamd_group_ctx(group, i, &c) != MCAMD_OK) continue;
        mcamd_device_free(c, traj[i]);
        mcamd_device_free(c, cnt[i]);
        mcamd_device_free(c, pts[i]);
    }
    mcamd_group_destroy(group);
    return rc == MCAMD_OK ? 0 : 1;
}
