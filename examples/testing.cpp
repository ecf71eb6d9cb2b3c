// testing.cpp — the reference's test driver (testing.cu:51-111) re-created on the shim, with the
// comparisons the reference leaves to the eye turned into checks: exits non-zero on a mismatch.
// Writes testing.csv in the reference's format (time,trajectory,value with a t=0 row per trajectory,
// testing.cu:37-47).
#include "monte_carlo.hpp"

#include <cmath>
#include <fstream>

static int failures = 0;

static void expect(bool ok, const char *what)
{
    std::cout << (ok ? "ok   " : "FAIL ") << what << "\n";
    failures += ok ? 0 : 1;
}

static void test_outer(int n_traj, int n_steps, int n_threads_per_block, uint64_t seed = 1234)
{
    Simulation parameters(n_traj, n_steps);
    auto out = parameters.simulate_outer_trajectories(n_threads_per_block, seed);
    std::cout << "Total size: " << out.size() << "\n";
    expect(out.size() == static_cast<size_t>(n_traj) * n_steps, "outer trajectories: size");
    bool positive = true;
    for (float v : out) positive = positive && std::isfinite(v) && v > 0.0f;
    expect(positive, "outer trajectories: finite and positive");

    std::ofstream csv("testing.csv");
    csv << "time,trajectory,value\n";
    for (int i = 0; i < n_traj * n_steps; ++i) {
        const int traj = i / n_steps, step = i % n_steps;
        if (step == 0) csv << 0.0 << "," << traj << "," << parameters.x_0 << "\n";
        csv << (1 + step) * parameters.dt() << "," << traj << "," << out[i] << "\n";
    }
}

int main()
{
    std::cout << "Hello from testing suite!\n";
    mcamd_shim::verbose() = false;
    Simulation default_parameters(1024, 100);

    auto cpu = default_parameters.simulate_trajectory_cpu();
    auto gpu = default_parameters.simulate_trajectory_gpu();
    std::cout << "Final value of CPU trajectory:" << cpu.back() << "\n";
    std::cout << "Len simulations: " << cpu.size() << "\n";
    bool same = cpu.size() == gpu.size();
    for (size_t i = 0; same && i < cpu.size(); ++i) same = std::fabs(cpu[i] - gpu[i]) <= 1e-4f * (1.0f + std::fabs(cpu[i]));
    expect(same, "array-driven payoffs: GPU == CPU");

    const float truth = default_parameters.sum_random_array();
    for (int i = 3; i < 7; ++i) {
        std::cout << "Testing reduction: " << i << "\n";
        auto out = default_parameters.test_reduction(1, 1024, i);
        for (float el : out) std::cout << el << "\n";
        expect(out.size() == 1 && std::fabs(out[0] - truth) < 0.05f, "reduction equals host sum");
        // several blocks: one partial per block (inc/testing.cuh:227-234), and the partials add up to the whole array
        auto parts = default_parameters.test_reduction(7, 1024, i);
        double total = 0.0;
        for (float el : parts) total += el;
        expect(parts.size() == 7 && std::fabs(total - truth) < 0.05, "block partials add up to the host sum");
    }

    test_outer(20, 150, 10, 555);
    return failures ? 1 : 0;
}
