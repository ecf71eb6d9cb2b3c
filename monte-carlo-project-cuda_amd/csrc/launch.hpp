// launch.hpp — host-side launcher interface between the C ABI (capi.cpp) and the kernel
// translation units.  Launchers only enqueue work on `stream`; they never synchronise or allocate.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcamd {

// Everything a path kernel needs, in double; launchers narrow to the path precision.
struct PathJob {
    double drift, vol;   // (r - v^2/2) dt, v sqrt(dt)
    double K, B, S_start;
    int32_t P1, P2, Ik;
    uint32_t n_sim;      // steps to simulate
    uint32_t n_steps;    // job's N_STEPS (indexing of points)
    uint64_t seed;
    uint64_t path_offset;
    uint64_t n_local;
    bool window;
    bool logspace;       // MCAMD_FLAG_LOG_SPACE (in-register and nested-MC kernels)
    int vr;              // variance reduction of the in-register kernel: bit 0 antithetic, bit 1 control variate
    double control_mean; // E[S_T] for the control variate
    int precision;       // 32 / 64
};

constexpr uint32_t kMaxGrid = 1u << 20;  // blocks; beyond this the kernels grid-stride

// number of partial records (one per block) a launch with this many local paths writes
// compute_units: of the device the job runs on (0: 256).  d_queue: one 64-bit word of device memory, used (and zeroed on
// `stream`) when the job goes to the lane-compacting window kernel, which pulls its tasks from it.
uint32_t price_grid(const PathJob &job, uint32_t compute_units);

// How a grid's block records become the final record.  ticket != nullptr: the simulation kernel finishes itself — the
// last workgroup to arrive sums the records in small_final_sum's order and writes out[0..N) (and, with n_value >= 0,
// the 6-double statistics layout); *ticket must be zero at launch and is zero again afterwards.  ticket == nullptr:
// the kernel only leaves its records and the caller launches launch_small_final (same order, same bits) or
// launch_final_reduce (grids of more than kFoldMaxRecords records).
struct FinishSpec {
    double *out = nullptr;
    unsigned int *ticket = nullptr;
    double n_value = -1.0;
};
// One workgroup of 256 threads sums up to this many records in a few microseconds; beyond it the separate
// 1024-thread reduction is faster than the lone last workgroup.
constexpr uint32_t kFoldMaxRecords = 8192;

hipError_t launch_price(const PathJob &job, uint32_t compute_units, double *d_partials, unsigned long long *d_queue,
                        uint32_t grid, const FinishSpec &finish, hipStream_t stream);
// the separate launch of grid_finish's sum: n_records <= kFoldMaxRecords records of record_doubles (2 or 5) doubles
hipError_t launch_small_final(const double *d_partials, uint32_t n_records, int record_doubles, double *d_out,
                              hipStream_t stream, double n_value = -1.0);

uint32_t store_grid(uint64_t n_local, int precision);
hipError_t launch_store(const PathJob &job, int layout, void *d_traj, int32_t *d_counts, void *d_payoffs,
                        double *d_partials, uint32_t grid, hipStream_t stream);

// diagnostic: the store kernel's shape (store_grid) and store stream with nothing simulated (n_local % (16 / sizeof(T)) == 0,
// 16-byte aligned buffers, a row's bytes addressable with 32 bits)
hipError_t launch_store_pattern(uint64_t n_local, uint32_t n_sim, int precision, void *d_traj, void *d_payoffs,
                                uint32_t grid, hipStream_t stream);

uint32_t array_grid(uint64_t n_local);
hipError_t launch_from_normals(const PathJob &job, const void *d_normals, void *d_payoffs, double *d_partials,
                               uint32_t grid, hipStream_t stream);

// sums n_records records of record_doubles (2, 4 = kNmcRecord, or 5) doubles into d_out[0..record_doubles)
// n_value >= 0: also zero-fill d_out[record_doubles..5) and write d_out[5] = n_value (the 6-double stats layout)
hipError_t launch_final_reduce(const double *d_partials, uint32_t n_records, int record_doubles, double *d_out,
                               hipStream_t stream, double n_value = -1.0);

hipError_t launch_generate_normals(uint64_t seed, uint64_t n, int precision, void *d_out, hipStream_t stream);

uint32_t reduce_grid(uint64_t n, int variant);
hipError_t launch_reduce(const void *d_in, uint64_t n, int precision, int variant, double *d_partials, uint32_t grid,
                         hipStream_t stream);

struct NmcJob {
    PathJob path;            // inner-path constants (seed = inner seed)
    uint32_t n_inner;
    double discount;         // exp(-r T)
    uint64_t n_points;       // n_local * n_steps
    uint32_t compute_units;  // of the device the job runs on (sizes the persistent grid of the wave-per-point kernel)
};
// the nested-MC kernels write 4-double block records: {sum of point prices, sum of their squares, wave-steps executed
// (x 64 = lane-steps of work), lane-steps of paths whose window was still open}
constexpr int kNmcRecord = 4;
uint32_t nmc_grid(const NmcJob &job, int variant);
// d_queue: one zeroed 64-bit word (the wave-per-point kernel's task counter; the launcher zeroes it on `stream`)
hipError_t launch_nmc_inner(const NmcJob &job, int layout, int variant, const void *d_prices, const int32_t *d_counts,
                            void *d_point_prices, double *d_partials, unsigned long long *d_queue, uint32_t grid,
                            hipStream_t stream);

uint32_t nmc_fused_grid(const NmcJob &job);
// d_queue: the context's 64-byte counter block (three of its words are this kernel's queues; zeroed on `stream`)
hipError_t launch_nmc_fused(const NmcJob &job, uint64_t outer_seed, int layout, void *d_prices, int32_t *d_counts,
                            void *d_point_prices, double *d_partials, unsigned long long *d_queue, uint32_t grid,
                            hipStream_t stream);

}  // namespace mcamd
