/*
 * mcamd.h — C ABI of the MI355X (gfx950) Monte Carlo option-pricing engine.
 *
 * This is the drop-in boundary for the hot path of amauryrlm/Monte-Carlo-Project-CUDA.
 * The reference has no FFI: its "API" is the header-level C++ surface of inc/wrappers.cuh,
 * inc/tool.cuh, inc/testing.cuh and inc/BlackandScholes.hpp called from hello.cu / testing.cu.
 * Each entry point below names the reference interface (file:line under the reference root)
 * it replaces; include/wrappers.hpp and friends re-expose the reference's own names on top of
 * this ABI.  Plain C types only: pointers, sizes, POD structs, int status codes.
 *
 * Conventions
 *  - every function returns MCAMD_OK (0) or an MCAMD_ERR_* code and never calls exit()
 *    (the reference's testCUDA / CHECK_MALLOC exit the process: inc/tool.cuh:47-53,92-100);
 *    mcamd_last_error() returns the calling thread's last message;
 *  - "d_" pointers are device (HBM) pointers on the context's device, owned by the caller;
 *  - calls are synchronous on the context's stream (as the reference's wrappers are:
 *    cudaDeviceSynchronize at inc/wrappers.cuh:48,79,115,157,233,297) unless named *_enqueue; a
 *    context is not thread-safe, distinct contexts are independent;
 *  - option parameters travel in the structs passed to each call; there is no global
 *    __constant__ symbol to upload first (reference: hello.cu:22, inc/trajectories.cuh:12);
 *  - random numbers: counter-based Philox4x32-10 held in registers, key = seed,
 *    subsequence = GLOBAL path id, i.e. exactly rocrand_init(seed, path_id, 0) followed by
 *    rocrand_normal4 (fp32, 4 steps per block) / rocrand_normal_double2 (fp64, 2 steps per
 *    block).  There is no RNG state array and no setup kernel (replaces setup_kernel,
 *    inc/tool.cuh:192-195, and init_rng_kernel, inc/testing.cuh:95-98).  Results therefore
 *    do not depend on how paths are sharded over GPUs, blocks or threads;
 *  - payoff sums are accumulated in fp64 whatever the path precision;
 *  - the in-register kernels carry ln(St/S0) through the step loop and exponentiate where the price is
 *    needed; MCAMD_FLAG_PRODUCT_FORM asks for the reference's recurrence St *= exp(...) as written
 *    (same draws, same scheme, ~1e-14 relative apart in fp64).
 */
#ifndef MCAMD_H
#define MCAMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCAMD_ABI_VERSION 5

/* status codes */
#define MCAMD_OK 0
#define MCAMD_ERR_INVALID 1   /* bad argument (null pointer, zero steps, shape mismatch ...) */
#define MCAMD_ERR_HIP 2       /* a HIP runtime call or kernel launch failed */
#define MCAMD_ERR_NODEVICE 3  /* no usable gfx950 device */
#define MCAMD_ERR_NOMEM 4     /* device or host allocation failed */

/* arithmetic type of the simulated paths */
#define MCAMD_F32 32
#define MCAMD_F64 64

/* trajectory / point-array layouts */
#define MCAMD_STEP_MAJOR 0 /* a[step * n_paths_local + path]  — coalesced, the engine's native layout */
#define MCAMD_PATH_MAJOR 1 /* a[path * n_steps + step]        — the reference's layout
                              (inc/trajectories.cuh:304-305, inc/testing.cuh:69) */

/* nested-MC strategies; all give the same per-point prices up to fp64 summation order (the third reference strategy,
 * the fused single launch, is mcamd_nmc_fused and equals WAVE_PER_POINT bit for bit) */
#define MCAMD_NMC_WAVE_PER_POINT 0  /* replaces compute_nmc_optimal, inc/nmc.cuh:280-386.  The fast one: with a window,
                                       wavefronts refill lanes whose path is over instead of waiting for the last */
#define MCAMD_NMC_BLOCK_PER_POINT 1 /* replaces compute_nmc_one_block_per_point, inc/nmc.cuh:12-108: one workgroup per point;
                                       with a window each of its wavefronts compacts its share of the point's paths */
#define MCAMD_NMC_BLOCK_PER_POINT_PLAIN 2 /* the same without lane compaction (path j on thread j mod 256, a wavefront waits
                                       for its last lane): the independent implementation the compacting kernels are
                                       fuzzed against (tools/fuzz_nmc.py); 2-3x slower with the reference's window */

/* mcamd_sim.flags */
#define MCAMD_FLAG_LOG_SPACE 1 /* accepted for compatibility and redundant: it names what the in-register kernels do by
                                  default (ABI <= 4 had it as an opt-in).  They carry ln(St/S0) through the step loop
                                  instead of St — one add (window-less) or one fma and a compare (barrier window) per step
                                  instead of an exponential and a multiply; every step still draws its normal, and the price
                                  is exponentiated where it is needed.  The same scheme (the exact GBM step,
                                  inc/trajectories.cuh:146) with the product of the step factors re-associated into the
                                  exponential of their sum; rounding differs at ~1e-14 relative in fp64 (measured: 17 % /
                                  21 % less time for window-less pricing in fp64 / fp32 with the pair sums of
                                  csrc/mc_device.hpp, 12 % for the nested-MC inner stage; same oracle tolerances). */
#define MCAMD_FLAG_PRODUCT_FORM 16 /* mcamd_price_paths, mcamd_nmc_inner, mcamd_nmc_fused: carry the price itself,
                                  St *= exp(...) every step — the reference's recurrence as written — and test the barrier
                                  on it.  Keeps the in-register terminal price the same BITS as the last row
                                  mcamd_simulate_trajectories stores for the same path, and the barrier counts the same
                                  integers as the stored ones.  (The store and array-driven kernels always run this form:
                                  they need St at every step.)  Not combinable with MCAMD_FLAG_LOG_SPACE. */

#define MCAMD_FLAG_ANTITHETIC 2 /* opt-in (mcamd_price_paths): a sample is the antithetic pair (G, -G) of one path's
                                  normals; its payoff is the pair's mean; n counts pairs.  New capability. */
#define MCAMD_FLAG_CONTROL_VARIATE 4 /* opt-in (mcamd_price_paths): S_T as control variate (E[S_T] = S e^{rT} is
                                  known); the call also returns the three cross sums and the finalized price uses the
                                  sample-optimal beta (mcamd_finalize_cv).  New capability. */

#define MCAMD_FLAG_SEPARATE_REDUCE 8 /* diagnostic (mcamd_price_paths[_enqueue]): sum the block records with a separate
                                  one-workgroup launch even where the simulation kernel would finish the sum itself (jobs
                                  of up to 8192 workgroups: its last workgroup to arrive does it).  Same summation order,
                                  same bits; exists so that a test can show exactly that. */

/* reduce variants: names follow the reference's ReductionType (inc/testing.cuh:100-106) */
#define MCAMD_REDUCE_SEQUENTIAL 3
#define MCAMD_REDUCE_FIRST_ADD 4
#define MCAMD_REDUCE_UNROLL_LAST 5
#define MCAMD_REDUCE_GRID_STRIDE 6

typedef struct mcamd_ctx mcamd_ctx;

/* Option parameters: the reference's OptionData (inc/tool.cuh:13-26) in double precision,
 * plus the restart triple the bullet kernels accept (inc/trajectories.cuh:116-117,140-143). */
typedef struct mcamd_option {
    double S0;          /* spot */
    double T;           /* maturity */
    double K;           /* strike */
    double r;           /* risk-free rate */
    double v;           /* volatility */
    double B;           /* barrier level (bullet option) */
    int32_t P1, P2;     /* payoff only if P1 <= #steps{B > St} <= P2 */
    int32_t use_window; /* 0: European call (no barrier test at all); 1: bullet window */
    int32_t Ik;         /* restart: initial barrier count */
    double Sk;          /* restart: initial price; 0 means S0 (inc/trajectories.cuh:141) */
    int32_t Tk;         /* restart: steps already elapsed; n_steps - Tk are simulated */
    int32_t reserved;
    double dt;          /* time step of the multi-step kernels, the reference's OptionData.step
                           (inc/tool.cuh:25, read at inc/trajectories.cuh:131, inc/nmc.cuh:28); 0 = T / n_steps.
                           The discount stays exp(-r T), as in the reference (inc/wrappers.cuh:85). */
} mcamd_option;

/* Simulation shape. The job has n_paths paths; this call simulates the shard
 * [path_offset, path_offset + n_paths_local) of it.  A single-GPU caller passes
 * path_offset = 0, n_paths_local = n_paths.  n_paths_local == 0 is a legal empty shard
 * (all-zero statistics, nothing launched). */
typedef struct mcamd_sim {
    uint64_t n_paths;
    uint64_t path_offset;
    uint64_t n_paths_local;
    uint32_t n_steps;       /* N_STEPS; dt = T / n_steps; 1 = the exact one-step pricer */
    uint32_t n_paths_inner; /* N_PATHS_INNER (nested MC only) */
    uint64_t seed;          /* the reference hard-codes 1234 / 1235 (inc/wrappers.cuh:41,163) */
    int32_t precision;      /* MCAMD_F32 or MCAMD_F64 */
    int32_t flags;          /* OR of MCAMD_FLAG_* (0 = the reference's plain estimator) */
} mcamd_sim;

/* Result of a pricing call. sum/sumsq/n are the shard's raw fp64 statistics (what a multi-GPU
 * caller all-reduces); price..ci_hi are finalized from them as if the shard were the whole job
 * (see mcamd_finalize). */
typedef struct mcamd_result {
    double sum;        /* sum of undiscounted payoffs */
    double sumsq;      /* sum of squared undiscounted payoffs */
    uint64_t n;        /* paths in the shard */
    double price;      /* exp(-rT) * sum / n              (inc/wrappers.cuh:51,85,118) */
    double std_err;    /* exp(-rT) * sqrt(s^2 / n)        (new: the reference has no variance) */
    double ci_lo;      /* price -/+ 1.96 std_err */
    double ci_hi;
    float kernel_ms;   /* HIP-event time of the simulation kernel alone, on the context's stream */
    float total_ms;    /* HIP-event time of the whole call's device work (kernel + final reduce + D2H; equal to kernel_ms
                          when the kernel finishes its own sum: MCAMD_FLAG_SEPARATE_REDUCE) */
    uint32_t grid;     /* launch shape the engine chose (threadsPerBlock / number_of_blocks of the */
    uint32_t block;    /* reference wrappers are accepted by the shim and ignored) */
    /* control variate (MCAMD_FLAG_CONTROL_VARIATE), zero otherwise: c = S_T - E[S_T] per sample */
    double sum_c;      /* sum of c */
    double sum_cc;     /* sum of c^2 */
    double sum_yc;     /* sum of payoff * c */
    double cv_beta;    /* sample-optimal coefficient cov(y, c) / var(c) used in price */
    double cv_rho;     /* sample correlation of payoff and control; variance shrinks by 1 - rho^2 */
    /* nested MC (mcamd_nmc_inner / mcamd_nmc_fused), zero otherwise: inner path-steps the kernel executed, counted as
     * 64 lanes x the steps each wavefront ran (a wavefront leaves a point's step loop as soon as every lane's barrier
     * count is beyond P2, where the payoff can no longer be non-zero): the work figure for throughput / roofline */
    double work_steps;
    /* of those, the lane-steps taken by paths whose window was still open (live_steps / work_steps = how full the
     * wavefronts ran; equal to work_steps without a window) */
    double live_steps;
} mcamd_result;

/* Device report: replaces getDeviceProperty (inc/tool.cuh:56-88) and the free/total memory
 * print of get_max_blocks (inc/tool.cuh:176-188). */
typedef struct mcamd_device_info {
    char name[256];
    char arch[64];
    uint64_t total_mem, free_mem;
    int32_t compute_units, wavefront_size, max_threads_per_block, clock_khz, mem_clock_khz, mem_bus_bits;
    int32_t lds_per_block, regs_per_block, l2_bytes, device_index, device_count, reserved;
} mcamd_device_info;

int mcamd_abi_version(void);
const char *mcamd_last_error(void);
/* 16-hex-digit hash of the kernel sources and compile flags this library was built from; profiles/valu_slots.json
 * (the ISA issue-slot counts bench.py prices the VALU roofline with) carries the same id when it describes this build */
const char *mcamd_build_id(void);
int mcamd_device_count(int *count);

/* hip_stream: a hipStream_t to launch on (e.g. the caller framework's current stream), or NULL
 * to let the context create its own non-blocking stream.  Note that the legacy default stream's handle IS
 * NULL (torch.cuda.current_stream().cuda_stream == 0 unless a stream was made current): to share the default
 * stream pass hipStreamLegacy, or — as bench.py does — make an explicit stream current and pass that.
 * Scratch buffers are owned by the context and reused. */
int mcamd_ctx_create(int device, void *hip_stream, mcamd_ctx **ctx);
int mcamd_ctx_destroy(mcamd_ctx *ctx);
int mcamd_get_device_info(mcamd_ctx *ctx, mcamd_device_info *info);

/* Device memory helpers so that a host program needs nothing but this library (the reference's
 * wrappers call cudaMalloc / cudaMemcpy / cudaFree directly: inc/wrappers.cuh:39,49,55).
 * Copies are synchronous with respect to the context's stream. */
int mcamd_device_malloc(mcamd_ctx *ctx, uint64_t bytes, void **d_ptr);
int mcamd_device_free(mcamd_ctx *ctx, void *d_ptr);
int mcamd_memcpy_to_host(mcamd_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes);
int mcamd_memcpy_to_device(mcamd_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes);

/* In-register Monte Carlo: RNG -> GBM steps -> payoff -> fp64 (sum, sumsq); nothing is stored.
 * Replaces the kernel + host tail of
 *   wrapper_gpu_option_vanilla          inc/wrappers.cuh:33-57   (n_steps = 1)
 *     simulateOptionPriceMultipleBlockGPUwithReduce  inc/trajectories.cuh:54-113
 *   wrapper_gpu_bullet_option[_atomic]  inc/wrappers.cuh:59-125  (use_window = 1)
 *     simulateBulletOptionPriceMultipleBlockGPU[atomic]  inc/trajectories.cuh:115-271
 * and is the multi-step European pricer of BASELINE configs 2 and 5 (use_window = 0).
 * The library picks the kernel: window jobs of millions of paths (plain estimator) run the lane-compacting kernel, other
 * window jobs one path per thread, window-less jobs the pair-sum loop (two paths per thread: ln(S_T/S_0) from the sum of
 * the path's normals, a Box-Muller pair contributing sqrt2 r sin(a + pi/4)); a path's payoff does not depend on which
 * (same Philox stream, same arithmetic), so any sharding of a job gives the same sums up to fp64 summation order.
 * Jobs of up to 8192 workgroups are ONE launch: the kernel's last workgroup sums the block records in a fixed order
 * (see MCAMD_FLAG_SEPARATE_REDUCE). */
int mcamd_price_paths(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, mcamd_result *res);

/* Asynchronous form of mcamd_price_paths: enqueues the simulation kernel (and, for large grids, the final reduction) on
 * the context's stream and returns without waiting.  d_stats (device, >= 6 doubles) receives {sum, sumsq, sum_c, sum_cc, sum_yc, n}
 * (the cross sums are zero without MCAMD_FLAG_CONTROL_VARIATE).  The caller orders later work on the same stream —
 * typically ONE all-reduce of d_stats over the ranks — and finalizes after synchronising (mcamd_finalize_stats), so
 * a multi-step driver pays no host round trip per step.  New (the reference is fully synchronous). */
int mcamd_price_paths_enqueue(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, double *d_stats);
/* HIP-event times (ms) of the simulation kernels of the last n_last (<= 64) enqueued calls, oldest first.
 * Synchronises the stream. */
int mcamd_enqueued_kernel_ms(mcamd_ctx *ctx, uint32_t n_last, float *ms);
/* Host: finalize a (possibly all-reduced) 6-double stats record copied back from the device. */
int mcamd_finalize_stats(const double stats[6], double r, double T, int control_variate, mcamd_result *res);

/* Trajectory store: as above, and every St (and, if d_counts != NULL, every running barrier
 * count) is written to HBM.  d_traj: n_sim_steps * n_paths_local elements of the path precision
 * (n_sim_steps = n_steps - Tk); d_counts: same shape, int32, or NULL; d_payoffs: n_paths_local
 * undiscounted payoffs of the path precision, or NULL.
 * Replaces simulateOptionPriceMultipleBlockGPU (trajectory overload) inc/testing.cuh:46-73 and
 * simulate_outer_trajectories inc/trajectories.cuh:273-351. */
int mcamd_simulate_trajectories(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, int layout,
                                void *d_traj, int32_t *d_counts, void *d_payoffs, mcamd_result *res);

/* Diagnostic: mcamd_simulate_trajectories' launch shape and store stream (step-major rows, one 16-byte non-temporal
 * store per lane per step, then the payoff row if d_payoffs != NULL) with nothing simulated: the HBM write ceiling of
 * this access pattern on this device, measured beside the real kernel (bench.py roofline_store.same_run_ceiling).
 * Requirements of the vector store path: n_paths_local a multiple of 4 (fp32) / 2 (fp64), 16-byte aligned buffers.
 * kernel_ms: HIP-event time of the kernel.  The buffers are overwritten with a counter pattern. */
int mcamd_diag_store_pattern(mcamd_ctx *ctx, uint64_t n_paths_local, uint32_t n_steps, int precision, void *d_traj,
                             void *d_payoffs, float *kernel_ms);

/* Array-driven pricer: normals are an input, d_normals[path * n_steps + step] (the reference's
 * layout), precision per sim->precision; d_payoffs (nullable): n_paths_local payoffs.
 * Replaces simulateOptionPriceGPU / simulateOptionPriceMultipleBlockGPU (array overloads)
 * inc/trajectories.cuh:14-52; CPU twin inc/testing.cuh:75-91. */
int mcamd_price_from_normals(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, const void *d_normals,
                             void *d_payoffs, mcamd_result *res);

/* Bulk N(0,1) fill of d_out[0..n): one Philox sequence (subsequence 0) consumed front to back,
 * 4 floats / 2 doubles per block.  Replaces generate_random_array's curandGenerateNormal,
 * inc/testing.cuh:17-24. */
int mcamd_generate_normals(mcamd_ctx *ctx, uint64_t seed, uint64_t n, int precision, void *d_out, float *kernel_ms);

/* Sum of d_in[0..n) (fp32 or fp64 elements) accumulated in fp64.  variant selects the schedule
 * named after the reference's reduce3..reduce6 (inc/reduce.cuh:9-227); *sum is the complete sum
 * for every variant (the reference leaves a per-block array for the caller to finish,
 * inc/testing.cuh:227-234). */
int mcamd_reduce_sum(mcamd_ctx *ctx, const void *d_in, uint64_t n, int precision, int variant, double *sum,
                     float *kernel_ms);

/* As mcamd_reduce_sum, but with the reference's launch shape and result shape: the schedule runs on n_blocks workgroups
 * and h_partials (HOST, n_blocks doubles) receives one partial sum per workgroup, as reduce3..6 leave one float per
 * block in g_odata for the caller to finish (inc/reduce.cuh:9-227; Simulation::test_reduction copies them back,
 * inc/testing.cuh:185-235).  Unlike reduce3..5 — whose blocks cover 2 * blockDim elements each and nothing beyond
 * n_blocks of those — every element is covered for any n_blocks (blocks stride over the array), so the partials
 * always add up to the complete sum.  1 <= n_blocks <= 2^20. */
int mcamd_reduce_partials(mcamd_ctx *ctx, const void *d_in, uint64_t n, int precision, int variant, uint32_t n_blocks,
                          double *h_partials, float *kernel_ms);

/* Nested Monte Carlo, inner stage: for every stored point (step, path) of the shard, n_paths_inner
 * continuation paths of n_steps - 1 - step steps from (d_prices, d_counts), windowed payoff, mean,
 * discount exp(-rT).  d_prices / d_counts are what mcamd_simulate_trajectories wrote (same layout
 * argument); d_point_prices receives one value of the path precision per point in that layout.
 * Inner stream: seed = sim->seed, subsequence = global_point_id * n_paths_inner + j with
 * global_point_id = global_path * n_steps + step (global_path = path_offset + local path): a shard of a job prices
 * its points from the same streams as the whole job.  WAVE_PER_POINT pools the continuation paths of 8 outer paths
 * whose GLOBAL ids share id / 8, so a point's price is bit-identical under any sharding wherever its pool lies
 * inside the shard, and differs by fp64 summation order only in a shard's partial first / last pool;
 * BLOCK_PER_POINT and the window-less job are bit-identical under any sharding.
 * Replaces compute_nmc_one_block_per_point / compute_nmc_optimal, inc/nmc.cuh:12-108,280-386.
 * res->sum / n: sum and count of the per-point prices (the wrappers' scalar diagnostic,
 * inc/wrappers.cuh:185-189,316-321). */
int mcamd_nmc_inner(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, int layout, int variant,
                    const void *d_prices, const int32_t *d_counts, void *d_point_prices, mcamd_result *res);

/* Nested Monte Carlo, outer + inner stage fused in ONE launch: the wavefronts of a persistent grid first simulate and
 * store the outer paths, 64 at a time from a device-scope queue (seed = outer_seed), publish them (release / acquire
 * across the device's L2s), then price the points exactly as mcamd_nmc_inner(MCAMD_NMC_WAVE_PER_POINT) does (inner
 * seed = sim->seed).  d_prices / d_counts / d_point_prices are OUTPUTS here (same shapes and layout rule as above; d_counts may be NULL for
 * use_window = 0).  Results are bit-identical to mcamd_simulate_trajectories + mcamd_nmc_inner with the same
 * two seeds.  Replaces compute_nmc_one_block_per_point_with_outter, inc/nmc.cuh:113-275
 * (wrapper_gpu_bullet_option_nmc_one_kernel, inc/wrappers.cuh:209-266). */
int mcamd_nmc_fused(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, uint64_t outer_seed, int layout,
                    void *d_prices, int32_t *d_counts, void *d_point_prices, mcamd_result *res);

/* Asynchronous forms of the trajectory store and the nested-MC calls, as mcamd_price_paths_enqueue: the simulation
 * kernel and the final reduction are enqueued on the context's stream, nothing waits on the host.  d_stats (device,
 * >= 6 doubles) receives
 *   store:      {sum, sumsq, 0, 0, 0, n}                              -> mcamd_finalize_stats
 *   nested MC:  {sum of point prices, sum of their squares, wave-steps executed, live lane-steps, 0, n points}
 *                                                                     -> mcamd_finalize_nmc_stats
 * so ONE all-reduce of 6 doubles carries a shard whichever call produced it.  The output arrays are complete when
 * the stream reaches the point after the call.  mcamd_enqueued_kernel_ms covers these calls too.  New (the
 * reference is fully synchronous). */
int mcamd_simulate_trajectories_enqueue(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, int layout,
                                        void *d_traj, int32_t *d_counts, void *d_payoffs, double *d_stats);
int mcamd_nmc_inner_enqueue(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, int layout, int variant,
                            const void *d_prices, const int32_t *d_counts, void *d_point_prices, double *d_stats);
int mcamd_nmc_fused_enqueue(mcamd_ctx *ctx, const mcamd_option *opt, const mcamd_sim *sim, uint64_t outer_seed,
                            int layout, void *d_prices, int32_t *d_counts, void *d_point_prices, double *d_stats);
/* Host: sum / sumsq / n / price (= mean point price) / work_steps / live_steps of a (possibly all-reduced) nested-MC
 * statistics record copied back from the device. */
int mcamd_finalize_nmc_stats(const double stats[6], mcamd_result *res);

/* ---- single-process multi-GPU (the shape of the reference's own main(): one host process) ----
 * A group owns one context per device and an RCCL communicator clique over them (ncclCommInitAll; RCCL is loaded
 * with dlopen on first use).  mcamd_group_price_paths splits [path_offset, path_offset + n_paths_local) of the job
 * into contiguous per-device shards, runs them concurrently (mcamd_price_paths_enqueue on every device), sums the
 * 6-double statistics records with ONE ncclAllReduce over xGMI and finalizes.  Because a path's random numbers
 * depend only on its global id, the result equals the single-device result up to fp64 summation order.
 * devices == NULL: devices 0..n_devices-1; n_devices <= 0: all visible devices.  The reference has no multi-GPU
 * code.  One-process-per-GPU hosts use mcamd_price_paths_enqueue + their own collective (bench.py). */
typedef struct mcamd_group mcamd_group;
int mcamd_group_create(int n_devices, const int *devices, mcamd_group **group);
int mcamd_group_destroy(mcamd_group *group);
int mcamd_group_size(mcamd_group *group, int *n_devices);
int mcamd_group_price_paths(mcamd_group *group, const mcamd_option *opt, const mcamd_sim *sim, mcamd_result *res);

/* The other two shards of SURVEY 8e: "trajectory-store mode shards the [step][path] buffer by path columns per GPU; no
 * exchange" and "NMC shards by outer path; per-point prices stay on the owning GPU".  The reference's store and
 * nested-MC hosts are single-process C++ (inc/trajectories.cuh:273-351, inc/wrappers.cuh:128-340), which is the shape
 * these calls serve.  Device i of the group works on the shard mcamd_group_shard reports for it (contiguous global
 * path ids, sizes differing by at most one) and writes into the caller's buffers ON THAT DEVICE: d_traj[i], d_counts[i],
 * d_payoffs[i], d_prices[i], d_point_prices[i] are arrays of n_devices device pointers, each sized for its own shard
 * (shape rules of the single-device calls with n_paths_local = the shard's; a NULL array = the single-device call's
 * NULL; entries of empty shards are ignored).  mcamd_group_ctx hands out device i's context for mcamd_device_malloc /
 * mcamd_memcpy_*.  All devices run concurrently (the *_enqueue forms); the only exchange is ONE ncclAllReduce of the
 * 6-double statistics record; res is finalized from the reduced record (store: price / SE / CI of the whole job;
 * nested MC: sum / mean / count of all point prices, executed and live lane-steps), kernel_ms = the slowest device's
 * simulation kernel.  Results equal the single-device call's: stored columns and per-point prices as documented at
 * mcamd_nmc_inner (bit-identical outside a shard's partial edge pools), statistics up to fp64 summation order. */
int mcamd_group_ctx(mcamd_group *group, int i, mcamd_ctx **ctx);
int mcamd_group_shard(mcamd_group *group, const mcamd_sim *sim, int i, uint64_t *path_offset, uint64_t *n_paths_local);
int mcamd_group_simulate_trajectories(mcamd_group *group, const mcamd_option *opt, const mcamd_sim *sim, int layout,
                                      void *const *d_traj, int32_t *const *d_counts, void *const *d_payoffs,
                                      mcamd_result *res);
int mcamd_group_nmc_inner(mcamd_group *group, const mcamd_option *opt, const mcamd_sim *sim, int layout, int variant,
                          const void *const *d_prices, const int32_t *const *d_counts, void *const *d_point_prices,
                          mcamd_result *res);
int mcamd_group_nmc_fused(mcamd_group *group, const mcamd_option *opt, const mcamd_sim *sim, uint64_t outer_seed,
                          int layout, void *const *d_prices, int32_t *const *d_counts, void *const *d_point_prices,
                          mcamd_result *res);

/* Host: discount + mean + standard error + 95% CI from (sum, sumsq, n) — after an all-reduce
 * over shards, or directly.  Fills price/std_err/ci_* (and copies sum/sumsq/n) in *res. */
int mcamd_finalize(double sum, double sumsq, uint64_t n, double r, double T, mcamd_result *res);

/* Host: as mcamd_finalize for a control-variate run.  sums = {sum y, sum y^2, sum c, sum c^2, sum y c} with c already
 * centred on its known mean (what mcamd_price_paths returns in sum, sumsq, sum_c, sum_cc, sum_yc — after an
 * all-reduce over shards, or directly): price = exp(-rT) (ybar - beta cbar), beta = cov(y,c)/var(c),
 * std_err from the residual variance var(y)(1 - rho^2). */
int mcamd_finalize_cv(const double sums[5], uint64_t n, double r, double T, mcamd_result *res);

/* Host: the reference's serial CPU Monte Carlo (the baseline of BASELINE configs[0]), restated: fp32 paths and an fp32
 * running sum, std::mt19937 + std::normal_distribution<float>, one draw per step in path order,
 *   St *= expf((r - v^2/2) dt + v sqrtf(dt) G),   count += (St < B)  [use_window],   payoff max(St - K, 0) if the
 *   window admits the count,   price = expf(-r T) * sum / n_paths.
 * n_steps = 1 with use_window = 0 is simulateOptionPriceCPU (inc/tool.cuh:104-130); n_steps = N_STEPS with the window is
 * simulateBulletOptionPriceCPU (inc/tool.cuh:133-173).  dt = opt->dt, or T / n_steps when that is 0.  The reference seeds
 * from std::random_device and is not reproducible (inc/tool.cuh:116,151): from_random_device != 0 does the same; otherwise
 * the generator is std::mt19937(seed), which makes the path testable.  payoff_sum (nullable) receives the undiscounted fp32
 * sum.  This is the reference's CPU baseline, not a fallback: no GPU entry point ever routes here.  Single thread. */
int mcamd_cpu_mc_f32(const mcamd_option *opt, uint64_t n_paths, uint32_t n_steps, uint64_t seed, int from_random_device,
                     float *price, float *payoff_sum);

/* Host closed form.  _f32 restates the reference's fp32 code path operation for operation
 * (CND: inc/BlackandScholes.hpp:8-30; black_scholes_CPU: :34-43); _f64 is the exact erfc form. */
float mcamd_cnd_f32(float x);
float mcamd_bs_call_f32(float x0, float strike, float T, float r, float sigma);
double mcamd_bs_call_f64(double x0, double strike, double T, double r, double sigma);

#ifdef __cplusplus
}
#endif
#endif /* MCAMD_H */
