// group.cpp — single-process multi-GPU entry points of the C ABI (mcamd_group_*): one context per device,
// path-sharded pricing on all devices at once, and ONE RCCL all-reduce of the 6-double statistics record
// over xGMI.  This is the route for a C++ host (the reference's main() is a single process); bench.py uses
// the other standard shape, one process per GPU with torch.distributed, on the same enqueue primitive.
//
// The reference has no multi-GPU code.  Shards are contiguous global path-id ranges, and the
// Philox subsequence of a path is its global id, so the result equals the single-GPU result up to fp64
// summation order.  RCCL is loaded with dlopen when the first group is created: libmcamd.so keeps no link-time
// dependency on it, and a process that already carries an RCCL (PyTorch ships its own) reuses that one.
#include "mcamd.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

int mcamd_set_error_(int code, const char *msg);  // capi.cpp: records the thread's last error string

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
int g_rccl_status = MCAMD_OK;
std::string g_rccl_error;

int load_rccl_once();

// Loads RCCL exactly once per process, whichever thread asks first; later callers get the recorded outcome.
int load_rccl()
{
    std::call_once(g_rccl_once, [] { g_rccl_status = load_rccl_once(); if (g_rccl_status) g_rccl_error = mcamd_last_error(); });
    if (g_rccl_status) return mcamd_set_error_(g_rccl_status, g_rccl_error.c_str());
    return MCAMD_OK;
}

int load_rccl_once()
{
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return mcamd_set_error_(MCAMD_ERR_HIP, "RCCL not found (dlopen librccl.so.1): multi-GPU groups need it");
    Rccl r;
    r.handle = h;
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.AllReduce || !r.GetErrorString)
        return mcamd_set_error_(MCAMD_ERR_HIP, "RCCL library lacks an expected symbol");
    g_rccl = r;
    return MCAMD_OK;
}

int nccl_fail(ncclResult_t e, const char *what)
{
    std::string msg = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "RCCL error");
    return mcamd_set_error_(MCAMD_ERR_HIP, msg.c_str());
}

int hip_fail(hipError_t e, const char *what)
{
    (void)hipGetLastError();
    std::string msg = std::string(what) + ": " + hipGetErrorString(e);
    return mcamd_set_error_(e == hipErrorOutOfMemory ? MCAMD_ERR_NOMEM : MCAMD_ERR_HIP, msg.c_str());
}

}  // namespace

struct mcamd_group {
    std::vector<int> devices;
    std::vector<mcamd_ctx *> ctx;
    std::vector<hipStream_t> streams;
    std::vector<double *> d_stats;  // 8 doubles per device
    std::vector<ncclComm_t> comms;
};

namespace {

// waits for everything enqueued on the group's streams (error paths: nothing may still be running when the call
// returns); errors of the wait itself are dropped, the caller is already reporting one
void drain(mcamd_group *g)
{
    for (size_t i = 0; i < g->devices.size(); ++i) {
        if (hipSetDevice(g->devices[i]) == hipSuccess && g->streams[i]) (void)hipStreamSynchronize(g->streams[i]);
        (void)hipGetLastError();
    }
}

// device i's shard of [path_offset, path_offset + n_paths_local): contiguous, sizes differ by at most one
mcamd_sim shard_of(const mcamd_sim &sim, int R, int i)
{
    const uint64_t base = sim.n_paths_local / R, rem = sim.n_paths_local % R;
    mcamd_sim s = sim;
    s.path_offset = sim.path_offset + static_cast<uint64_t>(i) * base + (static_cast<uint64_t>(i) < rem ? i : rem);
    s.n_paths_local = base + (static_cast<uint64_t>(i) < rem ? 1 : 0);
    return s;
}

// The shape every group call shares: device i enqueues its shard (enqueue(i, shard) -> one of the *_enqueue entry
// points, which leaves a 6-double statistics record in g->d_stats[i]), then the one collective of the path sums the
// records over the devices, every device is waited for, and the reduced record comes back to the host.
template <typename Enqueue>
int run_sharded(mcamd_group *g, const mcamd_sim *sim, double (&stats)[8], float *kernel_ms, Enqueue enqueue)
{
    const int R = static_cast<int>(g->devices.size());
    for (int i = 0; i < R; ++i) {
        // asynchronous: every device starts its shard before any host wait
        if (int rc = enqueue(i, shard_of(*sim, R, i))) {
            const std::string why = mcamd_last_error();
            drain(g);   // shards already enqueued on devices 0..i-1 must not outlive the failed call
            return mcamd_set_error_(rc, why.c_str());
        }
    }
    // A group that was started is always ended, whatever an AllReduce returned, so a failed call cannot leave
    // RCCL's thread-local group open for the next one.
    ncclResult_t ne = g_rccl.GroupStart();
    if (ne == ncclSuccess) {
        ncclResult_t first = ncclSuccess;
        for (int i = 0; i < R && first == ncclSuccess; ++i)
            first = g_rccl.AllReduce(g->d_stats[i], g->d_stats[i], 6, ncclDouble, ncclSum, g->comms[i], g->streams[i]);
        const ncclResult_t end = g_rccl.GroupEnd();
        ne = first != ncclSuccess ? first : end;
    }
    if (ne != ncclSuccess) {
        drain(g);
        return nccl_fail(ne, "ncclAllReduce");
    }
    hipError_t sync_err = hipSuccess;
    for (int i = 0; i < R; ++i) {   // every device is waited for, also after one of them has failed
        hipError_t e = hipSetDevice(g->devices[i]);
        if (e == hipSuccess) e = hipStreamSynchronize(g->streams[i]);
        if (e != hipSuccess && sync_err == hipSuccess) sync_err = e;
    }
    if (sync_err != hipSuccess) return hip_fail(sync_err, "group synchronise");
    *kernel_ms = 0.0f;
    for (int i = 0; i < R; ++i) {
        float ms = 0.0f;
        if (int rc = mcamd_enqueued_kernel_ms(g->ctx[i], 1, &ms)) return rc;
        *kernel_ms = std::fmax(*kernel_ms, ms);
    }
    hipError_t e = hipSetDevice(g->devices[0]);
    if (e == hipSuccess) e = hipMemcpy(stats, g->d_stats[0], 6 * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "group result copy");
    return MCAMD_OK;
}

}  // namespace

extern "C" {

int mcamd_group_create(int n_devices, const int *devices, mcamd_group **out)
{
    if (!out) return mcamd_set_error_(MCAMD_ERR_INVALID, "group out-pointer is NULL");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        (void)hipGetLastError();
        return mcamd_set_error_(MCAMD_ERR_NODEVICE, "no HIP device visible: this engine has no CPU fallback");
    }
    if (n_devices <= 0) n_devices = count;  // all visible devices
    if (n_devices > count) return mcamd_set_error_(MCAMD_ERR_INVALID, "more devices requested than visible");
    if (int rc = load_rccl()) return rc;
    mcamd_group *g = new (std::nothrow) mcamd_group;
    if (!g) return mcamd_set_error_(MCAMD_ERR_NOMEM, "out of host memory");
    for (int i = 0; i < n_devices; ++i) g->devices.push_back(devices ? devices[i] : i);
    for (int i = 0; i < n_devices; ++i) {
        mcamd_ctx *c = nullptr;
        hipStream_t s = nullptr;
        double *d = nullptr;
        int rc = MCAMD_OK;
        hipError_t e = hipSetDevice(g->devices[i]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc(&d, 8 * sizeof(double));
        if (e != hipSuccess) rc = hip_fail(e, "group device setup");
        if (!rc) rc = mcamd_ctx_create(g->devices[i], s, &c);
        g->ctx.push_back(c);
        g->streams.push_back(s);
        g->d_stats.push_back(d);
        if (rc) {
            mcamd_group_destroy(g);
            return rc;
        }
    }
    g->comms.resize(n_devices, nullptr);
    // single-process communicator clique over the device list: the RCCL ring/tree runs over xGMI
    ncclResult_t ne = g_rccl.CommInitAll(g->comms.data(), n_devices, g->devices.data());
    if (ne != ncclSuccess) {
        g->comms.clear();
        mcamd_group_destroy(g);
        return nccl_fail(ne, "ncclCommInitAll");
    }
    *out = g;
    return MCAMD_OK;
}

int mcamd_group_destroy(mcamd_group *g)
{
    if (!g) return MCAMD_OK;
    for (size_t i = 0; i < g->comms.size(); ++i)
        if (g->comms[i]) (void)g_rccl.CommDestroy(g->comms[i]);
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        (void)hipSetDevice(g->devices[i]);
        if (g->ctx[i]) (void)mcamd_ctx_destroy(g->ctx[i]);
        if (g->d_stats[i]) (void)hipFree(g->d_stats[i]);
        if (g->streams[i]) (void)hipStreamDestroy(g->streams[i]);
    }
    delete g;
    return MCAMD_OK;
}

int mcamd_group_size(mcamd_group *g, int *n_devices)
{
    if (!g || !n_devices) return mcamd_set_error_(MCAMD_ERR_INVALID, "group and n_devices must be non-NULL");
    *n_devices = static_cast<int>(g->devices.size());
    return MCAMD_OK;
}

int mcamd_group_ctx(mcamd_group *g, int i, mcamd_ctx **ctx)
{
    if (!g || !ctx) return mcamd_set_error_(MCAMD_ERR_INVALID, "group and ctx must be non-NULL");
    if (i < 0 || i >= static_cast<int>(g->ctx.size())) return mcamd_set_error_(MCAMD_ERR_INVALID, "device index out of range");
    *ctx = g->ctx[i];
    return MCAMD_OK;
}

int mcamd_group_shard(mcamd_group *g, const mcamd_sim *sim, int i, uint64_t *path_offset, uint64_t *n_paths_local)
{
    if (!g || !sim || !path_offset || !n_paths_local)
        return mcamd_set_error_(MCAMD_ERR_INVALID, "group, sim and the out-pointers must be non-NULL");
    if (i < 0 || i >= static_cast<int>(g->ctx.size())) return mcamd_set_error_(MCAMD_ERR_INVALID, "device index out of range");
    const mcamd_sim s = shard_of(*sim, static_cast<int>(g->devices.size()), i);
    *path_offset = s.path_offset;
    *n_paths_local = s.n_paths_local;
    return MCAMD_OK;
}

int mcamd_group_price_paths(mcamd_group *g, const mcamd_option *opt, const mcamd_sim *sim, mcamd_result *res)
{
    if (!g || !opt || !sim || !res) return mcamd_set_error_(MCAMD_ERR_INVALID, "group, opt, sim and res must be non-NULL");
    double stats[8] = {0};
    float kernel_ms = 0.0f;
    if (int rc = run_sharded(g, sim, stats, &kernel_ms, [&](int i, const mcamd_sim &s) {
            return mcamd_price_paths_enqueue(g->ctx[i], opt, &s, g->d_stats[i]);
        }))
        return rc;
    if (int rc = mcamd_finalize_stats(stats, opt->r, opt->T, (sim->flags & MCAMD_FLAG_CONTROL_VARIATE) != 0, res)) return rc;
    res->kernel_ms = kernel_ms;  // slowest device's simulation kernel
    res->total_ms = kernel_ms;
    res->block = 256;
    return MCAMD_OK;
}

int mcamd_group_simulate_trajectories(mcamd_group *g, const mcamd_option *opt, const mcamd_sim *sim, int layout,
                                      void *const *d_traj, int32_t *const *d_counts, void *const *d_payoffs,
                                      mcamd_result *res)
{
    if (!g || !opt || !sim || !res) return mcamd_set_error_(MCAMD_ERR_INVALID, "group, opt, sim and res must be non-NULL");
    if (!d_traj && sim->n_paths_local) return mcamd_set_error_(MCAMD_ERR_INVALID, "d_traj (one device pointer per device) is NULL");
    double stats[8] = {0};
    float kernel_ms = 0.0f;
    if (int rc = run_sharded(g, sim, stats, &kernel_ms, [&](int i, const mcamd_sim &s) {
            return mcamd_simulate_trajectories_enqueue(g->ctx[i], opt, &s, layout, d_traj ? d_traj[i] : nullptr,
                                                       d_counts ? d_counts[i] : nullptr, d_payoffs ? d_payoffs[i] : nullptr,
                                                       g->d_stats[i]);
        }))
        return rc;
    if (int rc = mcamd_finalize_stats(stats, opt->r, opt->T, 0, res)) return rc;
    res->kernel_ms = kernel_ms;
    res->total_ms = kernel_ms;
    res->block = 256;
    return MCAMD_OK;
}

int mcamd_group_nmc_inner(mcamd_group *g, const mcamd_option *opt, const mcamd_sim *sim, int layout, int variant,
                          const void *const *d_prices, const int32_t *const *d_counts, void *const *d_point_prices,
                          mcamd_result *res)
{
    if (!g || !opt || !sim || !res) return mcamd_set_error_(MCAMD_ERR_INVALID, "group, opt, sim and res must be non-NULL");
    if ((!d_prices || !d_point_prices) && sim->n_paths_local)
        return mcamd_set_error_(MCAMD_ERR_INVALID, "d_prices and d_point_prices (one device pointer per device) must be non-NULL");
    double stats[8] = {0};
    float kernel_ms = 0.0f;
    if (int rc = run_sharded(g, sim, stats, &kernel_ms, [&](int i, const mcamd_sim &s) {
            return mcamd_nmc_inner_enqueue(g->ctx[i], opt, &s, layout, variant, d_prices ? d_prices[i] : nullptr,
                                           d_counts ? d_counts[i] : nullptr, d_point_prices ? d_point_prices[i] : nullptr,
                                           g->d_stats[i]);
        }))
        return rc;
    if (int rc = mcamd_finalize_nmc_stats(stats, res)) return rc;
    res->kernel_ms = kernel_ms;
    res->total_ms = kernel_ms;
    res->block = 256;
    return MCAMD_OK;
}

int mcamd_group_nmc_fused(mcamd_group *g, const mcamd_option *opt, const mcamd_sim *sim, uint64_t outer_seed, int layout,
                          void *const *d_prices, int32_t *const *d_counts, void *const *d_point_prices, mcamd_result *res)
{
    if (!g || !opt || !sim || !res) return mcamd_set_error_(MCAMD_ERR_INVALID, "group, opt, sim and res must be non-NULL");
    if ((!d_prices || !d_point_prices) && sim->n_paths_local)
        return mcamd_set_error_(MCAMD_ERR_INVALID, "d_prices and d_point_prices (one device pointer per device) must be non-NULL");
    double stats[8] = {0};
    float kernel_ms = 0.0f;
    if (int rc = run_sharded(g, sim, stats, &kernel_ms, [&](int i, const mcamd_sim &s) {
            return mcamd_nmc_fused_enqueue(g->ctx[i], opt, &s, outer_seed, layout, d_prices ? d_prices[i] : nullptr,
                                           d_counts ? d_counts[i] : nullptr, d_point_prices ? d_point_prices[i] : nullptr,
                                           g->d_stats[i]);
        }))
        return rc;
    if (int rc = mcamd_finalize_nmc_stats(stats, res)) return rc;
    res->kernel_ms = kernel_ms;
    res->total_ms = kernel_ms;
    res->block = 256;
    return MCAMD_OK;
}

}  // extern "C"
