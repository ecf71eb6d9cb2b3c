"""ctypes binding of the C ABI in include/mcamd.h (libmcamd.so).

Host-side plumbing only: device buffers and streams come from the caller (torch tensors'
data_ptr / torch.cuda.current_stream in the tests and bench).  Nothing here computes a price:
every call goes to the HIP library, and loading fails loudly if the library is missing.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
# MCAMD_LIB: diagnostic override for same-box A/B runs of a variant build (build.build_variant, tools/ab_lib.py)
LIB_PATH = os.environ.get("MCAMD_LIB") or os.path.join(PKG, "libmcamd.so")

OK, ERR_INVALID, ERR_HIP, ERR_NODEVICE, ERR_NOMEM = 0, 1, 2, 3, 4
F32, F64 = 32, 64
STEP_MAJOR, PATH_MAJOR = 0, 1
NMC_WAVE_PER_POINT, NMC_BLOCK_PER_POINT, NMC_BLOCK_PER_POINT_PLAIN = 0, 1, 2
FLAG_LOG_SPACE, FLAG_ANTITHETIC, FLAG_CONTROL_VARIATE, FLAG_SEPARATE_REDUCE, FLAG_PRODUCT_FORM = 1, 2, 4, 8, 16
REDUCE_SEQUENTIAL, REDUCE_FIRST_ADD, REDUCE_UNROLL_LAST, REDUCE_GRID_STRIDE = 3, 4, 5, 6

# every symbol include/mcamd.h declares
EXPORTS = [
    "mcamd_abi_version", "mcamd_last_error", "mcamd_build_id", "mcamd_device_count", "mcamd_ctx_create", "mcamd_ctx_destroy",
    "mcamd_get_device_info", "mcamd_device_malloc", "mcamd_device_free", "mcamd_memcpy_to_host",
    "mcamd_memcpy_to_device", "mcamd_price_paths", "mcamd_price_paths_enqueue", "mcamd_enqueued_kernel_ms",
    "mcamd_finalize_stats", "mcamd_group_create", "mcamd_group_destroy", "mcamd_group_size",
    "mcamd_group_price_paths", "mcamd_group_ctx", "mcamd_group_shard", "mcamd_group_simulate_trajectories",
    "mcamd_group_nmc_inner", "mcamd_group_nmc_fused", "mcamd_simulate_trajectories_enqueue", "mcamd_diag_store_pattern", "mcamd_nmc_inner_enqueue",
    "mcamd_nmc_fused_enqueue", "mcamd_finalize_nmc_stats", "mcamd_simulate_trajectories", "mcamd_price_from_normals",
    "mcamd_generate_normals", "mcamd_reduce_sum", "mcamd_reduce_partials", "mcamd_cpu_mc_f32", "mcamd_nmc_inner", "mcamd_nmc_fused", "mcamd_finalize", "mcamd_finalize_cv", "mcamd_cnd_f32",
    "mcamd_bs_call_f32", "mcamd_bs_call_f64",
]


class Option(C.Structure):
    _fields_ = [("S0", C.c_double), ("T", C.c_double), ("K", C.c_double), ("r", C.c_double), ("v", C.c_double),
                ("B", C.c_double), ("P1", C.c_int32), ("P2", C.c_int32), ("use_window", C.c_int32),
                ("Ik", C.c_int32), ("Sk", C.c_double), ("Tk", C.c_int32), ("reserved", C.c_int32), ("dt", C.c_double)]


class Sim(C.Structure):
    _fields_ = [("n_paths", C.c_uint64), ("path_offset", C.c_uint64), ("n_paths_local", C.c_uint64),
                ("n_steps", C.c_uint32), ("n_paths_inner", C.c_uint32), ("seed", C.c_uint64),
                ("precision", C.c_int32), ("flags", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("sum", C.c_double), ("sumsq", C.c_double), ("n", C.c_uint64), ("price", C.c_double),
                ("std_err", C.c_double), ("ci_lo", C.c_double), ("ci_hi", C.c_double), ("kernel_ms", C.c_float),
                ("total_ms", C.c_float), ("grid", C.c_uint32), ("block", C.c_uint32), ("sum_c", C.c_double),
                ("sum_cc", C.c_double), ("sum_yc", C.c_double), ("cv_beta", C.c_double), ("cv_rho", C.c_double),
                ("work_steps", C.c_double), ("live_steps", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 256), ("arch", C.c_char * 64), ("total_mem", C.c_uint64),
                ("free_mem", C.c_uint64), ("compute_units", C.c_int32), ("wavefront_size", C.c_int32),
                ("max_threads_per_block", C.c_int32), ("clock_khz", C.c_int32), ("mem_clock_khz", C.c_int32),
                ("mem_bus_bits", C.c_int32), ("lds_per_block", C.c_int32), ("regs_per_block", C.c_int32),
                ("l2_bytes", C.c_int32), ("device_index", C.c_int32), ("device_count", C.c_int32),
                ("reserved", C.c_int32)]


class McamdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mcamd error {code}: {msg}")
        self.code = code


_lib = None


def load() -> C.CDLL:
    """Loads libmcamd.so; raises if it has not been built (there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: run `python __graft_entry__.py build` "
                          "(hipcc --offload-arch=gfx950); the engine has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, u64, i32, f32, f64 = C.c_void_p, C.c_uint64, C.c_int, C.c_float, C.c_double
    L.mcamd_abi_version.restype = i32
    L.mcamd_last_error.restype = C.c_char_p
    L.mcamd_build_id.restype = C.c_char_p
    L.mcamd_device_count.argtypes = [C.POINTER(i32)]
    L.mcamd_ctx_create.argtypes = [i32, vp, C.POINTER(vp)]
    L.mcamd_ctx_destroy.argtypes = [vp]
    L.mcamd_get_device_info.argtypes = [vp, C.POINTER(DeviceInfo)]
    L.mcamd_device_malloc.argtypes = [vp, u64, C.POINTER(vp)]
    L.mcamd_device_free.argtypes = [vp, vp]
    L.mcamd_memcpy_to_host.argtypes = [vp, vp, vp, u64]
    L.mcamd_memcpy_to_device.argtypes = [vp, vp, vp, u64]
    L.mcamd_price_paths.argtypes = [vp, C.POINTER(Option), C.POINTER(Sim), C.POINTER(Result)]
    L.mcamd_price_paths_enqueue.argtypes = [vp, C.POINTER(Option), C.POINTER(Sim), vp]
    L.mcamd_enqueued_kernel_ms.argtypes = [vp, C.c_uint32, C.POINTER(f32)]
    L.mcamd_finalize_stats.argtypes = [C.POINTER(f64), f64, f64, i32, C.POINTER(Result)]
    L.mcamd_group_create.argtypes = [i32, C.POINTER(i32), C.POINTER(vp)]
    L.mcamd_group_destroy.argtypes = [vp]
    L.mcamd_group_size.argtypes = [vp, C.POINTER(i32)]
    L.mcamd_group_price_paths.argtypes = [vp, C.POINTER(Option), C.POINTER(Sim), C.POINTER(Result)]
    L.mcamd_simulate_trajectories.argtypes = [vp, C.POINTER(Option), C.POINTER(Sim), i32, vp, vp, vp,
                                              C.POINTER(Result)]
    L.mcamd_simulate_trajectories_enqueue.argtypes = [vp, C.POINTER(Option), C.POINTER(Sim), i32, vp, vp, vp, vp]
    L.mcamd_nmc_inner_enqueue.argtypes = [vp, C.POINTER(Option), C.POINTER(Sim), i32, i32, vp, vp, vp, vp]
    L.mcamd_nmc_fused_enqueue.argtypes = [vp, C.POINTER(Option), C.POINTER(Sim), u64, i32, vp, vp, vp, vp]
    L.mcamd_finalize_nmc_stats.argtypes = [C.POINTER(f64), C.POINTER(Result)]
    pvp = C.POINTER(vp)
    L.mcamd_group_ctx.argtypes = [vp, i32, C.POINTER(vp)]
    L.mcamd_group_shard.argtypes = [vp, C.POINTER(Sim), i32, C.POINTER(u64), C.POINTER(u64)]
    L.mcamd_group_simulate_trajectories.argtypes = [vp, C.POINTER(Option), C.POINTER(Sim), i32, pvp, pvp, pvp,
                                                    C.POINTER(Result)]
    L.mcamd_group_nmc_inner.argtypes = [vp, C.POINTER(Option), C.POINTER(Sim), i32, i32, pvp, pvp, pvp, C.POINTER(Result)]
    L.mcamd_group_nmc_fused.argtypes = [vp, C.POINTER(Option), C.POINTER(Sim), u64, i32, pvp, pvp, pvp, C.POINTER(Result)]
    L.mcamd_diag_store_pattern.argtypes = [vp, u64, C.c_uint32, i32, vp, vp, C.POINTER(f32)]
    L.mcamd_price_from_normals.argtypes = [vp, C.POINTER(Option), C.POINTER(Sim), vp, vp, C.POINTER(Result)]
    L.mcamd_generate_normals.argtypes = [vp, u64, u64, i32, vp, C.POINTER(f32)]
    L.mcamd_reduce_sum.argtypes = [vp, vp, u64, i32, i32, C.POINTER(f64), C.POINTER(f32)]
    L.mcamd_reduce_partials.argtypes = [vp, vp, u64, i32, i32, C.c_uint32, C.POINTER(f64), C.POINTER(f32)]
    L.mcamd_cpu_mc_f32.argtypes = [C.POINTER(Option), u64, C.c_uint32, u64, i32, C.POINTER(f32), C.POINTER(f32)]
    L.mcamd_nmc_inner.argtypes = [vp, C.POINTER(Option), C.POINTER(Sim), i32, i32, vp, vp, vp, C.POINTER(Result)]
    L.mcamd_nmc_fused.argtypes = [vp, C.POINTER(Option), C.POINTER(Sim), u64, i32, vp, vp, vp, C.POINTER(Result)]
    L.mcamd_finalize.argtypes = [f64, f64, u64, f64, f64, C.POINTER(Result)]
    L.mcamd_finalize_cv.argtypes = [C.POINTER(f64), u64, f64, f64, C.POINTER(Result)]
    L.mcamd_cnd_f32.argtypes = [f32]
    L.mcamd_cnd_f32.restype = f32
    L.mcamd_bs_call_f32.argtypes = [f32] * 5
    L.mcamd_bs_call_f32.restype = f32
    L.mcamd_bs_call_f64.argtypes = [f64] * 5
    L.mcamd_bs_call_f64.restype = f64
    for name in EXPORTS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("mcamd_abi_version",):
            fn.restype = C.c_int
    _lib = L
    return L


def _check(rc: int):
    if rc != OK:
        raise McamdError(rc, load().mcamd_last_error().decode())


def make_option(S0=100.0, T=1.0, K=100.0, r=0.1, v=0.2, B=0.0, P1=0, P2=0, use_window=0, Ik=0, Sk=0.0,
                Tk=0, dt=0.0) -> Option:
    return Option(S0, T, K, r, v, B, P1, P2, use_window, Ik, Sk, Tk, 0, dt)


def make_sim(n_paths, n_steps=1, precision=F64, seed=1234, path_offset=0, n_paths_local=None,
             n_paths_inner=0, flags=0) -> Sim:
    return Sim(n_paths, path_offset, n_paths if n_paths_local is None else n_paths_local, n_steps, n_paths_inner,
               seed, precision, flags)


def _ptr(t):
    """device pointer of a torch tensor (or None / int passthrough)"""
    if t is None:
        return None
    if isinstance(t, int):
        return C.c_void_p(t)
    return C.c_void_p(t.data_ptr())


def finalize(sum_, sumsq, n, r, T) -> Result:
    res = Result()
    _check(load().mcamd_finalize(sum_, sumsq, n, r, T, C.byref(res)))
    return res


def finalize_cv(sums, n, r, T) -> Result:
    res = Result()
    arr = (C.c_double * 5)(*sums)
    _check(load().mcamd_finalize_cv(arr, n, r, T, C.byref(res)))
    return res


def finalize_stats(stats6, r, T, control_variate=False) -> Result:
    res = Result()
    arr = (C.c_double * 6)(*[float(x) for x in stats6])
    _check(load().mcamd_finalize_stats(arr, r, T, int(control_variate), C.byref(res)))
    return res


def finalize_nmc_stats(stats6) -> Result:
    res = Result()
    arr = (C.c_double * 6)(*[float(x) for x in stats6])
    _check(load().mcamd_finalize_nmc_stats(arr, C.byref(res)))
    return res


def cpu_mc_f32(opt: Option, n_paths: int, n_steps: int, seed: int = 0, from_random_device: bool = False):
    """(price, undiscounted fp32 payoff sum) of the reference's serial CPU Monte Carlo (mcamd_cpu_mc_f32)"""
    p, s = C.c_float(0), C.c_float(0)
    _check(load().mcamd_cpu_mc_f32(C.byref(opt), n_paths, n_steps, seed, int(from_random_device), C.byref(p), C.byref(s)))
    return p.value, s.value


def cnd_f32(x):
    return float(load().mcamd_cnd_f32(x))


def bs_call_f32(S0, K, T, r, sigma):
    return float(load().mcamd_bs_call_f32(S0, K, T, r, sigma))


def bs_call_f64(S0, K, T, r, sigma):
    return float(load().mcamd_bs_call_f64(S0, K, T, r, sigma))


def build_id() -> str:
    return load().mcamd_build_id().decode()


def device_count() -> int:
    n = C.c_int(0)
    rc = load().mcamd_device_count(C.byref(n))
    return n.value if rc == OK else 0


class Context:
    """Owns one mcamd_ctx (one device, one stream)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._L = load()
        self._h = C.c_void_p()
        _check(self._L.mcamd_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self._h)))

    def close(self):
        if self._h:
            self._L.mcamd_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def device_info(self) -> DeviceInfo:
        info = DeviceInfo()
        _check(self._L.mcamd_get_device_info(self._h, C.byref(info)))
        return info

    def price_paths(self, opt: Option, sim: Sim) -> Result:
        res = Result()
        _check(self._L.mcamd_price_paths(self._h, C.byref(opt), C.byref(sim), C.byref(res)))
        return res

    def price_paths_enqueue(self, opt: Option, sim: Sim, stats) -> None:
        """Asynchronous: leaves {sum, sumsq, sum_c, sum_cc, sum_yc, n} in the device tensor `stats` (>= 6 doubles)."""
        _check(self._L.mcamd_price_paths_enqueue(self._h, C.byref(opt), C.byref(sim), _ptr(stats)))

    def enqueued_kernel_ms(self, n_last: int):
        arr = (C.c_float * n_last)()
        _check(self._L.mcamd_enqueued_kernel_ms(self._h, n_last, arr))
        return list(arr)

    def simulate_trajectories(self, opt: Option, sim: Sim, traj, counts=None, payoffs=None,
                              layout=STEP_MAJOR) -> Result:
        res = Result()
        _check(self._L.mcamd_simulate_trajectories(self._h, C.byref(opt), C.byref(sim), layout, _ptr(traj),
                                                   _ptr(counts), _ptr(payoffs), C.byref(res)))
        return res

    def simulate_trajectories_enqueue(self, opt: Option, sim: Sim, traj, counts, payoffs, stats, layout=STEP_MAJOR) -> None:
        _check(self._L.mcamd_simulate_trajectories_enqueue(self._h, C.byref(opt), C.byref(sim), layout, _ptr(traj),
                                                           _ptr(counts), _ptr(payoffs), _ptr(stats)))

    def nmc_inner_enqueue(self, opt: Option, sim: Sim, prices, counts, point_prices, stats, layout=STEP_MAJOR,
                          variant=NMC_WAVE_PER_POINT) -> None:
        _check(self._L.mcamd_nmc_inner_enqueue(self._h, C.byref(opt), C.byref(sim), layout, variant, _ptr(prices),
                                               _ptr(counts), _ptr(point_prices), _ptr(stats)))

    def nmc_fused_enqueue(self, opt: Option, sim: Sim, outer_seed: int, prices, counts, point_prices, stats,
                          layout=STEP_MAJOR) -> None:
        _check(self._L.mcamd_nmc_fused_enqueue(self._h, C.byref(opt), C.byref(sim), outer_seed, layout, _ptr(prices),
                                               _ptr(counts), _ptr(point_prices), _ptr(stats)))

    def diag_store_pattern(self, n_paths_local: int, n_steps: int, precision: int, traj, payoffs=None) -> float:
        """kernel ms of the store kernel's pure store stream (diagnostic: the same-run HBM write ceiling)"""
        ms = C.c_float(0)
        _check(self._L.mcamd_diag_store_pattern(self._h, n_paths_local, n_steps, precision, _ptr(traj), _ptr(payoffs),
                                                C.byref(ms)))
        return ms.value

    def price_from_normals(self, opt: Option, sim: Sim, normals, payoffs=None) -> Result:
        res = Result()
        _check(self._L.mcamd_price_from_normals(self._h, C.byref(opt), C.byref(sim), _ptr(normals), _ptr(payoffs),
                                                C.byref(res)))
        return res

    def generate_normals(self, seed: int, n: int, precision: int, out) -> float:
        ms = C.c_float(0)
        _check(self._L.mcamd_generate_normals(self._h, seed, n, precision, _ptr(out), C.byref(ms)))
        return ms.value

    def reduce_sum(self, x, n: int, precision: int, variant: int = REDUCE_GRID_STRIDE):
        s, ms = C.c_double(0), C.c_float(0)
        _check(self._L.mcamd_reduce_sum(self._h, _ptr(x), n, precision, variant, C.byref(s), C.byref(ms)))
        return s.value, ms.value

    def reduce_partials(self, x, n: int, precision: int, variant: int, n_blocks: int):
        """one partial sum per workgroup (they add up to the complete sum) and the kernel's ms"""
        arr, ms = (C.c_double * n_blocks)(), C.c_float(0)
        _check(self._L.mcamd_reduce_partials(self._h, _ptr(x), n, precision, variant, n_blocks, arr, C.byref(ms)))
        return list(arr), ms.value

    def nmc_inner(self, opt: Option, sim: Sim, prices, counts, point_prices, layout=STEP_MAJOR,
                  variant=NMC_WAVE_PER_POINT) -> Result:
        res = Result()
        _check(self._L.mcamd_nmc_inner(self._h, C.byref(opt), C.byref(sim), layout, variant, _ptr(prices),
                                       _ptr(counts), _ptr(point_prices), C.byref(res)))
        return res

    def nmc_fused(self, opt: Option, sim: Sim, outer_seed: int, prices, counts, point_prices,
                  layout=STEP_MAJOR) -> Result:
        res = Result()
        _check(self._L.mcamd_nmc_fused(self._h, C.byref(opt), C.byref(sim), outer_seed, layout, _ptr(prices),
                                       _ptr(counts), _ptr(point_prices), C.byref(res)))
        return res


class Group:
    """Single-process multi-GPU group: one context per device + an RCCL communicator clique (mcamd_group_*)."""

    def __init__(self, n_devices: int = 0, devices=None):
        self._L = load()
        self._h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices) if devices else None
        _check(self._L.mcamd_group_create(len(devices) if devices else n_devices, arr, C.byref(self._h)))

    def close(self):
        if self._h:
            self._L.mcamd_group_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def size(self) -> int:
        n = C.c_int(0)
        _check(self._L.mcamd_group_size(self._h, C.byref(n)))
        return n.value

    def price_paths(self, opt: Option, sim: Sim) -> Result:
        res = Result()
        _check(self._L.mcamd_group_price_paths(self._h, C.byref(opt), C.byref(sim), C.byref(res)))
        return res

    def shard(self, sim: Sim, i: int):
        """(first global path id, number of paths) device i of the group works on for this job."""
        lo, n = C.c_uint64(0), C.c_uint64(0)
        _check(self._L.mcamd_group_shard(self._h, C.byref(sim), i, C.byref(lo), C.byref(n)))
        return lo.value, n.value

    @staticmethod
    def _ptrs(per_device):
        """array of one device pointer per device (None: no such output)"""
        if per_device is None:
            return None
        return (C.c_void_p * len(per_device))(*[None if t is None else t.data_ptr() for t in per_device])

    def simulate_trajectories(self, opt: Option, sim: Sim, traj, counts=None, payoffs=None, layout=STEP_MAJOR) -> Result:
        res = Result()
        _check(self._L.mcamd_group_simulate_trajectories(self._h, C.byref(opt), C.byref(sim), layout, self._ptrs(traj),
                                                         self._ptrs(counts), self._ptrs(payoffs), C.byref(res)))
        return res

    def nmc_inner(self, opt: Option, sim: Sim, prices, counts, point_prices, layout=STEP_MAJOR,
                  variant=NMC_WAVE_PER_POINT) -> Result:
        res = Result()
        _check(self._L.mcamd_group_nmc_inner(self._h, C.byref(opt), C.byref(sim), layout, variant, self._ptrs(prices),
                                             self._ptrs(counts), self._ptrs(point_prices), C.byref(res)))
        return res

    def nmc_fused(self, opt: Option, sim: Sim, outer_seed: int, prices, counts, point_prices, layout=STEP_MAJOR) -> Result:
        res = Result()
        _check(self._L.mcamd_group_nmc_fused(self._h, C.byref(opt), C.byref(sim), outer_seed, layout, self._ptrs(prices),
                                             self._ptrs(counts), self._ptrs(point_prices), C.byref(res)))
        return res
