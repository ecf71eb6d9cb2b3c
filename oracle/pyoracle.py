"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when present, the
reference builds under oracle/_ref/.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF_DIR = os.path.join(_HERE, "_ref")


def build(force: bool = False) -> str:
    """Compile oracle.c with gcc if the .so is missing or older than its sources."""
    srcs = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle.h")]
    stale = force or not os.path.exists(_LIB) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
    if stale:
        subprocess.check_call(
            ["gcc", "-O2", "-fPIC", "-std=c11", "-ffp-contract=off", "-fopenmp", "-shared",
             srcs[0], "-o", _LIB, "-lm"])
    return _LIB


def build_ref() -> None:
    """Run oracle/build_ref.sh (no-op on a machine without /root/reference)."""
    subprocess.check_call(["bash", os.path.join(_HERE, "build_ref.sh")])


class Params(C.Structure):
    _fields_ = [
        ("S0", C.c_double), ("T", C.c_double), ("K", C.c_double), ("r", C.c_double),
        ("v", C.c_double), ("B", C.c_double),
        ("P1", C.c_int32), ("P2", C.c_int32),
        ("n_paths", C.c_uint64),
        ("n_steps", C.c_uint32), ("n_paths_inner", C.c_uint32),
        ("seed", C.c_uint64),
        ("use_window", C.c_int32),
        ("Ik", C.c_int32),
        ("Sk", C.c_double),
        ("Tk", C.c_int32),
        ("dt", C.c_double),
    ]


def make_params(S0=100.0, T=1.0, K=100.0, r=0.1, v=0.2, B=0.0, P1=0, P2=0, n_paths=0, n_steps=1,
                n_paths_inner=0, seed=1234, use_window=0, Ik=0, Sk=0.0, Tk=0, dt=0.0) -> Params:
    return Params(S0, T, K, r, v, B, P1, P2, n_paths, n_steps, n_paths_inner, seed, use_window,
                  Ik, Sk, Tk, dt)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        u64, u32, i32, f32, f64 = C.c_uint64, C.c_uint32, C.c_int32, C.c_float, C.c_double
        pf32, pf64, pi32, pu32 = (C.POINTER(C.c_float), C.POINTER(C.c_double),
                                  C.POINTER(C.c_int32), C.POINTER(C.c_uint32))
        L.oracle_philox4x32_10.argtypes = [u64, u64, u64, pu32]
        L.oracle_philox4x32_10.restype = None
        L.oracle_normal4_f32.argtypes = [u64, u64, u64, pf32]
        L.oracle_normal2_f64.argtypes = [u64, u64, u64, pf64]
        L.oracle_generate_normals_f32.argtypes = [u64, u64, pf32]
        L.oracle_generate_normals_f64.argtypes = [u64, u64, pf64]
        L.oracle_cnd_f32.argtypes = [f32]
        L.oracle_cnd_f32.restype = f32
        L.oracle_bs_call_f32.argtypes = [f32] * 5
        L.oracle_bs_call_f32.restype = f32
        L.oracle_bs_call_f64.argtypes = [f64] * 5
        L.oracle_bs_call_f64.restype = f64
        L.oracle_price_from_normals_f32.argtypes = [pf32, u64, u32, f32, f32, f32, f32, f32, f32, pf32]
        L.oracle_price_from_normals_f32.restype = f32
        L.oracle_price_from_normals_f64.argtypes = [pf64, u64, u32, f64, f64, f64, f64, f64, f64, pf64]
        L.oracle_price_from_normals_f64.restype = f64
        L.oracle_mc_paths.argtypes = [C.POINTER(Params), C.c_int, u64, u64, pf64, pf64, pi32, pf64, pf64,
                                      C.c_int]
        L.oracle_mc_paths.restype = None
        L.oracle_mc_paths_vr.argtypes = [C.POINTER(Params), C.c_int, u64, u64, C.c_int, f64, pf64, C.c_int]
        L.oracle_mc_paths_vr.restype = None
        L.oracle_nmc_point.argtypes = [C.POINTER(Params), C.c_int, u64, u32, f64, i32]
        L.oracle_nmc_point.restype = f64
        L.oracle_finalize.argtypes = [f64, f64, u64, f64, f64, pf64, pf64, pf64, pf64]
        L.oracle_sum_f32.argtypes = [pf32, u64]
        L.oracle_sum_f32.restype = f64
        L.oracle_sum_f64.argtypes = [pf64, u64]
        L.oracle_sum_f64.restype = f64
        L.oracle_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype)) if a is not None else None


# ---- RNG ----
def philox(seed: int, subsequence: int, block: int) -> np.ndarray:
    out = np.zeros(4, dtype=np.uint32)
    lib().oracle_philox4x32_10(seed, subsequence, block, _p(out, C.c_uint32))
    return out


def normal4_f32(seed, subsequence, block) -> np.ndarray:
    out = np.zeros(4, dtype=np.float32)
    lib().oracle_normal4_f32(seed, subsequence, block, _p(out, C.c_float))
    return out


def normal2_f64(seed, subsequence, block) -> np.ndarray:
    out = np.zeros(2, dtype=np.float64)
    lib().oracle_normal2_f64(seed, subsequence, block, _p(out, C.c_double))
    return out


def generate_normals(seed: int, n: int, precision: int = 32) -> np.ndarray:
    if precision == 32:
        out = np.zeros(n, dtype=np.float32)
        lib().oracle_generate_normals_f32(seed, n, _p(out, C.c_float))
    else:
        out = np.zeros(n, dtype=np.float64)
        lib().oracle_generate_normals_f64(seed, n, _p(out, C.c_double))
    return out


# ---- closed form ----
def cnd_f32(x: float) -> float:
    return float(lib().oracle_cnd_f32(x))


def bs_call_f32(S0, K, T, r, sigma) -> float:
    return float(lib().oracle_bs_call_f32(S0, K, T, r, sigma))


def bs_call_f32_grid(n: int) -> float:
    L = lib()
    L.oracle_bs_call_f32_grid.argtypes = [C.c_int]
    L.oracle_bs_call_f32_grid.restype = C.c_double
    return float(L.oracle_bs_call_f32_grid(n))


def bs_call_f64(S0, K, T, r, sigma) -> float:
    return float(lib().oracle_bs_call_f64(S0, K, T, r, sigma))


# ---- array-driven ----
def price_from_normals(normals: np.ndarray, n_paths: int, n_steps: int, S0, sigma, r, K, T):
    """Returns (undiscounted mean, per-path payoffs); dtype follows `normals`."""
    if normals.dtype == np.float32:
        dt = np.float32(T) / np.float32(n_steps)
        sq = np.sqrt(dt, dtype=np.float32)
        pay = np.zeros(n_paths, dtype=np.float32)
        nm = np.ascontiguousarray(normals)
        m = lib().oracle_price_from_normals_f32(_p(nm, C.c_float), n_paths, n_steps, S0, sigma, float(sq), r,
                                                K, float(dt), _p(pay, C.c_float))
    else:
        dt = T / n_steps
        sq = np.sqrt(dt)
        pay = np.zeros(n_paths, dtype=np.float64)
        nm = np.ascontiguousarray(normals, dtype=np.float64)
        m = lib().oracle_price_from_normals_f64(_p(nm, C.c_double), n_paths, n_steps, S0, sigma, sq, r, K, dt,
                                                _p(pay, C.c_double))
    return float(m), pay


# ---- RNG-driven MC ----
def mc_paths(params: Params, precision: int, path_lo: int, n_local: int, want_payoffs=False,
             want_traj=False, want_counts=False, threads: int = 1):
    """Returns dict(sum, sumsq, payoffs?, traj? [nsim, n_local], counts? [nsim, n_local])."""
    nsim = params.n_steps - params.Tk
    pay = np.zeros(n_local, dtype=np.float64) if want_payoffs else None
    traj = np.zeros((nsim, n_local), dtype=np.float64) if want_traj else None
    cnt = np.zeros((nsim, n_local), dtype=np.int32) if want_counts else None
    s, s2 = C.c_double(0), C.c_double(0)
    lib().oracle_mc_paths(C.byref(params), precision, path_lo, n_local, _p(pay, C.c_double),
                          _p(traj, C.c_double), _p(cnt, C.c_int32), C.byref(s), C.byref(s2), threads)
    return {"sum": s.value, "sumsq": s2.value, "payoffs": pay, "traj": traj, "counts": cnt}


def mc_paths_vr(params: Params, precision: int, path_lo: int, n_local: int, antithetic: bool, control_mean: float,
                threads: int = 1) -> np.ndarray:
    """Five raw sums {sum y, sum y^2, sum c, sum c^2, sum y c} of the variance-reduced estimator."""
    sums = np.zeros(5, dtype=np.float64)
    lib().oracle_mc_paths_vr(C.byref(params), precision, path_lo, n_local, int(antithetic), control_mean,
                             _p(sums, C.c_double), threads)
    return sums


def nmc_point(params: Params, precision: int, point_id: int, step: int, St: float, count: int) -> float:
    return float(lib().oracle_nmc_point(C.byref(params), precision, point_id, step, St, count))


def finalize(sum_, sumsq, n, r, T):
    p, se, lo, hi = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    lib().oracle_finalize(sum_, sumsq, n, r, T, C.byref(p), C.byref(se), C.byref(lo), C.byref(hi))
    return {"price": p.value, "std_err": se.value, "ci_lo": lo.value, "ci_hi": hi.value}


def max_threads() -> int:
    return int(lib().oracle_max_threads())


# ---- reference builds (oracle/_ref) ----
def ref_bs():
    path = os.path.join(_REF_DIR, "libref_bs.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.ref_CND.argtypes = [C.c_float]
    L.ref_CND.restype = C.c_float
    L.ref_black_scholes_CPU.argtypes = [C.c_float] * 5
    L.ref_black_scholes_CPU.restype = C.c_float
    if hasattr(L, "ref_black_scholes_grid"):
        L.ref_black_scholes_grid.argtypes = [C.c_int]
        L.ref_black_scholes_grid.restype = C.c_double
    return L


def ref_cpumc():
    path = os.path.join(_REF_DIR, "libref_cpumc.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    f, i, pf = C.c_float, C.c_int, C.POINTER(C.c_float)
    L.ref_sizeof_OptionData.restype = i
    L.ref_simulateOptionPriceCPU.argtypes = [f, f, f, f, f, i]
    L.ref_simulateOptionPriceCPU.restype = f
    L.ref_simulateBulletOptionPriceCPU.argtypes = [f, f, f, f, f, f, i, i, i, i]
    L.ref_simulateBulletOptionPriceCPU.restype = f
    L.ref_simulateOptionPriceCPU_array.argtypes = [i, i, pf, f, f, f, f, f, f, pf]
    L.ref_simulateOptionPriceCPU_array.restype = f
    L.ref_mt19937_normals.argtypes = [C.c_uint, i, pf]
    L.ref_mt19937_normals.restype = None
    return L
