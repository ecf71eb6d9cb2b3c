"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/mcamd.h declares, keeps its struct layout, and its host-only entry points (closed form,
finalize, argument errors) behave.  No kernels are launched here."""
import ctypes as C
import importlib
import math
import os
import re

import numpy as np
import pytest

from conftest import ROOT

pkg = importlib.import_module("monte-carlo-project-cuda_amd")
capi = pkg.capi


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.LIB_PATH):
        pkg.build()
    return capi.load()


def test_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "mcamd.h")).read()
    declared = set(re.findall(r"\b(mcamd_[a-z0-9_]+)\s*\(", header))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mcamd_abi_version() == 5


def test_slot_counts_describe_the_built_library(lib):
    # bench.py prices the VALU roofline with ISA slot counts (profiles/valu_slots.json); they carry the id of the
    # sources they were counted from, and build() refreshes them: library, sources and counts must agree
    import json
    bmod = importlib.import_module("monte-carlo-project-cuda_amd.build")
    with open(os.path.join(ROOT, "profiles", "valu_slots.json")) as f:
        counts = json.load(f)
    assert capi.build_id() == bmod.build_id() == counts["build_id"]
    assert 50 < counts["price_f64"] < 120 and 20 < counts["price_f32"] < 40
    # the fixed yardstick of bench.py (W_FLOOR) is a floor: never above the shipped loops' counts,
    # and the default loops are the cheaper ones
    from importlib import util as ilu
    spec = ilu.spec_from_file_location("bench_for_floor", os.path.join(ROOT, "bench.py"))
    bench = ilu.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for key, floor in bench.W_FLOOR.items():
        assert floor <= counts[key] <= 1.1 * floor, (key, floor, counts[key])
        assert counts[key] < counts[key + "_product"]
    assert counts["nmc_wave_f64_window"] < counts["nmc_wave_f64_window_product"]


def test_struct_layout_is_the_documented_abi():
    assert C.sizeof(capi.Option) == 88 and C.sizeof(capi.Sim) == 48
    assert C.sizeof(capi.Result) == 128 and C.sizeof(capi.DeviceInfo) == 384


def test_closed_form_matches_reference_golden_bitwise(lib, golden):
    g = golden("bs_closed_form.json")
    for c in g["call"]:
        assert np.float32(capi.bs_call_f32(c["S0"], c["K"], c["T"], c["r"], c["sigma"])) == np.float32(c["ref_call_f32"])
    for c in g["cnd"]:
        assert np.float32(capi.cnd_f32(c["x"])) == np.float32(c["ref_cnd_f32"])
    assert abs(capi.bs_call_f64(100, 100, 1, 0.1, 0.2) - 13.269676584660893) < 1e-11


def test_closed_form_matches_oracle(lib, oracle):
    rng = np.random.default_rng(11)
    for _ in range(500):
        a = [float(np.float32(x)) for x in (rng.uniform(10, 300), rng.uniform(10, 300), rng.uniform(0.05, 4),
                                            rng.uniform(0, 0.15), rng.uniform(0.05, 0.9))]
        assert np.float32(capi.bs_call_f32(*a)) == np.float32(oracle.bs_call_f32(*a))
        assert math.isclose(capi.bs_call_f64(*a), oracle.bs_call_f64(*a), rel_tol=1e-14, abs_tol=1e-300)


def test_finalize_matches_oracle(lib, oracle):
    x = np.random.default_rng(3).gamma(2.0, 8.0, size=1000)
    res = capi.finalize(float(x.sum()), float((x * x).sum()), x.size, 0.1, 1.0)
    want = oracle.finalize(float(x.sum()), float((x * x).sum()), x.size, 0.1, 1.0)
    for k in ("price", "std_err", "ci_lo", "ci_hi"):
        assert math.isclose(getattr(res, k), want[k], rel_tol=1e-15)
    assert math.isclose(res.std_err, math.exp(-0.1) * x.std(ddof=1) / math.sqrt(x.size), rel_tol=1e-10)
    empty = capi.finalize(0.0, 0.0, 0, 0.1, 1.0)
    assert empty.price == 0.0 and empty.std_err == 0.0 and empty.n == 0


def test_finalize_cv_matches_numpy(lib):
    rng = np.random.default_rng(5)
    c = rng.normal(0.0, 20.0, size=5000)
    y = np.maximum(0.6 * c + rng.normal(8.0, 4.0, size=5000), 0.0)
    sums = [y.sum(), (y * y).sum(), c.sum(), (c * c).sum(), (y * c).sum()]
    res = capi.finalize_cv(sums, y.size, 0.1, 1.0)
    beta = np.cov(y, c, ddof=1)[0, 1] / c.var(ddof=1)
    d = math.exp(-0.1)
    assert math.isclose(res.cv_beta, beta, rel_tol=1e-10)
    assert math.isclose(res.price, d * (y.mean() - beta * c.mean()), rel_tol=1e-12)
    resid = y - beta * c
    assert math.isclose(res.std_err, d * resid.std(ddof=1) / math.sqrt(y.size), rel_tol=1e-3)
    assert math.isclose(res.cv_rho, np.corrcoef(y, c)[0, 1], rel_tol=1e-10)


def test_no_gpu_means_loud_failure_not_fallback(lib):
    if capi.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(capi.McamdError) as e:
        capi.Context(0)
    assert e.value.code == capi.ERR_NODEVICE and "no CPU fallback" in str(e.value)


def test_null_arguments_are_errors_not_crashes(lib):
    assert lib.mcamd_price_paths(None, None, None, None) == capi.ERR_INVALID
    assert b"NULL" in lib.mcamd_last_error()
    assert lib.mcamd_finalize(0.0, 0.0, 0, 0.0, 1.0, None) == capi.ERR_INVALID
    assert lib.mcamd_reduce_sum(None, None, 0, 32, 6, None, None) == capi.ERR_INVALID
    assert lib.mcamd_ctx_destroy(None) == capi.OK


def test_product_does_not_reach_into_the_oracle():
    # the shipped package and headers must not import, link or call anything under oracle/
    for base in ("monte-carlo-project-cuda_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                    text = open(os.path.join(d, f)).read()
                    assert "pyoracle" not in text and "liboracle" not in text and "oracle/" not in text, (d, f)


def test_cpu_mc_f32_is_the_reference_arithmetic_bit_for_bit(lib, oracle):
    """mcamd_cpu_mc_f32 restates inc/tool.cuh:104-173 with a seed.  The reference's own build (oracle/_ref) holds the same
    recurrence with the normals as an input (inc/testing.cuh:75-91) and the std::mt19937 / std::normal_distribution<float>
    stream: feeding one into the other must reproduce the seeded pricer's fp32 payoff sum exactly."""
    ref = oracle.ref_cpumc()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    for n, steps, seed in ((1000, 1, 7), (257, 12, 1234), (64, 100, 99)):
        normals = np.zeros(n * steps, dtype=np.float32)
        ref.ref_mt19937_normals(seed, n * steps, normals.ctypes.data_as(C.POINTER(C.c_float)))
        pay = np.zeros(n, dtype=np.float32)
        dt = np.float32(1.0) / np.float32(steps)
        ref.ref_simulateOptionPriceCPU_array.restype = C.c_float
        mean = ref.ref_simulateOptionPriceCPU_array(n, steps, normals.ctypes.data_as(C.POINTER(C.c_float)), 100.0, 0.2,
                                                    float(np.sqrt(dt)), 0.1, 100.0, float(dt),
                                                    pay.ctypes.data_as(C.POINTER(C.c_float)))
        price, total = capi.cpu_mc_f32(capi.make_option(), n, steps, seed)
        want_total = np.float32(0.0)
        for p_ in pay:
            want_total = np.float32(want_total + p_)
        assert np.float32(total) == want_total and total > 0
        assert np.float32(total) / np.float32(n) == np.float32(mean)
        assert np.float32(price) == np.float32(np.exp(np.float32(-0.1)) * want_total / np.float32(n)) or \
            abs(price - math.exp(-0.1) * float(want_total) / n) < 1e-5 * price
    # window: a bullet job against a plain-python restatement on the same stream
    opt = capi.make_option(B=101.0, P1=2, P2=7, use_window=1)
    n, steps, seed = 50, 10, 5
    normals = np.zeros(n * steps, dtype=np.float32)
    ref.ref_mt19937_normals(seed, n * steps, normals.ctypes.data_as(C.POINTER(C.c_float)))
    f = np.float32
    dt = f(1.0) / f(steps)
    drift, vol = (f(0.1) - (f(0.2) * f(0.2)) / f(2)) * dt, f(0.2) * np.sqrt(dt, dtype=np.float32)
    total = f(0.0)
    for i in range(n):
        st, cnt = f(100.0), 0
        for j in range(steps):
            st = f(st * np.exp(f(drift + f(vol * normals[i * steps + j])), dtype=np.float32))
            cnt += 1 if st < f(101.0) else 0
        if 2 <= cnt <= 7:
            total = f(total + max(f(st - f(100.0)), f(0.0)))
    price, got = capi.cpu_mc_f32(opt, n, steps, seed)
    assert abs(got - float(total)) <= 2e-5 * max(1.0, float(total))      # numpy's expf vs libm's: an ulp here and there
    # unseeded = std::random_device, like the reference: two runs differ, both near the closed form
    a, _ = capi.cpu_mc_f32(capi.make_option(), 200_000, 1, 0, True)
    b, _ = capi.cpu_mc_f32(capi.make_option(), 200_000, 1, 0, True)
    assert a != b and abs(a - 13.2697) < 0.2 and abs(b - 13.2697) < 0.2
    assert capi.cpu_mc_f32(capi.make_option(), 0, 1, 1)[0] == 0.0
