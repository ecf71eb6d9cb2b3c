"""The C++ shim (include/*.hpp) re-exposes the reference's call surface over the C ABI.
CPU: the two re-created drivers compile with plain g++ against libmcamd.so.
GPU: they run; hello prints every label hello.cu prints with sane values, testing exits 0
(its own GPU-vs-CPU checks) and writes the reference's CSV format."""
import importlib
import math
import os
import re
import subprocess

import pytest

from conftest import ROOT

EX = os.path.join(ROOT, "examples")
pkg = importlib.import_module("monte-carlo-project-cuda_amd")


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(pkg.capi.LIB_PATH):
        pkg.build()
    subprocess.check_call(["make", "-C", EX, "-s", "all"])
    return EX


def test_shim_drivers_compile_with_gxx(built):
    assert os.path.exists(os.path.join(built, "hello")) and os.path.exists(os.path.join(built, "testing"))


def test_shim_headers_cite_reference_lines():
    for h in ("tool.hpp", "wrappers.hpp", "testing.hpp", "BlackandScholes.hpp", "monte_carlo.hpp", "option_price.hpp"):
        text = open(os.path.join(ROOT, "include", h)).read()
        assert re.search(r"inc/\w+\.(cuh|hpp)", text) and re.search(r":\d+-\d+", text), h


def test_every_shim_header_is_self_contained(tmp_path):
    # each public header must compile on its own (plain g++, C and C++ where it applies)
    inc = os.path.join(ROOT, "include")
    for h in sorted(os.listdir(inc)):
        src = tmp_path / ("use_" + h.replace(".", "_") + (".c" if h.endswith(".h") else ".cpp"))
        src.write_text(f'#include "{h}"\nint main(void) {{ return 0; }}\n')
        cc = ["gcc", "-std=c11"] if h.endswith(".h") else ["g++", "-std=c++17"]
        subprocess.check_call(cc + ["-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I" + inc, str(src)])


@pytest.mark.gpu
def test_hello_runs_and_prices_are_sane(built):
    out = subprocess.run([os.path.join(built, "hello")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    vals = {}
    for label in ("Average CPU Vanilla Option", "Monte Carlo CPU Bullet Option Price", "Average GPU",
                  "Average GPU bullet option", "Average GPU bullet option atomic",
                  "Average GPU bullet option nmc one point per block", "Average GPU bullet option nmc one kernel",
                  "Average GPU bullet option nmc optimal", "call Black Scholes", "seeded CPU vanilla",
                  "seeded CPU bullet"):
        m = re.search(re.escape(label) + r"\s*:\s*([-0-9.e+]+)", out.stdout)
        assert m, (label, out.stdout)
        vals[label] = float(m.group(1))
    se = 16.109 / math.sqrt(100000)
    assert abs(vals["call Black Scholes"] - 13.2697) < 1e-3
    assert abs(vals["Average GPU"] - 13.2697) < 4 * se
    assert abs(vals["Average CPU Vanilla Option"] - 13.2697) < 5 * se
    assert abs(vals["Average GPU bullet option"] - vals["Monte Carlo CPU Bullet Option Price"]) < 0.25
    assert vals["Average GPU bullet option"] == vals["Average GPU bullet option atomic"]
    a, b, c = (vals[k] for k in ("Average GPU bullet option nmc one point per block",
                                 "Average GPU bullet option nmc one kernel", "Average GPU bullet option nmc optimal"))
    assert a == b and abs(a - c) < 1e-3 * max(1.0, abs(a)) and a > 0
    assert "fp64 vanilla" in out.stdout
    # the seeded CPU pricers are repeatable (mcamd_cpu_mc_f32 with std::mt19937(1234)), so GPU-vs-CPU agreement is a
    # fixed-tolerance check: two independent 100 000-path estimates of the same price differ by N(0, 2 SE^2)
    assert abs(vals["Average GPU"] - vals["seeded CPU vanilla"]) < 4 * math.sqrt(2) * se
    se_bullet = 3.0 / math.sqrt(100000)     # bullet payoff: mostly zero, standard deviation ~3
    assert abs(vals["Average GPU bullet option"] - vals["seeded CPU bullet"]) < 4 * math.sqrt(2) * se_bullet + 1e-3
    again = subprocess.run([os.path.join(built, "hello")], capture_output=True, text=True, timeout=600)
    m2 = re.search(r"seeded CPU bullet\s*:\s*([-0-9.e+]+)", again.stdout)
    assert m2 and float(m2.group(1)) == vals["seeded CPU bullet"]          # same seed, same price


@pytest.mark.gpu
def test_testing_driver_passes_and_writes_csv(built, tmp_path):
    out = subprocess.run([os.path.join(built, "testing")], capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "FAIL" not in out.stdout
    rows = open(tmp_path / "testing.csv").read().strip().split("\n")
    assert rows[0] == "time,trajectory,value"
    assert len(rows) == 1 + 20 * 151            # t=0 row + 150 steps for each of 20 trajectories
    assert rows[1].split(",")[0] == "0" and float(rows[1].split(",")[2]) == 100.0


@pytest.mark.gpu
def test_multi_gpu_cpp_host_runs_on_the_visible_devices(built):
    out = subprocess.run([os.path.join(built, "multi_gpu"), "4000000", "16", "0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    m = re.search(r"devices (\d+) .* price ([0-9.]+) \+- ([0-9.]+) .*closed form ([0-9.]+)", out.stdout)
    assert m, out.stdout
    assert int(m.group(1)) >= 1 and abs(float(m.group(2)) - float(m.group(4))) < 4.5 * float(m.group(3))
    # the nested-MC host on the same group: per-device buffers, mcamd_group_nmc_fused; the mean of all point prices must be
    # the single-context call's (same seeds, same global path ids)
    n = re.search(r"nested MC on (\d+) device\(s\): (\d+) points x 200 inner paths, mean point price ([0-9.]+), lane efficiency ([0-9.]+)", out.stdout)
    assert n, out.stdout
    assert int(n.group(2)) == 4096 * 100 and 0.3 < float(n.group(4)) <= 1.0
    import torch
    capi = pkg.capi
    with capi.Context(0) as ctx:
        opt = capi.make_option(B=120.0, P1=10, P2=50, use_window=1)
        t, c, o = (torch.empty(4096 * 100, dtype=d, device="cuda") for d in (torch.float64, torch.int32, torch.float64))
        torch.cuda.synchronize()
        want = ctx.nmc_fused(opt, capi.make_sim(4096, 100, capi.F64, seed=1235, n_paths_inner=200), 1234, t, c, o)
    assert math.isclose(float(n.group(3)), want.price, rel_tol=1e-10)
