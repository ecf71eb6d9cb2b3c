#!/usr/bin/env python3
"""Regenerates the golden fixtures in tests/golden/.  Run in the build container only
(needs /root/reference for oracle/_ref and hipcc + rocRAND headers for the RNG KATs):

    python tests/golden/make_golden.py

Fixtures are DATA (inputs + expected outputs):
  bs_closed_form.json       outputs of the reference's own CND / black_scholes_CPU
                            (inc/BlackandScholes.hpp compiled as oracle/_ref/libref_bs.so)
  array_driven.json         std::mt19937 + std::normal_distribution<float> normals (libstdc++ 11)
                            and the payoffs the reference's array-driven CPU pricer
                            (inc/testing.cuh:75-91, oracle/_ref/libref_cpumc.so) returns on them
  rocrand_philox_kat.json   rocRAND 7.2 Philox4x32-10 raw words / normal4 / normal_double2,
                            executed on the host (tests/golden/gen_rocrand_kat.cpp)
"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyoracle as o  # noqa: E402


def f32(x):
    return float(np.float32(x))


def closed_form():
    L = o.ref_bs()
    assert L is not None, "oracle/_ref/libref_bs.so missing: run oracle/build_ref.sh"
    cases = [(100, 100, 1, .1, .2), (100, 110, 1, .1, .2), (100, 90, 1, .1, .2),
             (100, 100, .5, .05, .3), (100, 100, 2, .02, .4), (50, 60, 1, .03, .25)]
    rng = np.random.default_rng(20261004)
    for _ in range(250):
        S0 = rng.uniform(20, 200)
        cases.append((S0, S0 * rng.uniform(0.5, 1.6), rng.uniform(0.05, 3.0), rng.uniform(0.0, 0.12),
                      rng.uniform(0.05, 0.8)))
    out = []
    for c in cases:
        c32 = tuple(f32(x) for x in c)
        out.append({"S0": c32[0], "K": c32[1], "T": c32[2], "r": c32[3], "sigma": c32[4],
                    "ref_call_f32": float(L.ref_black_scholes_CPU(*c32))})
    xs = [0.6, 0.4, -1.0, 0.0] + [f32(x) for x in np.linspace(-6, 6, 97)]
    cnd = [{"x": f32(x), "ref_cnd_f32": float(L.ref_CND(f32(x)))} for x in xs]
    return {"source": "reference inc/BlackandScholes.hpp compiled standalone with g++ (oracle/build_ref.sh); "
                      "inputs are float32 values",
            "call": out, "cnd": cnd}


def array_driven():
    L = o.ref_cpumc()
    assert L is not None, "oracle/_ref/libref_cpumc.so missing: run oracle/build_ref.sh"
    pf = C.POINTER(C.c_float)
    cases = []
    for (seed, n_paths, n_steps, S0, K, T, r, sigma) in [
            (1234, 8, 4, 100.0, 100.0, 1.0, 0.1, 0.2),        # SURVEY.md 8c vector
            (555, 20, 150, 100.0, 100.0, 1.0, 0.1, 0.2),      # testing.cu:104-108 shape
            (99, 5, 252, 100.0, 110.0, 2.0, 0.03, 0.35),
            (7, 3, 1, 50.0, 45.0, 0.5, 0.05, 0.25)]:
        z = np.zeros(n_paths * n_steps, dtype=np.float32)
        L.ref_mt19937_normals(seed, z.size, z.ctypes.data_as(pf))
        dt = np.float32(T) / np.float32(n_steps)
        sq = np.sqrt(dt, dtype=np.float32)
        pay = np.zeros(n_paths, dtype=np.float32)
        mean = L.ref_simulateOptionPriceCPU_array(n_paths, n_steps, z.ctypes.data_as(pf), S0, sigma, float(sq), r,
                                                  K, float(dt), pay.ctypes.data_as(pf))
        cases.append({"mt19937_seed": seed, "n_paths": n_paths, "n_steps": n_steps, "S0": S0, "K": K, "T": T,
                      "r": r, "sigma": sigma, "dt": float(dt), "sqrdt": float(sq),
                      "normals": [float(x) for x in z], "ref_payoffs": [float(x) for x in pay],
                      "ref_mean_undiscounted": float(mean)})
    return {"source": "reference inc/testing.cuh:75-91 via oracle/_ref/libref_cpumc.so; normals[path*n_steps+step]",
            "cases": cases}


def rocrand_kat():
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "gen_kat")
        subprocess.check_call(["hipcc", "-O2", "-w", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                               os.path.join(HERE, "gen_rocrand_kat.cpp"), "-o", exe, "-L/opt/rocm/lib",
                               "-lamdhip64"])
        return json.loads(subprocess.check_output([exe]))


def main():
    o.build_ref()
    for name, fn in (("bs_closed_form.json", closed_form), ("array_driven.json", array_driven),
                     ("rocrand_philox_kat.json", rocrand_kat)):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(fn(), f, indent=1)
        print("wrote", name)


if __name__ == "__main__":
    main()
