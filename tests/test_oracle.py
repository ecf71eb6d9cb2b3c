"""Pins the CPU oracle (oracle/oracle.c) against every golden vector available for the hot
path: the reference's closed form and array-driven pricer (fixtures generated from
oracle/_ref, i.e. the reference's own code compiled here), rocRAND's Philox known answers,
and — when oracle/_ref is present — the reference libraries themselves on fresh inputs.
CPU only."""
import ctypes as C
import math

import numpy as np
import pytest


# ---------------- closed form: inc/BlackandScholes.hpp ----------------
def test_closed_form_matches_reference_golden_bitwise(oracle, golden):
    g = golden("bs_closed_form.json")
    for c in g["call"]:
        got = oracle.bs_call_f32(c["S0"], c["K"], c["T"], c["r"], c["sigma"])
        assert np.float32(got) == np.float32(c["ref_call_f32"]), c
    for c in g["cnd"]:
        assert np.float32(oracle.cnd_f32(c["x"])) == np.float32(c["ref_cnd_f32"]), c


def test_closed_form_survey_values(oracle):
    # reference fp32 output for the benchmark option and friends
    want = {(100, 100, 1, .1, .2): 13.2696915, (100, 110, 1, .1, .2): 8.18306637, (100, 90, 1, .1, .2): 19.9885712,
            (100, 100, .5, .05, .3): 9.63486862, (100, 100, 2, .02, .4): 23.8472633, (50, 60, 1, .03, .25): 2.23166084}
    for k, v in want.items():
        assert np.float32(oracle.bs_call_f32(*k)) == np.float32(v)
    for x, v in ((0.6, 0.72574693), (0.4, 0.655421615), (-1.0, 0.158655271), (0.0, 0.49999994)):
        assert np.float32(oracle.cnd_f32(x)) == np.float32(v)


def test_closed_form_f64_vs_f32_within_published_error(oracle, golden):
    # the reference's A&S polynomial is good to ~1e-4 absolute on these ranges (|err| <= 2.3e-5 on the 6 survey cases)
    for c in golden("bs_closed_form.json")["call"]:
        exact = oracle.bs_call_f64(c["S0"], c["K"], c["T"], c["r"], c["sigma"])
        assert abs(exact - c["ref_call_f32"]) < 2e-4 * max(1.0, c["S0"] / 100)
    assert abs(oracle.bs_call_f64(100, 100, 1, 0.1, 0.2) - 13.269676584660893) < 1e-11


def test_closed_form_against_reference_library_fresh_inputs(oracle):
    L = oracle.ref_bs()
    if L is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    rng = np.random.default_rng(7)
    for _ in range(2000):
        S0 = np.float32(rng.uniform(10, 300))
        args = (float(S0), float(np.float32(S0 * rng.uniform(0.4, 2.0))), float(np.float32(rng.uniform(0.02, 5))),
                float(np.float32(rng.uniform(0, 0.2))), float(np.float32(rng.uniform(0.03, 1.0))))
        assert np.float32(oracle.bs_call_f32(*args)) == np.float32(L.ref_black_scholes_CPU(*args)), args


# ---------------- array-driven pricer: inc/testing.cuh:75-91 ----------------
def test_array_driven_matches_reference_golden_bitwise(oracle, golden):
    for c in golden("array_driven.json")["cases"]:
        z = np.array(c["normals"], dtype=np.float32)
        mean, pay = oracle.price_from_normals(z, c["n_paths"], c["n_steps"], c["S0"], c["sigma"], c["r"], c["K"], c["T"])
        assert np.array_equal(pay, np.array(c["ref_payoffs"], dtype=np.float32)), c["mt19937_seed"]
        assert np.float32(mean) == np.float32(c["ref_mean_undiscounted"])


def test_array_driven_survey_vector(oracle, golden):
    c = golden("array_driven.json")["cases"][0]
    want = [10.533936, 36.184418, 4.766663, 3.504868, 0, 13.813805, 0, 12.642952]
    assert np.allclose(c["ref_payoffs"], want, rtol=0, atol=5e-6)
    assert abs(c["ref_mean_undiscounted"] - 10.180830) < 5e-6


def test_array_driven_against_reference_library_fresh_inputs(oracle):
    L = oracle.ref_cpumc()
    if L is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    assert L.ref_sizeof_OptionData() == 48  # inc/tool.cuh:13-26: 12 x 4 bytes
    pf = C.POINTER(C.c_float)
    for seed, n_paths, n_steps in ((1, 64, 1), (2, 33, 100), (3, 7, 252)):
        z = np.zeros(n_paths * n_steps, dtype=np.float32)
        L.ref_mt19937_normals(seed, z.size, z.ctypes.data_as(pf))
        dt = np.float32(1.0) / np.float32(n_steps)
        sq = np.sqrt(dt, dtype=np.float32)
        pay = np.zeros(n_paths, dtype=np.float32)
        m = L.ref_simulateOptionPriceCPU_array(n_paths, n_steps, z.ctypes.data_as(pf), 100.0, 0.2, float(sq), 0.1, 100.0,
                                               float(dt), pay.ctypes.data_as(pf))
        m2, pay2 = oracle.price_from_normals(z, n_paths, n_steps, 100.0, 0.2, 0.1, 100.0, 1.0)
        assert np.array_equal(pay, pay2)
        assert np.float32(m) == np.float32(m2)


def test_array_driven_empty_input(oracle):
    m, pay = oracle.price_from_normals(np.zeros(0, dtype=np.float32), 0, 4, 100.0, 0.2, 0.1, 100.0, 1.0)
    assert m == 0.0 and pay.size == 0


# ---------------- RNG: Philox4x32-10 + rocRAND Box-Muller ----------------
def test_philox_known_answers(oracle, golden):
    for c in golden("rocrand_philox_kat.json")["cases"]:
        seed, sub = int(c["seed"]), int(c["subsequence"])
        raw = np.concatenate([oracle.philox(seed, sub, b) for b in range(3)])
        assert np.array_equal(raw, np.array(c["raw"], dtype=np.uint32)), (seed, sub)
        n4 = np.concatenate([oracle.normal4_f32(seed, sub, b) for b in range(2)])
        # libm sinf/cosf/logf vs rocRAND's host build: a few fp32 ulp
        assert np.allclose(n4, np.array(c["normal4"], dtype=np.float32), rtol=2e-6, atol=2e-7), (seed, sub)
        d2 = np.concatenate([oracle.normal2_f64(seed, sub, b) for b in range(2)])
        assert np.allclose(d2, np.array(c["normal_double2"]), rtol=1e-14, atol=1e-15), (seed, sub)


def test_philox_survey_words(oracle):
    assert [hex(x) for x in oracle.philox(1234, 0, 0)] == ['0x2090b348', '0xda7cf0ab', '0x4401906f', '0xcbca470e']
    assert [hex(x) for x in oracle.philox(1234, 1, 0)] == ['0xd115a128', '0x52fc7c75', '0xc7f33f17', '0xf1539db']


def test_philox_random123_kat(oracle):
    # Random123 kat_vectors: philox4x32-10, counter = key = 0 and all-ones
    assert [hex(x) for x in oracle.philox(0, 0, 0)] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    ones = 0xffffffffffffffff
    assert [hex(x) for x in oracle.philox(ones, ones, ones)] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']


def test_bulk_normals_moments(oracle):
    z = oracle.generate_normals(1234, 400001, 32)
    assert z.size == 400001 and abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3
    z64 = oracle.generate_normals(1234, 100001, 64)
    assert abs(z64.mean()) < 1e-2 and abs(z64.std() - 1) < 1e-2
    assert np.array_equal(z[:4], oracle.normal4_f32(1234, 0, 0))
    assert np.array_equal(z64[2:4], oracle.normal2_f64(1234, 0, 1))


# ---------------- RNG-driven MC ----------------
BENCH = dict(S0=100.0, T=1.0, K=100.0, r=0.1, v=0.2)


@pytest.mark.parametrize("precision", [32, 64])
@pytest.mark.parametrize("n_steps", [1, 12])
def test_mc_european_within_3se_of_closed_form(oracle, precision, n_steps):
    n = 200_000
    p = oracle.make_params(**BENCH, n_paths=n, n_steps=n_steps, seed=1234)
    res = oracle.mc_paths(p, precision, 0, n, threads=oracle.max_threads())
    fin = oracle.finalize(res["sum"], res["sumsq"], n, p.r, p.T)
    bs = oracle.bs_call_f64(100, 100, 1, 0.1, 0.2)
    assert abs(fin["price"] - bs) <= 3 * fin["std_err"]
    assert abs(fin["std_err"] - 16.109 / math.sqrt(n)) < 0.02 * 16.109 / math.sqrt(n) * 3
    assert fin["ci_lo"] < fin["price"] < fin["ci_hi"]


def test_mc_sharding_is_exact_partition(oracle):
    # union of shards == whole job, any split: payoffs identical, sums equal to rounding
    n = 5000
    p = oracle.make_params(**BENCH, n_paths=n, n_steps=8, seed=42)
    whole = oracle.mc_paths(p, 64, 0, n, want_payoffs=True)
    a = oracle.mc_paths(p, 64, 0, 1234, want_payoffs=True)
    b = oracle.mc_paths(p, 64, 1234, n - 1234, want_payoffs=True)
    assert np.array_equal(np.concatenate([a["payoffs"], b["payoffs"]]), whole["payoffs"])
    assert math.isclose(a["sum"] + b["sum"], whole["sum"], rel_tol=1e-13)
    assert math.isclose(a["sumsq"] + b["sumsq"], whole["sumsq"], rel_tol=1e-13)


def test_mc_threads_do_not_change_payoffs(oracle):
    p = oracle.make_params(**BENCH, n_paths=3000, n_steps=5, seed=9)
    a = oracle.mc_paths(p, 32, 0, 3000, want_payoffs=True, threads=1)
    b = oracle.mc_paths(p, 32, 0, 3000, want_payoffs=True, threads=4)
    assert np.array_equal(a["payoffs"], b["payoffs"]) and math.isclose(a["sum"], b["sum"], rel_tol=1e-12)


def test_mc_trajectory_terminal_equals_payoff(oracle):
    p = oracle.make_params(**BENCH, n_paths=257, n_steps=10, seed=5)
    res = oracle.mc_paths(p, 64, 0, 257, want_payoffs=True, want_traj=True)
    assert res["traj"].shape == (10, 257)
    assert np.array_equal(np.maximum(res["traj"][-1] - 100.0, 0.0), res["payoffs"])
    # the stream feeding path i is rocRAND subsequence i: first step reproduces normal2(block 0)[0]
    z0 = oracle.normal2_f64(5, 3, 0)[0]
    dt = 0.1
    want = 100.0 * math.exp((0.1 - 0.02) * dt + 0.2 * math.sqrt(dt) * z0)
    assert math.isclose(res["traj"][0, 3], want, rel_tol=1e-15)


def test_mc_bullet_window_and_counts(oracle):
    # hello.cu:11-13 barrier parameters; counts are running totals of (B > St)
    p = oracle.make_params(**BENCH, B=120.0, P1=10, P2=50, n_paths=2000, n_steps=100, seed=1234, use_window=1)
    res = oracle.mc_paths(p, 32, 0, 2000, want_payoffs=True, want_traj=True, want_counts=True)
    cnt = np.cumsum(res["traj"] < 120.0, axis=0).astype(np.int32)
    assert np.array_equal(cnt, res["counts"])
    ok = (cnt[-1] >= 10) & (cnt[-1] <= 50)
    assert np.array_equal(res["payoffs"], np.where(ok, np.maximum(res["traj"][-1] - 100.0, 0), 0.0))
    fin = oracle.finalize(res["sum"], res["sumsq"], 2000, p.r, p.T)
    # reference CPU bullet price at these parameters is 4.839 at 1M paths; loose statistical check
    assert abs(fin["price"] - 4.839) < 4 * fin["std_err"] + 0.05


def test_mc_bullet_statistical_agreement_with_reference_cpu(oracle):
    L = oracle.ref_cpumc()
    if L is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    n = 200_000
    ref = L.ref_simulateBulletOptionPriceCPU(100.0, 1.0, 100.0, 0.1, 0.2, 120.0, 10, 50, n, 100)
    p = oracle.make_params(**BENCH, B=120.0, P1=10, P2=50, n_paths=n, n_steps=100, seed=77, use_window=1)
    res = oracle.mc_paths(p, 32, 0, n, threads=oracle.max_threads())
    fin = oracle.finalize(res["sum"], res["sumsq"], n, p.r, p.T)
    assert abs(fin["price"] - ref) < 5 * math.sqrt(2) * fin["std_err"]
    ref1 = L.ref_simulateOptionPriceCPU(100.0, 1.0, 100.0, 0.1, 0.2, n)
    assert abs(ref1 - oracle.bs_call_f64(100, 100, 1, 0.1, 0.2)) < 5 * 16.109 / math.sqrt(n)


def test_mc_restart_triple(oracle):
    # (Ik, Sk, Tk): inc/trajectories.cuh:116-117,140-143
    p = oracle.make_params(**BENCH, B=120.0, P1=0, P2=100, n_paths=100, n_steps=20, seed=3, use_window=1, Ik=4,
                           Sk=90.0, Tk=15)
    res = oracle.mc_paths(p, 64, 0, 100, want_traj=True, want_counts=True)
    assert res["traj"].shape == (5, 100)
    assert (res["counts"][0] >= 4).all() and (res["counts"][-1] <= 9).all()


def test_nmc_point_matches_mc_paths(oracle):
    # a point's inner price == windowed continuation from (St, count) with its own substreams
    p = oracle.make_params(**BENCH, B=120.0, P1=2, P2=30, n_paths=4, n_steps=16, n_paths_inner=64, seed=1235,
                           use_window=1)
    point, step, St, cnt = 37, 5, 104.5, 3
    got = oracle.nmc_point(p, 64, point, step, St, cnt)
    q = oracle.make_params(**BENCH, B=120.0, P1=2, P2=30, n_paths=64, n_steps=16, seed=1235, use_window=1, Ik=cnt,
                           Sk=St, Tk=step + 1)
    res = oracle.mc_paths(q, 64, point * 64, 64)
    assert math.isclose(got, res["sum"] * math.exp(-0.1) / 64, rel_tol=1e-13)
    assert oracle.nmc_point(p, 64, point, step, St, 31) == 0.0  # count already > P2: inc/nmc.cuh:53


def test_finalize(oracle):
    x = np.array([0.0, 1.0, 4.0, 10.0])
    fin = oracle.finalize(x.sum(), (x * x).sum(), 4, 0.1, 2.0)
    d = math.exp(-0.2)
    assert math.isclose(fin["price"], d * x.mean(), rel_tol=1e-15)
    assert math.isclose(fin["std_err"], d * x.std(ddof=1) / 2.0, rel_tol=1e-14)
    z = oracle.finalize(0.0, 0.0, 0, 0.1, 1.0)
    assert z["price"] == 0.0 and z["std_err"] == 0.0
