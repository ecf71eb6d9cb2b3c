#!/usr/bin/env python3
"""Benchmark driver.  `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line.

A "step" is one pass of the hot path over one batch: every rank prices its shard of a European
call (S0=K=100, T=1, r=0.1, sigma=0.2 — hello.cu:6-10) through the C ABI (mcamd_price_paths_enqueue):
Philox RNG -> 252 GBM steps -> payoff -> fp64 (sum, sumsq), nothing stored — BASELINE.json
configs[1] (10M paths, 252 steps, fp64, in-register) per GPU.  With N > 1 the global job is N x 10M
paths (or --global-paths, e.g. 1e9 = configs[4]) sharded by contiguous global path id, and each step ends
with ONE all-reduce of the 6-double statistics record over RCCL, enqueued behind the kernel on the same
stream: no host round trip per step.  Inputs are a handful of scalars, so nothing crosses PCIe in the
timed region.  A bare `--gpus N` (no WORLD_SIZE in the environment) starts the N ranks itself.

Extra objects on the line (N = 1 only, all measured in this run):
  device          name / arch / CUs / clock / HBM total and free (mcamd_get_device_info; inc/tool.cuh:56-88,176-188)
  roofline        the dominant kernel against the VALU issue roofline (this path has ~zero HBM traffic and no
                  matrix work): W issue slots per path-step (counted from the ISA of the library that is loaded;
                  refused when the count was taken from other sources) x measured path-steps/s
  sweep           the north-star path counts 1M / 10M / 100M x {fp64, fp32}, in-register, each with kernel time,
                  whole-call time and roofline fraction
  roofline_store  BASELINE configs[2] (100M paths x 252 steps fp32 stored step-major + the payoff vector,
                  101.2 GB) against the HBM roofline — the bandwidth-bound path
  accuracy_252    "price within 1e-4 of closed form" on a 252-step BASELINE shape: 1e9 antithetic pairs with the
                  S_T control variate, fp64
  cfg1            BASELINE configs[0] on this host: closed form (1M evaluations) and the reference's serial MC
  cpu_baseline    the reference's own CPU Monte Carlo (oracle/_ref) or the oracle port, timed on this host
Other workloads: --workload store | nmc | vanilla1 | european252_f32 (see --help).
"""
from __future__ import annotations

import argparse
import importlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BS_EXACT = 13.269676584660893  # closed form, fp64, benchmark option
OPTION = dict(S0=100.0, T=1.0, K=100.0, r=0.1, v=0.2)
BULLET = dict(B=120.0, P1=10, P2=50, use_window=1)     # hello.cu:11-13

# VALU issue roofline (MI355X_MICROARCH.md): 256 CUs x 4 SIMD-32 x 2.4 GHz; a wave64 VALU
# instruction issues in 2 cycles at full rate -> 32 lane-ops / cycle / SIMD.
PEAK_VALU_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12   # 78.6 Tlane-op/s (= 157.3 TFLOP/s fp32 FMA / 2)
PEAK_HBM_GBS = 8000.0

# Fixed yardsticks for the VALU-bound loops: the operation-count minimum of the scheme itself (DESIGN.md section 5 derives
# them), in the same 2-cycle issue slots as W.  Unlike W — which is counted from whatever loop ships and falls when an
# instruction is removed — these do not move with the implementation, so paths/s x W_FLOOR / peak only rises when the
# kernel gets faster.  Window-less European path-step: Philox4x32-10 (17 multiplies + 17 three-input xors per block that
# no hoisting removes) + the uniforms + -2 ln u + sqrt + ONE sine per Box-Muller pair + one accumulate.
# Window loop of the nested-MC inner stage (fp64): the same Philox and uniforms, -2 ln u, the volatility-scaled radius,
# sine AND cosine (each step needs its own exponent), and per step one add, one compare and one count update.
W_FLOOR = {"price_f64": 62.5, "price_f32": 22.25, "nmc_wave_f64_window": 72.5}

METRIC = "MC paths/sec, European call (price error vs closed-form BS reported)"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="european252",
                    choices=["european252", "european252_f32", "vanilla1", "store", "nmc"])
    ap.add_argument("--paths", type=int, default=0, help="paths per GPU per step (default: the config's)")
    ap.add_argument("--global-paths", type=int, default=0,
                    help="strong scaling: total paths per step, split over the ranks by contiguous path id "
                         "(e.g. 1000000000 for BASELINE configs[4])")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1: nccl (= RCCL over xGMI, the real path) or gloo (rehearsal of the "
                         "multi-rank control flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--nmc-window", default="bullet", choices=["bullet", "european"],
                    help="nested MC: the reference's bullet window B=120, P1=10, P2=50 (hello.cu:11-13), or the "
                         "European-window variant (B=0, P2=N_STEPS) whose work count is deterministic")
    ap.add_argument("--nmc-strategy", default="wave", choices=["wave", "block", "fused"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-store-roofline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--store-leg-first", action="store_true",
                    help="run the configs[2] store leg right after the headline instead of after the sweep (diagnostic: "
                         "does its time depend on what ran before it?)")
    ap.add_argument("--no-accuracy", action="store_true")
    ap.add_argument("--no-nmc", action="store_true", help="skip the BASELINE configs[3] nested-MC side leg")
    ap.add_argument("--accuracy-pairs", type=int, default=1_000_000_000)
    ap.add_argument("--cpu-sample-paths", type=int, default=0)
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="launcher test only: spawn the ranks, rendezvous, shard and all-reduce the path counts, price "
                         "nothing (needs no GPU); the line carries value null and \"rehearsal\": true")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher: bare `bench.py --gpus N`
# ------------------------------------------------------------------------------------------------
def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(argv, n_ranks: int, port: int):
    """The child command a bare `bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) runs: the driver's own
    multi-rank form, one fresh process per GPU.  `argv` is this process's argument list, passed through unchanged."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def self_launch(args, argv) -> int:
    """Bare `python bench.py --gpus N`: start N rank processes and relay rank 0's JSON line.  Runs BEFORE anything
    in this process touches the GPU (torch.cuda.device_count() does not initialise it on this image), and starts
    the ranks as children — the parent never execs and never makes a HIP call."""
    import subprocess
    need_devices = args.backend == "nccl" and not args.rehearse_launch
    if need_devices:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs for the RCCL path, found {have} "
                  "(use --backend gloo to rehearse the multi-rank control flow on fewer GPUs)", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = launcher_command(argv, args.gpus, free_port())
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    for l in proc.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if proc.returncode != 0 or len(lines) != 1:
        print(f"bench.py: the {args.gpus}-rank child exited with {proc.returncode} and printed {len(lines)} JSON line(s)",
              file=sys.stderr)
        return proc.returncode or 1
    print(lines[0])
    return 0


def rehearse(args, world: int, rank: int):
    """--rehearse-launch: launcher + rendezvous + sharding + the one all-reduce, with path COUNTS in place of payoff
    sums.  No kernel runs and nothing is priced; it exists so that the multi-rank launch path has a test on a box
    with no GPU (tests/test_bench_contract.py)."""
    import torch
    import torch.distributed as dist
    pkg = importlib.import_module("monte-carlo-project-cuda_amd")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    n_total = args.global_paths or (args.paths or 10_000_000) * world
    lo, n_local = pkg.sharding.shard_range(n_total, world, rank) if args.global_paths else (rank * (n_total // world), n_total // world)
    rec = torch.tensor([float(n_local), float(lo), 1.0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(rec)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None,
                          "unit": "paths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "rehearsal": True, "ranks_seen": int(rec[2].item()), "paths_covered": int(rec[0].item()),
                          "config": {"workload": workload_name(args.workload, n_total // world, n_total, world,
                                                               bool(args.global_paths)),
                                     "global_paths": n_total}}))


def workload_name(wl: str, per_gpu: int, n_total: int, world: int, strong: bool, nmc_window: str = "bullet") -> str:
    """config.workload, from the numbers actually run (BASELINE config named only when the shape is that config's)."""
    def cnt(n):
        for div, suf in ((1_000_000_000, "B"), (1_000_000, "M"), (1_000, "k")):
            if n >= div and n % div == 0:
                return f"{n // div}{suf}"
        return str(n)
    if wl == "european252":
        if strong:
            tag = " (BASELINE configs[4])" if n_total == 1_000_000_000 else ""
            return f"European call, {cnt(n_total)} paths x 252 steps, fp64, in-register, path-sharded across {world} GPU(s){tag}"
        tag = " (BASELINE configs[1])" if per_gpu == 10_000_000 else ""
        return f"European call, {cnt(per_gpu)} paths/GPU x 252 steps, fp64, in-register{tag}"
    if wl == "european252_f32":
        return f"European call, {cnt(per_gpu)} paths/GPU x 252 steps, fp32, in-register"
    if wl == "vanilla1":
        return f"European call, {cnt(per_gpu)} paths/GPU x 1 exact step, fp64, in-register"
    if wl == "store":
        tag = " (BASELINE configs[2])" if per_gpu == 100_000_000 else ""
        return f"European call, {cnt(per_gpu)} paths/GPU x 252 steps, fp32, trajectories stored step-major{tag}"
    tag = " (BASELINE configs[3])" if per_gpu == 65_536 else ""
    win = "bullet window B=120 P1=10 P2=50 (hello.cu:11-13)" if nmc_window == "bullet" else "European window (B=0, P2=N_STEPS)"
    return f"nested MC, {cnt(per_gpu)} outer paths/GPU x 252 steps x 1000 inner, fp64, {win}{tag}"


# ------------------------------------------------------------------------------------------------
# roofline helpers
# ------------------------------------------------------------------------------------------------
def load_w_slots(build_id: str):
    """ISA issue-slot counts of the shipped inner loops (tools/count_valu_slots.py, refreshed by build()).  They are
    only valid for the sources they were counted from: a count taken from other sources than the loaded library is
    refused (every W is None and roofline.frac is null, with the two ids in roofline.stale)."""
    path = os.path.join(ROOT, "profiles", "valu_slots.json")
    if not os.path.exists(path):
        return {}, {"reason": "profiles/valu_slots.json missing"}
    with open(path) as f:
        d = json.load(f)
    if d.get("build_id") != build_id:
        return {}, {"reason": "ISA slot counts were taken from other sources than the loaded library",
                    "library_build_id": build_id, "counts_build_id": d.get("build_id")}
    return d, None


def pmc_traffic_bytes(kernel_key: str):
    """HBM bytes per launch of a kernel from the committed PMC digest (profiles/*_pmc_per_kernel.json, produced by
    tools/profile.sh: separate WRITE_SIZE and FETCH_SIZE passes over this same command).  WRITE_SIZE is in KB and
    exact for 16 B-per-lane stores; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128 B requests at
    64 B).  None when no digest has been committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_per_kernel.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        d = json.load(f)
    for k, v in d.items():
        if kernel_key in k and "WRITE_SIZE" in v and "FETCH_SIZE" in v:
            return (v["WRITE_SIZE"] + 2.0 * v["FETCH_SIZE"]) * 1024.0
    return None


def pmc_valu_busy(kernel_key: str, build_id: str):
    """Fraction of the elapsed cycles the vector ALUs spent issuing, for one kernel, from the committed PMC digest
    (SQ_ACTIVE_INST_VALU x 4 / SIMDs over GRBM_GUI_ACTIVE / 8 XCDs, as profiles/README.md derives it) — only when the
    digest was taken from the library that is loaded (it carries mcamd_build_id like the slot counts do)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_per_kernel.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        d = json.load(f)
    if d.get("build_id") != build_id:
        return {"value": None, "stale": {"digest": os.path.basename(files[-1]), "digest_build_id": d.get("build_id"),
                                         "library_build_id": build_id}}
    for k, v in d.items():
        if isinstance(v, dict) and kernel_key in k and v.get("GRBM_GUI_ACTIVE") and "SQ_ACTIVE_INST_VALU" in v:
            simds = 256 * 4
            return {"value": (v["SQ_ACTIVE_INST_VALU"] * 4 / simds) / (v["GRBM_GUI_ACTIVE"] / 8),
                    "digest": os.path.basename(files[-1]), "kernel": k}
    return None


def valu_roofline(W, stale, key, kernel, lane_steps_per_s, traffic_key=None, extra=None, build_id=None):
    w = W.get(key)
    rl = {"bound": "valu", "peak": PEAK_VALU_TLANEOPS, "unit": "Tlane-op/s",
          "traffic": pmc_traffic_bytes(traffic_key) if traffic_key else None, "kernel": kernel,
          "path_steps_per_s_kernel": lane_steps_per_s, "valu_slots_per_path_step": w,
          "achieved": lane_steps_per_s * w / 1e12 if w else None,
          "frac": lane_steps_per_s * w / 1e12 / PEAK_VALU_TLANEOPS if w else None}
    if key in W_FLOOR:   # the fixed yardstick: does not fall when an instruction is removed
        rl["valu_slots_floor"] = W_FLOOR[key]
        rl["frac_vs_floor"] = lane_steps_per_s * W_FLOOR[key] / 1e12 / PEAK_VALU_TLANEOPS
    if traffic_key and build_id:
        rl["valu_busy"] = pmc_valu_busy(traffic_key, build_id)
    det = W.get(key + "_detail")
    if w and det and det.get("measured_cost_cycles_per_iteration"):
        # the same instruction count priced with the per-instruction issue costs MEASURED on MI355X
        # (tools/ubench_valu.hip -> profiles/r01_valu_issue_costs.json; e.g. v_fma_f64 4.23 cycles, not 4): the
        # fraction of the issue rate the hardware actually sustains for this instruction mix
        ratio = det["measured_cost_cycles_per_iteration"] / det["issue_cycles_per_iteration"]
        rl["issue_model"] = {"nominal_cycles_per_iteration": det["issue_cycles_per_iteration"],
                             "measured_cost_cycles_per_iteration": det["measured_cost_cycles_per_iteration"],
                             "path_steps_per_iteration": det["path_steps_per_iteration"],
                             "frac_of_measured_issue_rate": rl["frac"] * ratio}
    if stale:
        rl["stale"] = stale
    if extra:
        rl.update(extra)
    return rl


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


# ------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only; the only place bench.py touches oracle/)
# ------------------------------------------------------------------------------------------------
def cpu_baseline(n_steps: int, sample_paths: int):
    """Times the CPU path on this host: reference build if present, else the oracle port."""
    from oracle import pyoracle as o
    out = {}
    ref = o.ref_cpumc()
    cores_avail = os.cpu_count() or 1
    if ref is not None:
        n = sample_paths or (2_000_000 if n_steps > 1 else 50_000_000)
        t0 = time.perf_counter()
        if n_steps > 1:
            # reference CPU multi-step pricer with the barrier window wide open = European call
            price = ref.ref_simulateBulletOptionPriceCPU(100.0, 1.0, 100.0, 0.1, 0.2, 0.0, 0, n_steps, n, n_steps)
        else:
            price = ref.ref_simulateOptionPriceCPU(100.0, 1.0, 100.0, 0.1, 0.2, n)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": n / dt, "unit": "paths/s", "cores": 1, "kind": "reference",
            "sample": f"{n} paths x {n_steps} steps, fp32, mt19937, reference inc/tool.cuh:"
                      f"{'133-173' if n_steps > 1 else '104-130'} compiled as oracle/_ref, {dt:.1f} s",
            "price": float(price), "host_cores_available": cores_avail}
    threads = o.max_threads()
    n = sample_paths or (threads * 150_000 if n_steps > 1 else 20_000_000)
    p = o.make_params(n_paths=n, n_steps=n_steps, seed=1234)
    t0 = time.perf_counter()
    res = o.mc_paths(p, 64, 0, n, threads=threads)
    dt = time.perf_counter() - t0
    fin = o.finalize(res["sum"], res["sumsq"], n, 0.1, 1.0)
    port = {"value": n / dt, "unit": "paths/s", "cores": threads, "kind": "port",
            "sample": f"{n} paths x {n_steps} steps, fp64, Philox (same stream as the GPU), OpenMP, {dt:.1f} s",
            "price": fin["price"], "host_cores_available": cores_avail}
    if "cpu_baseline" in out:
        out["cpu_baseline_port"] = port
    else:
        out["cpu_baseline"] = port
    return out


def cpu_cfg1(capi):
    """BASELINE configs[0] on this host: the closed form evaluated 1M times over a (K, sigma) grid
    (inc/BlackandScholes.hpp:34-43) and the 1M-path, 1-step serial Monte Carlo (inc/tool.cuh:104-130), one core."""
    from oracle import pyoracle as o
    out = {"workload": "BASELINE configs[0]: European call, 1M paths x 1 step + closed form, CPU, 1 core"}
    ref_bs, ref_mc = o.ref_bs(), o.ref_cpumc()
    n_eval = 1_000_000
    if ref_bs is not None:
        t0 = time.perf_counter()
        chk = ref_bs.ref_black_scholes_grid(n_eval) if hasattr(ref_bs, "ref_black_scholes_grid") else None
        dt = time.perf_counter() - t0
        if chk is not None:
            out["closed_form"] = {"kind": "reference", "evals": n_eval, "seconds": dt, "evals_per_s": n_eval / dt,
                                  "checksum": float(chk), "what": "black_scholes_CPU over a 1000 x 1000 (K, sigma) grid, fp32"}
    if "closed_form" not in out:
        t0 = time.perf_counter()
        chk = o.bs_call_f32_grid(n_eval)
        dt = time.perf_counter() - t0
        out["closed_form"] = {"kind": "port", "evals": n_eval, "seconds": dt, "evals_per_s": n_eval / dt,
                              "checksum": float(chk), "what": "oracle_bs_call_f32 over a 1000 x 1000 (K, sigma) grid, fp32"}
    out["closed_form"]["price_benchmark_option_f32"] = capi.bs_call_f32(100.0, 100.0, 1.0, 0.1, 0.2)
    n = 1_000_000
    if ref_mc is not None:
        t0 = time.perf_counter()
        price = ref_mc.ref_simulateOptionPriceCPU(100.0, 1.0, 100.0, 0.1, 0.2, n)
        dt = time.perf_counter() - t0
        out["serial_mc"] = {"kind": "reference", "paths": n, "steps": 1, "seconds": dt, "paths_per_s": n / dt,
                            "price": float(price), "abs_err_vs_bs": abs(float(price) - BS_EXACT),
                            "what": "simulateOptionPriceCPU (inc/tool.cuh:104-130), fp32, mt19937, unseeded, 1 core"}
    p = o.make_params(n_paths=n, n_steps=1, seed=1234)
    t0 = time.perf_counter()
    res = o.mc_paths(p, 64, 0, n, threads=1)
    dt = time.perf_counter() - t0
    fin = o.finalize(res["sum"], res["sumsq"], n, 0.1, 1.0)
    out["serial_mc_port_f64"] = {"kind": "port", "paths": n, "steps": 1, "seconds": dt, "paths_per_s": n / dt,
                                 "price": fin["price"], "std_err": fin["std_err"],
                                 "abs_err_vs_bs": abs(fin["price"] - BS_EXACT)}
    return out


def cpu_nmc_baseline(opt_kw, n_paths, n_steps, n_inner, traj, cnt, budget_s=12.0):
    """Nested MC on this host, one core: oracle_nmc_point (inc/nmc.cuh:47-66,100-103 — every remaining step of every
    inner path, as the reference runs it) on a sample of stored points spread over the steps."""
    from oracle import pyoracle as o
    p = o.make_params(**OPTION, B=opt_kw["B"], P1=opt_kw["P1"], P2=opt_kw["P2"], use_window=1, n_paths=n_paths,
                      n_steps=n_steps, n_paths_inner=n_inner, seed=1235)
    S, C = traj.view(n_steps, n_paths), cnt.view(n_steps, n_paths)
    steps = [0, 25, 50, 100, 150, 200, 240]
    t0 = time.perf_counter()
    done, path_steps = 0, 0
    q = 0
    while time.perf_counter() - t0 < budget_s and q < 64:
        for s_ in steps:
            c0 = int(C[s_, q].item())
            o.nmc_point(p, 64, q * n_steps + s_, s_, float(S[s_, q].item()), c0)
            done += 1
            if not (c0 > p.P2):
                path_steps += n_inner * (n_steps - 1 - s_)
        q += 1
    dt = time.perf_counter() - t0
    return {"value": done * n_inner / dt, "unit": "paths/s", "cores": 1, "kind": "port",
            "sample": f"{done} stored points x {n_inner} inner paths (steps {steps}, {q} outer paths), fp64, "
                      f"oracle_nmc_point = inc/nmc.cuh:47-66 restated, {dt:.1f} s",
            "inner_path_steps_per_s": path_steps / dt, "host_cores_available": os.cpu_count() or 1}


def store_leg(ctx, capi, torch, opt, position):
    """Bandwidth-bound path beside the headline (N = 1 only): BASELINE configs[2], 100M paths x 252 steps fp32 stored
    step-major with the payoff vector, 2 warm-up + 7 timed launches — and, alternating with them in the same process, the
    same launch shape and store stream with nothing simulated (mcamd_diag_store_pattern): the HBM write ceiling of this
    access pattern on THIS box at THIS moment, so the line tells a slow box from a slow kernel."""
    try:
        n3, s3 = 100_000_000, 252
        free, _ = torch.cuda.mem_get_info()
        if free <= n3 * s3 * 4 + (4 << 30):
            return {"error": "not enough free HBM for the 100.8 GB trajectory buffer"}
        buf = torch.empty(n3 * s3, dtype=torch.float32, device="cuda")
        pay = torch.empty(n3, dtype=torch.float32, device="cuda")
        sim3 = capi.make_sim(n3, s3, capi.F32, 1234)
        for _ in range(2):   # first touches of a fresh 100.8 GB allocation are slower
            ctx.simulate_trajectories(opt, sim3, buf, None, pay)
        ks, cs = [], []
        for i in range(7):
            if i % 2 == 0:
                cs.append(ctx.diag_store_pattern(n3, s3, capi.F32, buf, pay))
            r3 = ctx.simulate_trajectories(opt, sim3, buf, None, pay)
            ks.append(r3.kernel_ms)
        kms, cms = sum(ks) / len(ks), sum(cs) / len(cs)
        nbytes = n3 * s3 * 4 + n3 * 4 + 16 * r3.grid     # trajectories + payoff vector + block records: 101.2 GB
        ach = nbytes / (kms / 1e3) / 1e9
        out = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
               "traffic": pmc_traffic_bytes("store_kernel<float"), "kernel": "store_kernel<float,false,STEP_MAJOR,vec>",
               "kernel_ms": kms, "kernel_ms_min": min(ks), "launches": len(ks),
               "workload": "BASELINE configs[2]: 100M paths x 252 steps fp32 stored step-major + the payoff vector",
               "algorithmic_bytes_per_launch": nbytes, "paths_per_s": n3 / (kms / 1e3),
               "same_run_ceiling": {"what": "the same launch shape and store stream with nothing simulated "
                                            "(mcamd_diag_store_pattern), alternating with the timed launches",
                                    "kernel_ms": cms, "kernel_ms_min": min(cs), "launches": len(cs),
                                    "GB_per_s": (n3 * s3 * 4 + n3 * 4) / (cms / 1e3) / 1e9},
               "frac_of_same_run_store_ceiling": cms / kms, "leg_position": position,
               "price": r3.price, "std_err": r3.std_err, "abs_err_vs_bs": abs(r3.price - BS_EXACT)}
        del buf, pay
        return out
    except Exception as e:  # the headline must survive a failure of the side measurement
        return {"error": str(e)}


# ------------------------------------------------------------------------------------------------
def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, argv))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world   # launched by torch.distributed.run: the launcher's world size is authoritative
    if args.rehearse_launch:
        return rehearse(args, world, rank)

    import torch
    import torch.distributed as dist

    pkg = importlib.import_module("monte-carlo-project-cuda_amd")
    capi = pkg.capi
    W, stale = load_w_slots(capi.build_id())

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    device_index = local_rank % torch.cuda.device_count()   # == local_rank on a node with one GPU per rank
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group("gloo")
    coll_device = "cuda" if args.backend == "nccl" else "cpu"

    wl = args.workload
    prec = capi.F32 if wl in ("european252_f32", "store") else capi.F64
    n_steps = 1 if wl == "vanilla1" else 252
    default_paths = {"european252": 10_000_000, "european252_f32": 10_000_000, "vanilla1": 100_000_000,
                     "store": 100_000_000, "nmc": 65_536}
    sharding = pkg.sharding
    if args.global_paths:
        n_total = args.global_paths
        lo, per_gpu = sharding.shard_range(n_total, world, rank)   # strong scaling
    else:
        per_gpu = args.paths or default_paths[wl]                  # weak scaling: fixed work per GPU
        n_total = per_gpu * world
        lo = rank * per_gpu
    opt = capi.make_option(**OPTION)
    # One explicit stream for everything: the library launches on it, torch allocates / reduces on it (made the
    # current stream), so "enqueue kernel -> all-reduce its output" is ordered on the device with no host sync.
    # (torch's default stream has handle 0, which the C ABI reads as "create your own stream".)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx = capi.Context(device_index, stream.cuda_stream)
    info = ctx.device_info()
    device = {"name": info.name.decode(), "arch": info.arch.decode(), "compute_units": info.compute_units,
              "wavefront_size": info.wavefront_size, "clock_mhz": info.clock_khz / 1e3,
              "mem_clock_mhz": info.mem_clock_khz / 1e3, "mem_bus_bits": info.mem_bus_bits,
              "hbm_total_gb": info.total_mem / 1e9, "hbm_free_gb": info.free_mem / 1e9,
              "lds_per_block_kb": info.lds_per_block / 1024, "l2_mb": info.l2_bytes / 2**20,
              "devices_visible": info.device_count, "library_build_id": capi.build_id()}

    traj = payoffs = None
    nmc = None
    if wl == "store":
        traj = torch.empty(per_gpu * n_steps, dtype=torch.float32, device="cuda")
        payoffs = torch.empty(per_gpu, dtype=torch.float32, device="cuda")
    if wl == "nmc":
        n_inner = 1000
        win = dict(BULLET) if args.nmc_window == "bullet" else dict(B=0.0, P1=0, P2=n_steps, use_window=1)
        nmc = {"opt": capi.make_option(**OPTION, **win), "win": win, "n_inner": n_inner,
               "traj": torch.empty(per_gpu * n_steps, dtype=torch.float64, device="cuda"),
               "cnt": torch.empty(per_gpu * n_steps, dtype=torch.int32, device="cuda"),
               "out": torch.empty(per_gpu * n_steps, dtype=torch.float64, device="cuda"),
               "variant": {"wave": capi.NMC_WAVE_PER_POINT, "block": capi.NMC_BLOCK_PER_POINT, "fused": -1}[args.nmc_strategy]}

    is_price = wl in ("european252", "european252_f32", "vanilla1")
    # in-register workloads run asynchronously: each step enqueues the simulation + final reduce, then (N > 1) ONE
    # all-reduce of the 6-double stats record on the same stream; the host synchronises once, after the K steps.
    stats = torch.zeros(max(args.steps, 1) + args.warmup, 8, dtype=torch.float64, device="cuda")
    pending = []
    nmc_runs = []

    def one_step(i: int, slot: int = 0):
        seed = 1234 + i
        if is_price:
            ctx.price_paths_enqueue(opt, capi.make_sim(n_total, n_steps, prec, seed, lo, per_gpu), stats[slot])
            if world > 1:
                # the one collective of the path: the 6-double record over RCCL/xGMI.  async_op: the collective
                # waits for this step's kernels, but the NEXT step's kernels do not wait for the collective —
                # it runs on RCCL's own stream underneath them; everything is joined once after the K steps.
                pending.append(dist.all_reduce(stats[slot], async_op=True))
            return None, None
        if wl == "store":
            sim = capi.make_sim(n_total, n_steps, prec, seed, lo, per_gpu)
            res = ctx.simulate_trajectories(opt, sim, traj, None, payoffs)
        else:
            sim_o = capi.make_sim(n_total, n_steps, prec, seed, lo, per_gpu)
            sim_i = capi.make_sim(n_total, n_steps, prec, seed + 100_003, lo, per_gpu, nmc["n_inner"])
            if nmc["variant"] == -1:
                res = ctx.nmc_fused(nmc["opt"], sim_i, seed, nmc["traj"], nmc["cnt"], nmc["out"])
                outer_ms = 0.0
            else:
                r_o = ctx.simulate_trajectories(nmc["opt"], sim_o, nmc["traj"], nmc["cnt"])
                res = ctx.nmc_inner(nmc["opt"], sim_i, nmc["traj"], nmc["cnt"], nmc["out"], variant=nmc["variant"])
                outer_ms = r_o.kernel_ms
            work, live, k_ms_local = res.work_steps, res.live_steps, res.kernel_ms
            if world > 1:
                # the one collective of the path: the shard's statistics record (sum and count of its point prices,
                # with the work counters riding along), summed over the ranks; per-point prices stay on their GPU
                s, s2, n, work, live = sharding.allreduce_vector([res.sum, res.sumsq, res.n, work, live], device=coll_device)
                res = capi.finalize_nmc_stats([s, s2, work / 64.0, live, 0.0, n])
                res.kernel_ms = k_ms_local
            nmc_runs.append((k_ms_local, outer_ms, work, live))   # work / live: whole job (all ranks) per pass
            return res, res
        if world > 1:
            s, s2, n = sharding.allreduce_stats(res.sum, res.sumsq, res.n, device=coll_device)
            fin = capi.finalize(s, s2, n, opt.r, opt.T)
        else:
            fin = res
        return res, fin

    def fence():
        for w in pending:
            w.wait()
        pending.clear()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(-1 - i, args.steps + i)
    fence()
    nmc_runs.clear()
    t0 = time.perf_counter()
    kernel_ms = []
    fin = res = None
    for i in range(args.steps):
        res, fin = one_step(i, i)
        if res is not None:
            kernel_ms.append(res.kernel_ms)
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if is_price:
        kernel_ms = ctx.enqueued_kernel_ms(min(args.steps, 64))
        fin = capi.finalize_stats(stats[args.steps - 1, :6].tolist(), opt.r, opt.T)

    line = None
    if rank == 0:
        units = n_total * args.steps
        avg_kernel_s = sum(kernel_ms) / len(kernel_ms) / 1e3
        line = {
            "metric": METRIC,
            "value": units / elapsed, "unit": "paths/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.global_paths else "weak", "vs_baseline": None,
            "dtype": "f32" if prec == capi.F32 else "f64", "data": "synthetic",
            "config": {"workload": workload_name(wl, per_gpu, n_total, world, bool(args.global_paths), args.nmc_window),
                       "paths_per_gpu": per_gpu, "n_steps": n_steps, "global_paths": n_total,
                       "sharding": f"path-id ranges over {world} rank(s), one {args.backend} all-reduce of (sum,sumsq,n) per step"
                       if world > 1 else "single GPU", "seed": "1234+step", "rng": "Philox4x32-10, subsequence = global path id"},
            "path_steps_per_s": units * n_steps / elapsed,
            "kernel_ms_avg": avg_kernel_s * 1e3, "kernel_ms_min": min(kernel_ms), "kernel_ms_max": max(kernel_ms),
            "device": device,
        }
        if wl == "nmc":
            n_inner = nmc["n_inner"]
            k_ms = sum(r[0] for r in nmc_runs) / len(nmc_runs)
            o_ms = sum(r[1] for r in nmc_runs) / len(nmc_runs)
            work = sum(r[2] for r in nmc_runs) / len(nmc_runs)          # executed inner lane-steps per pass
            inner_paths = per_gpu * world * n_steps * n_inner              # inner paths per pass (all points)
            line["value"] = (n_total + inner_paths) * args.steps / elapsed
            line["config"]["value_counts"] = "outer + inner paths per second (inner paths of closed-window points included: they are priced, at zero)"
            line["config"]["strategy"] = args.nmc_strategy
            line["config"]["window"] = nmc["win"]
            line["inner_paths_per_s"] = inner_paths * args.steps / elapsed
            live = sum(r[3] for r in nmc_runs) / len(nmc_runs)         # of those, lane-steps of paths whose window was open
            line["executed_inner_path_steps_per_pass"] = work
            line["live_inner_path_steps_per_pass"] = live
            line["lane_efficiency"] = live / work if work else None
            line["executed_inner_path_steps_per_s"] = work * args.steps / elapsed
            line["european_window_inner_path_steps"] = n_total * n_inner * (n_steps * (n_steps - 1) // 2)
            line["inner_kernel_ms"] = k_ms
            line["outer_kernel_ms"] = o_ms
            line["mean_point_price"] = fin.price
            line["path_steps_per_s"] = None
            # imbalance: kernel time against the time the executed steps would take at the in-register kernel's rate
            key = "nmc_wave_f64_window"
            line["roofline"] = valu_roofline(W, stale, key, f"nmc_{args.nmc_strategy}_kernel<double,window>",
                                             work / world / (k_ms / 1e3),   # per GPU: rank 0's kernel time, 1/world of the job's steps
                                             extra={"work": "64 lanes x steps each wavefront ran; W is the loop over fresh paths, resumed batches run the dearer loop",
                                                    "valu_slots_resumed_batches": W.get("nmc_wave_f64_window_resumed")})
        if wl != "nmc":
            line.update({"price": fin.price, "std_err": fin.std_err, "ci95": [fin.ci_lo, fin.ci_hi],
                         "bs_closed_form": BS_EXACT, "abs_err_vs_bs": abs(fin.price - BS_EXACT),
                         "within_3se": abs(fin.price - BS_EXACT) <= 3 * fin.std_err,
                         "within_1e-4": abs(fin.price - BS_EXACT) <= 1e-4})
        # roofline of the dominant kernel, from the library's HIP events on the launch stream
        if wl == "store":
            bytes_per_launch = per_gpu * n_steps * 4 + per_gpu * 4 + 16 * res.grid   # trajectories + payoffs + block records
            ach = bytes_per_launch / avg_kernel_s / 1e9
            line["roofline"] = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": ach / PEAK_HBM_GBS, "traffic": pmc_traffic_bytes("store_kernel<float"),
                                "kernel": "store_kernel<float,false,STEP_MAJOR,vec>",
                                "algorithmic_bytes_per_launch": bytes_per_launch}
            if per_gpu % 4 == 0:   # the same store stream with nothing simulated, right after the timed steps: this box's ceiling
                cs = [ctx.diag_store_pattern(per_gpu, n_steps, capi.F32, traj, payoffs) for _ in range(3)]
                cms = sum(cs) / len(cs)
                line["roofline"]["same_run_ceiling"] = {"kernel_ms": cms, "launches": len(cs),
                                                        "GB_per_s": (per_gpu * n_steps * 4 + per_gpu * 4) / (cms / 1e3) / 1e9}
                line["roofline"]["frac_of_same_run_store_ceiling"] = cms / (avg_kernel_s * 1e3)
        elif wl in ("european252", "european252_f32"):
            f64 = prec == capi.F64
            line["roofline"] = valu_roofline(W, stale, "price_f64" if f64 else "price_f32",
                                             f"price_kernel<{'double' if f64 else 'float'},no window,log-space>",
                                             per_gpu * n_steps / avg_kernel_s,
                                             "price_kernel<double, false, true, 0>" if f64 else "price_kernel<float, false, true, 0>",
                                             build_id=capi.build_id())

    solo = rank == 0 and world == 1

    if solo and wl == "european252" and not args.no_store_roofline and args.store_leg_first:
        line["roofline_store"] = store_leg(ctx, capi, torch, opt, "right after the headline")

    # the north-star sweep: 1M / 10M / 100M paths x 252 steps, in-register, both precisions.  Synchronous calls
    # (mcamd_price_paths): kernel_ms = HIP events around the simulation kernel, call_ms = host wall time of the whole
    # call (launches + final reduce + 16-byte copy + sync), so call_ms - kernel_ms is the per-call overhead.
    if solo and wl == "european252" and not args.no_sweep:
        sweep = []
        for p_, key in ((capi.F64, "price_f64"), (capi.F32, "price_f32")):
            for n in (1_000_000, 10_000_000, 100_000_000):
                ks, cs = [], []
                r_ = None
                for rep in range(6):
                    torch.cuda.synchronize()
                    tc = time.perf_counter()
                    r_ = ctx.price_paths(opt, capi.make_sim(n, 252, p_, 4321 + rep))
                    cs.append((time.perf_counter() - tc) * 1e3)
                    ks.append(r_.kernel_ms)
                k, c = median(ks[1:]), median(cs[1:])
                w = W.get(key)
                e = {"paths": n, "steps": 252, "dtype": "f64" if p_ == capi.F64 else "f32", "kernel_ms": k, "call_ms": c,
                     "call_over_kernel": c / k, "paths_per_s": n / (c / 1e3), "paths_per_s_kernel": n / (k / 1e3),
                     "roofline_frac": (n * 252 / (k / 1e3)) * w / 1e12 / PEAK_VALU_TLANEOPS if w else None,
                     "roofline_frac_vs_floor": (n * 252 / (k / 1e3)) * W_FLOOR[key] / 1e12 / PEAK_VALU_TLANEOPS,
                     # a wavefront runs whole paths: ceil(n / 64) wavefront-paths over 1024 SIMDs, the fullest SIMD sets the time
                     "wave_quantization_ceiling": (n / 64 / 1024) / math.ceil(math.ceil(n / 64) / 1024),
                     "valu_slots_per_path_step": w, "price": r_.price, "std_err": r_.std_err,
                     "abs_err_vs_bs": abs(r_.price - BS_EXACT), "within_3se": abs(r_.price - BS_EXACT) <= 3 * r_.std_err}
                sweep.append(e)
        line["sweep"] = sweep

    # the opt-in product form (MCAMD_FLAG_PRODUCT_FORM): St *= exp(...) every step, the reference's recurrence as written
    # and the form whose terminal price is the same bits as the store kernel's last row.  Reported beside the headline
    # (which sums the log-returns and exponentiates once: same draws, same scheme).  Measured BEFORE the store pass: the
    # clock stays low for a while after 100 GB of stores.
    if solo and wl == "european252":
        ks = []
        for i in range(12):
            rl = ctx.price_paths(opt, capi.make_sim(n_total, n_steps, prec, 1234 + i, lo, per_gpu, flags=capi.FLAG_PRODUCT_FORM))
            ks.append(rl.kernel_ms)
        kms = sum(ks[2:]) / len(ks[2:])
        line["product_form_mode"] = {"kernel_ms": kms, "paths_per_s_kernel": per_gpu / (kms / 1e3), "price": rl.price,
                                     "std_err": rl.std_err, "abs_err_vs_bs": abs(rl.price - BS_EXACT),
                                     "valu_slots_per_path_step": W.get("price_f64_product")}

    if solo and wl == "european252" and not args.no_store_roofline and not args.store_leg_first:
        line["roofline_store"] = store_leg(ctx, capi, torch, opt, "after the sweep and the product-form leg")

    # "price within 1e-4 of closed form" on a 252-step BASELINE shape: the plain estimator would need > 2.6e10 paths
    # (sigma_payoff = 16.1), so this uses the engine's variance reduction — antithetic pairs + S_T control variate,
    # the same kernel family — on 1e9 samples (BASELINE configs[4]'s count; a sample = one antithetic PAIR), fp64.
    if solo and wl == "european252" and not args.no_accuracy:
        fl = capi.FLAG_ANTITHETIC | capi.FLAG_CONTROL_VARIATE
        n_acc = args.accuracy_pairs
        t_acc = time.perf_counter()
        ra = ctx.price_paths(opt, capi.make_sim(n_acc, 252, capi.F64, 20260101, flags=fl))
        line["accuracy_252"] = {
            "workload": f"European call, {n_acc} antithetic pairs x 252 steps, fp64, in-register, S_T control variate",
            "samples": n_acc, "path_evaluations": 2 * n_acc, "price": ra.price, "std_err": ra.std_err,
            "ci95": [ra.ci_lo, ra.ci_hi], "abs_err_vs_bs": abs(ra.price - BS_EXACT),
            "within_1e-4": abs(ra.price - BS_EXACT) <= 1e-4, "within_3se": abs(ra.price - BS_EXACT) <= 3 * ra.std_err,
            "cv_rho": ra.cv_rho, "cv_beta": ra.cv_beta, "kernel_ms": ra.kernel_ms, "seconds": time.perf_counter() - t_acc,
            "variance_ratio_vs_plain": (16.109 * math.exp(-0.1) / math.sqrt(n_acc) / ra.std_err) ** 2 if ra.std_err > 0 else None}
        # and the exact one-step scheme (BASELINE configs[0]'s scheme on the GPU), plain estimator, 1e11 paths
        opt_acc = capi.make_option(**OPTION, B=0.0, P1=0, P2=1, use_window=1)   # a kernel symbol of its own in profiles
        t_acc = time.perf_counter()
        r1 = ctx.price_paths(opt_acc, capi.make_sim(100_000_000_000, 1, capi.F64, 1234))
        line["accuracy_demo"] = {"workload": "European call, 1e11 paths x 1 exact step, fp64, in-register, plain estimator",
                                 "paths": 100_000_000_000, "price": r1.price, "std_err": r1.std_err,
                                 "abs_err_vs_bs": abs(r1.price - BS_EXACT), "within_1e-4": abs(r1.price - BS_EXACT) <= 1e-4,
                                 "within_3se": abs(r1.price - BS_EXACT) <= 3 * r1.std_err,
                                 "seconds": time.perf_counter() - t_acc, "kernel_ms": r1.kernel_ms}

    # the reference's bullet option (hello.cu:11-13: B = 120, P1 = 10, P2 = 50) on configs[1]'s shape, 10M paths x 252
    # steps, fp64: the window closes for most paths after ~51 steps, and from 3.1M paths on the library prices such
    # jobs with its lane-compacting kernel (csrc/price_impl.hpp)
    if solo and wl == "european252" and not args.no_nmc:
        try:
            optb = capi.make_option(**OPTION, **BULLET)
            rb_ = [ctx.price_paths(optb, capi.make_sim(n_total, n_steps, prec, 1234 + i, lo, per_gpu)) for i in range(6)]
            kb = sum(r.kernel_ms for r in rb_[2:]) / len(rb_[2:])
            line["bullet_252"] = {"workload": f"bullet call B=120 P1=10 P2=50, {per_gpu} paths x {n_steps} steps, fp64, in-register",
                                  "kernel_ms": kb, "paths_per_s_kernel": per_gpu / (kb / 1e3), "price": rb_[-1].price,
                                  "std_err": rb_[-1].std_err, "grid": rb_[-1].grid,
                                  "kernel": "price_window_compact_kernel<double>" if rb_[-1].grid <= 4096 else "price_kernel<double,window>"}
        except Exception as e:
            line["bullet_252"] = {"error": str(e)}

    # BASELINE configs[3] beside the headline: nested MC, 65 536 outer x 252 steps x 1000 inner paths, fp64, with the
    # reference's bullet window (hello.cu:11-13), two-launch route (outer store + wave-per-point inner stage), one
    # untimed and two timed passes (~0.2 s each).  `--workload nmc` is the full line for this config.
    if solo and wl == "european252" and not args.no_nmc:
        try:
            n4, s4, i4 = 65_536, 252, 1000
            opt4 = capi.make_option(**OPTION, **BULLET)
            t4 = torch.empty(n4 * s4, dtype=torch.float64, device="cuda")
            c4 = torch.empty(n4 * s4, dtype=torch.int32, device="cuda")
            o4 = torch.empty(n4 * s4, dtype=torch.float64, device="cuda")
            runs = []
            for i in range(3):
                ro = ctx.simulate_trajectories(opt4, capi.make_sim(n4, s4, capi.F64, 1234 + i, 0, n4), t4, c4)
                ri = ctx.nmc_inner(opt4, capi.make_sim(n4, s4, capi.F64, 1234 + i + 100_003, 0, n4, i4), t4, c4, o4,
                                   variant=capi.NMC_WAVE_PER_POINT)
                runs.append((ro.kernel_ms, ri.kernel_ms, ri.work_steps, ri.live_steps, ri.price))
            runs = runs[1:]
            k4 = sum(r[1] for r in runs) / len(runs)
            w4 = sum(r[2] for r in runs) / len(runs)
            l4 = sum(r[3] for r in runs) / len(runs)
            rf = valu_roofline(W, stale, "nmc_wave_f64_window", "nmc_wave_kernel<double,window>", w4 / (k4 / 1e3),
                               extra={"work": "64 lanes x steps each wavefront ran; W is the loop over fresh paths, resumed batches run the dearer loop",
                                                    "valu_slots_resumed_batches": W.get("nmc_wave_f64_window_resumed")})
            line["nmc_config4"] = {
                "workload": "BASELINE configs[3]: nested MC 65536 x 252 points x 1000 inner paths, fp64, bullet window B=120 P1=10 P2=50",
                "outer_kernel_ms": sum(r[0] for r in runs) / len(runs), "inner_kernel_ms": k4,
                "inner_paths_per_s": n4 * s4 * i4 / (k4 / 1e3), "executed_inner_path_steps": w4,
                "live_inner_path_steps": l4, "lane_efficiency": l4 / w4 if w4 else None,
                "european_window_inner_path_steps": n4 * i4 * (s4 * (s4 - 1) // 2),
                "mean_point_price": runs[-1][4], "roofline": rf}
            del t4, c4, o4
        except Exception as e:  # the headline must survive a failure of the side measurement
            line["nmc_config4"] = {"error": str(e)}

    if solo and not args.no_cpu_baseline:
        if wl == "nmc":
            line["cpu_baseline"] = cpu_nmc_baseline(nmc["win"], per_gpu, n_steps, nmc["n_inner"], nmc["traj"], nmc["cnt"])
        else:
            line.update(cpu_baseline(n_steps, args.cpu_sample_paths))
        if wl == "european252":
            line["cfg1"] = cpu_cfg1(capi)

    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
