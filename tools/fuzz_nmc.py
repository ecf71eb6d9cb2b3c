#!/usr/bin/env python3
"""Differential fuzz of the nested-MC strategies on the GPU (no oracle: the PLAIN block-per-point kernel, which does
not compact lanes, is the reference for the compacting wave-per-point, block-per-point and fused kernels).  Runs for --seconds.
    python3 tools/fuzz_nmc.py --seconds 120 --seed 1
Prints one JSON line: cases run, worst relative deviation, failures (empty when all agree)."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    pkg = importlib.import_module("monte-carlo-project-cuda_amd")
    capi = pkg.capi
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0, stream.cuda_stream)
    rng = np.random.default_rng(args.seed)
    tt = {capi.F32: torch.float32, capi.F64: torch.float64}
    t0 = time.time()
    n_cases, worst, fails = 0, 0.0, []
    while time.time() - t0 < args.seconds:
        prec = capi.F64 if rng.random() < 0.7 else capi.F32
        n_paths = int(rng.integers(1, 41))
        n_steps = int(rng.integers(2, 90))
        n_inner = int(rng.choice([1, 7, 63, 64, 65, 100, 129, 500, 1000, 1537, 3000]))
        B = 100.0 * float(rng.choice([0.0, 0.85, 0.97, 1.0, 1.03, 1.1, 1.3, 3.0]))
        P1 = int(rng.integers(0, n_steps + 1))
        P2 = int(rng.integers(P1, n_steps + 2)) if rng.random() < 0.9 else 2**31 - 1
        flags = capi.FLAG_PRODUCT_FORM if rng.random() < 0.4 else 0   # the reference's recurrence, or the default (log space)
        layout = capi.STEP_MAJOR if rng.random() < 0.5 else capi.PATH_MAJOR
        v = float(rng.choice([0.05, 0.2, 0.6]))
        opt = capi.make_option(100.0, 1.0, 100.0, 0.1, v, B=B, P1=P1, P2=P2, use_window=1)
        so, si = int(rng.integers(1, 1 << 30)), int(rng.integers(1 << 30, 1 << 31))
        # the job is a SHARD [lo, lo + n_paths) of a larger one (what a rank of a multi-GPU run prices): lo feeds the
        # outer stream, the inner subsequences and the cut of the compaction pools (multiples of 8 of the global id)
        r_ = rng.random()
        lo = 0 if r_ < 0.3 else int(rng.integers(1, 64)) if r_ < 0.7 else int(rng.integers(1, 1 << 40))
        n_total = lo + n_paths + int(rng.integers(0, 100))
        outer = capi.make_sim(n_total, n_steps, prec, seed=so, path_offset=lo, n_paths_local=n_paths)
        inner = capi.make_sim(n_total, n_steps, prec, seed=si, path_offset=lo, n_paths_local=n_paths, n_paths_inner=n_inner,
                              flags=flags)
        n = n_paths * n_steps
        traj = torch.empty(n, dtype=tt[prec], device="cuda")
        cnt = torch.empty(n, dtype=torch.int32, device="cuda")
        w, b, f, bc = (torch.empty(n, dtype=tt[prec], device="cuda") for _ in range(4))
        ctx.simulate_trajectories(opt, outer, traj, cnt, None, layout)
        rw = ctx.nmc_inner(opt, inner, traj, cnt, w, layout, capi.NMC_WAVE_PER_POINT)
        rb = ctx.nmc_inner(opt, inner, traj, cnt, b, layout, capi.NMC_BLOCK_PER_POINT_PLAIN)
        ctx.nmc_inner(opt, inner, traj, cnt, bc, layout, capi.NMC_BLOCK_PER_POINT)
        t2, c2 = torch.empty_like(traj), torch.empty_like(cnt)
        ctx.nmc_fused(opt, inner, so, t2, c2, f, layout)
        tol = (1e-11, 1e-11) if prec == capi.F64 else (3e-4, 3e-4)
        dev = float(((w - b).abs() / (b.abs() + (1e-3 if prec == capi.F32 else 1e-9))).max().item())
        worst = max(worst, dev) if prec == capi.F64 else worst
        ok = (torch.allclose(w, b, rtol=tol[0], atol=tol[1]) and torch.allclose(bc, b, rtol=tol[0], atol=tol[1])
              and torch.equal(w, f) and torch.equal(t2, traj)
              and torch.equal(c2, cnt) and rw.live_steps <= rw.work_steps and bool(torch.isfinite(w).all()))
        # the two kernels count a path's last, window-closing block differently at most
        ok = ok and abs(rw.live_steps - rb.live_steps) <= 0.25 * max(rb.live_steps, 1.0) + 64 * 4
        if not ok:
            fails.append({"case": n_cases, "prec": prec, "n_paths": n_paths, "n_steps": n_steps, "n_inner": n_inner, "B": B,
                          "P1": P1, "P2": P2, "flags": flags, "layout": layout, "v": v, "so": so, "si": si, "lo": lo, "dev": dev})
        n_cases += 1
    print(json.dumps({"cases": n_cases, "worst_rel_dev_f64": worst, "failures": fails[:10], "n_failures": len(fails)}))
    ctx.close()
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
