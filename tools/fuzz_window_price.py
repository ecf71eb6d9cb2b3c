#!/usr/bin/env python3
"""Differential fuzz of window (bullet) pricing on the GPU: a job large enough for the lane-compacting kernel
(csrc/price_impl.hpp: >= 12 groups of 1024 paths per CU) against the same path ids priced as shards small enough for the
one-path-per-thread kernel.  Same Philox streams, same arithmetic per path: the sums may differ by summation order only.
    python3 tools/fuzz_window_price.py --seconds 120 --seed 1
Prints one JSON line: cases run, worst relative deviation of the fp64 sums, failures (empty when all agree)."""
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
    threshold = ctx.device_info().compute_units * 12 * 1024
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    n_cases, worst, fails = 0, 0.0, []
    while time.time() - t0 < args.seconds:
        prec = capi.F64 if rng.random() < 0.7 else capi.F32
        n = int(rng.integers(threshold, threshold + 3_000_000))
        n_steps = int(rng.integers(8, 130))
        Tk = int(rng.integers(0, n_steps - 8)) if (n_steps > 9 and rng.random() < 0.3) else 0
        Ik = int(rng.integers(0, 6)) if Tk else 0
        Sk = float(rng.choice([90.0, 100.0, 118.0])) if Tk else 0.0
        B = 100.0 * float(rng.choice([0.0, 0.9, 1.0, 1.05, 1.2, 3.0]))
        P1 = int(rng.integers(0, max(1, n_steps // 2)))
        P2 = int(rng.integers(P1, n_steps + 2)) if rng.random() < 0.9 else 2**31 - 1
        flags = capi.FLAG_PRODUCT_FORM if rng.random() < 0.4 else 0   # the reference's recurrence, or the default (log space)
        v = float(rng.choice([0.05, 0.2, 0.6]))
        opt = capi.make_option(100.0, 1.0, 100.0, 0.1, v, B=B, P1=P1, P2=P2, use_window=1, Ik=Ik, Sk=Sk, Tk=Tk)
        seed = int(rng.integers(1, 1 << 31))
        off = int(rng.integers(0, 1 << 45))
        total = off + n + int(rng.integers(0, 1000))
        whole = ctx.price_paths(opt, capi.make_sim(total, n_steps, prec, seed, off, n, flags=flags))
        shard = int(rng.integers(threshold // 3, threshold - 1))
        s1 = s2 = 0.0
        a = 0
        plain_grids = []
        while a < n:
            b = min(n, a + shard)
            r = ctx.price_paths(opt, capi.make_sim(total, n_steps, prec, seed, off + a, b - a, flags=flags))
            s1 += r.sum
            s2 += r.sumsq
            plain_grids.append(r.grid)
            a = b
        tol = 1e-11 if prec == capi.F64 else 1e-9
        d1 = abs(whole.sum - s1) / max(abs(s1), 1e-300) if s1 else abs(whole.sum)
        d2 = abs(whole.sumsq - s2) / max(abs(s2), 1e-300) if s2 else abs(whole.sumsq)
        worst = max(worst, d1, d2)
        # the whole job ran the compacting kernel (persistent grid, 4 workgroups per CU), every shard the plain one (a
        # block per 256 paths, or per 256 * ceil(32 / steps) for short paths)
        per_thread = 1 if n_steps - Tk >= 32 else -(-32 // (n_steps - Tk))
        plain_ok = all(g == -(-(-(-min(shard, n - i * shard) // per_thread)) // 256) for i, g in enumerate(plain_grids))
        ok = d1 <= tol and d2 <= tol and whole.grid == threshold // (12 * 1024) * 4 and plain_ok and np.isfinite(whole.sum)
        if not ok:
            fails.append({"case": n_cases, "prec": prec, "n": n, "n_steps": n_steps, "Tk": Tk, "Ik": Ik, "Sk": Sk, "B": B, "P1": P1,
                          "P2": P2, "flags": flags, "v": v, "seed": seed, "off": off, "d1": d1, "d2": d2,
                          "grid": whole.grid, "plain_grid": min(plain_grids)})
        n_cases += 1
    print(json.dumps({"cases": n_cases, "worst_rel_dev": worst, "failures": fails[:10], "n_failures": len(fails)}))
    ctx.close()
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
