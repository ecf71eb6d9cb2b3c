#!/usr/bin/env python3
"""BASELINE configs[3] with the reference's bullet window, once per nested-MC strategy (wave per point, block per
point, fused outer + inner): the command tools/profile.sh profiles in mode nmc_all, so that every nmc kernel has a
rocprofv3 row and PMC counters under profiles/.  Prints one JSON line with the three timings."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    pkg = importlib.import_module("monte-carlo-project-cuda_amd")
    capi = pkg.capi
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0, stream.cuda_stream)
    n_paths, n_steps, n_inner = 65_536, 252, 1000
    opt = capi.make_option(100.0, 1.0, 100.0, 0.1, 0.2, B=120.0, P1=10, P2=50, use_window=1)
    outer = capi.make_sim(n_paths, n_steps, capi.F64, seed=1234)
    inner = capi.make_sim(n_paths, n_steps, capi.F64, seed=1235, n_paths_inner=n_inner)
    traj = torch.empty(n_paths * n_steps, dtype=torch.float64, device="cuda")
    cnt = torch.empty(n_paths * n_steps, dtype=torch.int32, device="cuda")
    out = torch.empty(n_paths * n_steps, dtype=torch.float64, device="cuda")
    ro = ctx.simulate_trajectories(opt, outer, traj, cnt)
    res = {"outer_kernel_ms": ro.kernel_ms}
    for name, variant in (("wave", capi.NMC_WAVE_PER_POINT), ("block", capi.NMC_BLOCK_PER_POINT),
                          ("block_plain", capi.NMC_BLOCK_PER_POINT_PLAIN)):
        r = ctx.nmc_inner(opt, inner, traj, cnt, out, variant=variant)
        res[name] = {"kernel_ms": r.kernel_ms, "work_steps": r.work_steps, "live_steps": r.live_steps,
                     "lane_efficiency": r.live_steps / r.work_steps, "mean_point_price": r.price,
                     "lane_steps_per_s": r.work_steps / (r.kernel_ms / 1e3)}
    r = ctx.nmc_fused(opt, inner, 1234, traj, cnt, out)
    res["fused"] = {"kernel_ms": r.kernel_ms, "work_steps": r.work_steps, "live_steps": r.live_steps,
                    "lane_efficiency": r.live_steps / r.work_steps, "mean_point_price": r.price,
                    "lane_steps_per_s": r.work_steps / (r.kernel_ms / 1e3)}
    ctx.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
