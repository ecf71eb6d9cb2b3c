#!/usr/bin/env python3
"""BASELINE configs[3] (bullet window) inner stage, wave-per-point kernel: the default loop (ln(St/S0) carried)
against MCAMD_FLAG_PRODUCT_FORM (St *= exp(...) every step), alternating launches in one process.  Run on an MI355X."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("monte-carlo-project-cuda_amd"); capi = pkg.capi
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = capi.Context(0, stream.cuda_stream)
n_paths, n_steps, n_inner = 65_536, 252, 1000
opt = capi.make_option(100.0, 1.0, 100.0, 0.1, 0.2, B=120.0, P1=10, P2=50, use_window=1)
traj = torch.empty(n_paths * n_steps, dtype=torch.float64, device="cuda")
cnt = torch.empty(n_paths * n_steps, dtype=torch.int32, device="cuda")
out = {f: torch.empty(n_paths * n_steps, dtype=torch.float64, device="cuda") for f in (0, capi.FLAG_PRODUCT_FORM)}
ctx.simulate_trajectories(opt, capi.make_sim(n_paths, n_steps, capi.F64, seed=1234), traj, cnt)
ms = {0: [], capi.FLAG_PRODUCT_FORM: []}
for rep in range(4):
    for fl in (capi.FLAG_PRODUCT_FORM, 0):
        r = ctx.nmc_inner(opt, capi.make_sim(n_paths, n_steps, capi.F64, seed=1235, n_paths_inner=n_inner, flags=fl), traj, cnt, out[fl])
        ms[fl].append(r.kernel_ms)
        last = r
P = capi.FLAG_PRODUCT_FORM
dev = float(((out[0] - out[P]).abs() / (out[P].abs() + 1e-9)).max().item())
print(json.dumps({"product_form_ms": sorted(ms[P])[1], "default_ms": sorted(ms[0])[1], "max_rel_dev_point_prices": dev,
                  "points_differing": int((out[0] != out[P]).sum().item()), "points": n_paths * n_steps,
                  "build_id": capi.build_id()}))
