#!/usr/bin/env python3
"""Same-process A/B of the in-kernel finish (default) against the separate reduction launch (MCAMD_FLAG_SEPARATE_REDUCE):
kernel time (HIP events) and whole-call wall time of mcamd_price_paths, alternating, medians.  Run on an MI355X."""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("monte-carlo-project-cuda_amd"); capi = pkg.capi
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = capi.Context(0, stream.cuda_stream)
opt = capi.make_option()
med = lambda xs: sorted(xs)[len(xs) // 2]
for _ in range(20):   # clocks up
    ctx.price_paths(opt, capi.make_sim(10_000_000, 252, capi.F64, 1))
out = []
for prec in (capi.F64, capi.F32):
    for n in (100_000, 1_000_000, 2_000_000):
        k = {0: [], 8: []}; c = {0: [], 8: []}
        for rep in range(31):
            for fl in (0, capi.FLAG_SEPARATE_REDUCE):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = ctx.price_paths(opt, capi.make_sim(n, 252, prec, 100 + rep, flags=fl))
                c[fl].append((time.perf_counter() - t0) * 1e3); k[fl].append(r.kernel_ms)
        e = {"paths": n, "dtype": prec, "folded_kernel_ms": med(k[0]), "folded_call_ms": med(c[0]),
             "separate_kernel_ms": med(k[8]), "separate_call_ms": med(c[8])}
        e["call_over_kernel_folded"] = e["folded_call_ms"] / e["folded_kernel_ms"]
        e["call_over_kernel_separate"] = e["separate_call_ms"] / e["separate_kernel_ms"]
        out.append(e); print(json.dumps(e))
