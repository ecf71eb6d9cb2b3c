#!/usr/bin/env python3
"""Same-process A/B of the two window-less in-register loops of one library: the default (log-returns summed as
Box-Muller pair sums, one exponential per path) against MCAMD_FLAG_PRODUCT_FORM (St *= exp(...) every step), alternating
launches, median kernel time.  Run on an MI355X."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("monte-carlo-project-cuda_amd"); capi = pkg.capi
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = capi.Context(0, stream.cuda_stream)
opt = capi.make_option()
med = lambda xs: sorted(xs)[len(xs) // 2]
for _ in range(20):
    ctx.price_paths(opt, capi.make_sim(10_000_000, 252, capi.F64, 1))
for prec in (capi.F64, capi.F32):
    for n in (1_000_000, 10_000_000, 100_000_000):
        k = {0: [], 1: []}
        for rep in range(11):
            for i, fl in enumerate((0, capi.FLAG_PRODUCT_FORM)):
                r = ctx.price_paths(opt, capi.make_sim(n, 252, prec, 100 + rep, flags=fl))
                k[i].append(r.kernel_ms)
        print(json.dumps({"paths": n, "dtype": prec, "default_ms": med(k[0]), "product_form_ms": med(k[1]),
                          "ratio": med(k[0]) / med(k[1]), "build_id": capi.build_id()}))
