#!/usr/bin/env python3
"""Times the kernels beside the two pricing paths at sizes the 256 MiB Infinity Cache cannot flatter (>= 32 GB per
pass): the four reduce schedules, the bulk normal fills and the array-driven pricer.  One JSON object on stdout;
tools/profile.sh-style rocprofv3 rows of the same command go to profiles/ beside it.
    python tools/aux_bench.py [gigabytes=32]
"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    gb = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
    pkg = importlib.import_module("monte-carlo-project-cuda_amd")
    capi = pkg.capi
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0, stream.cuda_stream)
    out = {"bytes_per_pass": None}
    n = int(gb * (1 << 30)) // 4 // 1024 * 1024          # fp32 elements
    out["bytes_per_pass"] = n * 4
    x = torch.ones(n, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for v in (3, 4, 5, 6):
        ms = []
        for _ in range(5):
            s, m = ctx.reduce_sum(x, n, capi.F32, v)
            ms.append(m)
        assert s == float(n), (v, s, n)
        out[f"reduce_f32_variant{v}"] = {"ms": min(ms[1:]), "GBs": n * 4 / (min(ms[1:]) * 1e-3) / 1e9}
    xd = x.view(torch.float64)                              # same bytes read as n/2 doubles (the values do not matter)
    for v in (3, 6):
        ms = [ctx.reduce_sum(xd, n // 2, capi.F64, v)[1] for _ in range(5)]
        out[f"reduce_f64_variant{v}"] = {"ms": min(ms[1:]), "GBs": n * 4 / (min(ms[1:]) * 1e-3) / 1e9}
    for prec, name, b in ((capi.F32, "normals_f32", 4), (capi.F64, "normals_f64", 8)):
        m_ = n if prec == capi.F32 else n // 2
        buf = x if prec == capi.F32 else xd
        ms = [ctx.generate_normals(7, m_, prec, buf) for _ in range(5)]
        out[name] = {"ms": min(ms[1:]), "GBs": m_ * b / (min(ms[1:]) * 1e-3) / 1e9, "normals_per_s": m_ / (min(ms[1:]) * 1e-3)}
    # array-driven pricer over the same buffer: paths x 252 steps of fp32 normals (just filled), then fp64 over half as many
    nsteps = 252
    npaths = n // nsteps
    ctx.generate_normals(3, npaths * nsteps, capi.F32, x)
    ms = [ctx.price_from_normals(capi.make_option(), capi.make_sim(npaths, nsteps, capi.F32), x).kernel_ms for _ in range(5)]
    out["from_normals_f32"] = {"paths": npaths, "ms": min(ms[1:]), "GBs": npaths * nsteps * 4 / (min(ms[1:]) * 1e-3) / 1e9}
    npd = n // 2 // nsteps
    ctx.generate_normals(3, npd * nsteps, capi.F64, xd)
    ms = [ctx.price_from_normals(capi.make_option(), capi.make_sim(npd, nsteps, capi.F64), xd).kernel_ms for _ in range(5)]
    out["from_normals_f64"] = {"paths": npd, "ms": min(ms[1:]), "GBs": npd * nsteps * 8 / (min(ms[1:]) * 1e-3) / 1e9}
    ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
