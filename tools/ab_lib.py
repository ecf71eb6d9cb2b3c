#!/usr/bin/env python3
"""Same-box A/B of two builds of the library: worker processes alternate (A, B, A, B, ...), each loads ONE library
(MCAMD_LIB) and times a few jobs with the library's own HIP events; medians per library are printed side by side.
    python3 tools/ab_lib.py [--rounds 3] LIB_A LIB_B          # on an MI355X
A variant library comes from monte-carlo-project-cuda_amd/build.py build_variant(tag, ["-DFLAG"]) in the build container."""
import argparse, importlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JOBS = [("price f64 1M x 252", dict(n=1_000_000, prec=64, window=0)),
        ("price f32 1M x 252", dict(n=1_000_000, prec=32, window=0)),
        ("price f64 10M x 252", dict(n=10_000_000, prec=64, window=0)),
        ("price f64 100M x 252", dict(n=100_000_000, prec=64, window=0)),
        ("price f32 10M x 252", dict(n=10_000_000, prec=32, window=0)),
        ("bullet f64 1M x 252 (one path per thread)", dict(n=1_000_000, prec=64, window=1))]


def worker():
    sys.path.insert(0, ROOT)
    import torch
    capi = importlib.import_module("monte-carlo-project-cuda_amd").capi
    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
    ctx = capi.Context(0, stream.cuda_stream)
    med = lambda xs: sorted(xs)[len(xs) // 2]
    plain, bullet = capi.make_option(), capi.make_option(B=120.0, P1=10, P2=50, use_window=1)
    for _ in range(15):
        ctx.price_paths(plain, capi.make_sim(10_000_000, 252, capi.F64, 1))
    out = {"build_id": capi.build_id()}
    for name, j in JOBS:
        ks = [ctx.price_paths(bullet if j["window"] else plain, capi.make_sim(j["n"], 252, j["prec"], 10 + r)).kernel_ms
              for r in range(11)]
        out[name] = med(ks)
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--worker", action="store_true")
    args = ap.parse_args()
    if args.worker:
        return worker()
    res = {lib: [] for lib in args.libs}
    for _ in range(args.rounds):
        for lib in args.libs:
            env = dict(os.environ, MCAMD_LIB=os.path.abspath(lib))
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=env, capture_output=True, text=True)
            line = [l for l in o.stdout.splitlines() if l.startswith("{")]
            if o.returncode or not line:
                print("worker failed for", lib, o.stderr[-500:], file=sys.stderr)
                return 1
            res[lib].append(json.loads(line[0]))
    med = lambda xs: sorted(xs)[len(xs) // 2]
    for name, _ in JOBS:
        row = {os.path.basename(lib): round(med([r[name] for r in rs]), 4) for lib, rs in res.items()}
        print(json.dumps({"job": name, "kernel_ms_median_of_rounds": row,
                          "all": {os.path.basename(lib): [round(r[name], 4) for r in rs] for lib, rs in res.items()}}))
    print(json.dumps({"build_ids": {os.path.basename(lib): rs[0]["build_id"] for lib, rs in res.items()}}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
