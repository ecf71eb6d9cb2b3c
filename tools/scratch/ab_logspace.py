import importlib, sys, statistics
sys.path.insert(0, '.')
import torch
pkg = importlib.import_module("monte-carlo-project-cuda_amd"); capi = pkg.capi
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = capi.Context(0, s.cuda_stream)
opt = capi.make_option(S0=100.0, K=100.0, T=1.0, r=0.1, v=0.2)
def run(flags, n=10_000_000, reps=40):
    ks = []
    for i in range(reps):
        r = ctx.price_paths(opt, capi.make_sim(n, 252, capi.F64, 1234 + i, 0, n, flags=flags))
        ks.append(r.kernel_ms)
    ks = ks[5:]
    return statistics.mean(ks), min(ks), r.price
for rnd in range(3):
    print("product ", run(0)); print("logspace", run(capi.FLAG_LOG_SPACE))
