import importlib, sys, statistics
sys.path.insert(0, '.')
import torch
pkg = importlib.import_module("monte-carlo-project-cuda_amd"); capi = pkg.capi
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = capi.Context(0, s.cuda_stream)
opt = capi.make_option(100.0, 1.0, 100.0, 0.1, 0.2, B=120.0, P1=10, P2=50, use_window=1)
for prec in (capi.F64, capi.F32):
    n = 10_000_000
    whole = [ctx.price_paths(opt, capi.make_sim(n, 252, prec, 1234, 0, n)) for _ in range(5)]
    parts = [[ctx.price_paths(opt, capi.make_sim(n, 252, prec, 1234, k * 2_500_000, 2_500_000)) for k in range(4)] for _ in range(3)]
    print(prec, "compacting: kernel ms", [round(r.kernel_ms, 3) for r in whole], "grid", whole[0].grid, "price", whole[0].price)
    print(prec, "plain (4 shards): kernel ms", [round(sum(r.kernel_ms for r in p), 3) for p in parts], "sum rel diff",
          abs(sum(r.sum for r in parts[0]) - whole[0].sum) / whole[0].sum)
