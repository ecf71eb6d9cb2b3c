import importlib, sys, torch, json
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("monte-carlo-project-cuda_amd"); capi = pkg.capi
ctx = capi.Context(0)
out = {}
n = 1 << 30
x = torch.ones(n, dtype=torch.float32, device="cuda")
for v in (3, 4, 5, 6):
    ms = []
    for i in range(4):
        s, m = ctx.reduce_sum(x, n, capi.F32, v); ms.append(m)
    assert s == float(n)
    out[f"reduce_f32_variant{v}"] = {"ms": min(ms[1:]), "GBs": n * 4 / (min(ms[1:]) * 1e-3) / 1e9}
xd = torch.ones(n // 2, dtype=torch.float64, device="cuda")
ms = [ctx.reduce_sum(xd, n // 2, capi.F64, 6)[1] for _ in range(4)]
out["reduce_f64_variant6"] = {"ms": min(ms[1:]), "GBs": n * 4 / (min(ms[1:]) * 1e-3) / 1e9}
del xd
for prec, name, b in ((capi.F32, "normals_f32", 4), (capi.F64, "normals_f64", 8)):
    m = n if prec == capi.F32 else n // 2
    buf = x if prec == capi.F32 else torch.empty(m, dtype=torch.float64, device="cuda")
    ms = [ctx.generate_normals(7, m, prec, buf) for _ in range(4)]
    out[name] = {"ms": min(ms[1:]), "GBs": m * b / (min(ms[1:]) * 1e-3) / 1e9, "normals_per_s": m / (min(ms[1:]) * 1e-3)}
# array-driven: 4M paths x 252 steps fp32 (4 GB of normals)
npaths, nsteps = 4_000_000, 252
z = torch.empty(npaths * nsteps, dtype=torch.float32, device="cuda")
ctx.generate_normals(3, z.numel(), capi.F32, z)
ms = [ctx.price_from_normals(capi.make_option(), capi.make_sim(npaths, nsteps, capi.F32), z).kernel_ms for _ in range(4)]
out["from_normals_f32"] = {"ms": min(ms[1:]), "GBs": z.numel() * 4 / (min(ms[1:]) * 1e-3) / 1e9}
print(json.dumps(out, indent=1))
