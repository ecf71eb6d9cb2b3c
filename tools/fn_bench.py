import importlib, sys, torch
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("monte-carlo-project-cuda_amd"); capi = pkg.capi
ctx = capi.Context(0)
npaths, nsteps = 4_000_000, 252
z = torch.empty(npaths * nsteps, dtype=torch.float32, device="cuda")
ctx.generate_normals(3, z.numel(), capi.F32, z)
for _ in range(3):
    r = ctx.price_from_normals(capi.make_option(), capi.make_sim(npaths, nsteps, capi.F32), z)
print(r.kernel_ms)
