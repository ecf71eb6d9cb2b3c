"""The engine's random numbers against rocRAND's own device API on the same GPU.

north_star: "per-thread rocRAND Philox RNG".  The engine hand-writes Philox4x32-10 and the Box-Muller transform;
this test builds a tiny HIP helper around rocrand_kernel.h (rocrand_init(seed, subsequence, 0), rocrand_normal4,
rocrand_normal_double2) and checks, for thousands of subsequences and several blocks each, that
  * a path simulated with vol = 1, drift = 0 over n steps reproduces exp(sum of rocRAND's normals) — i.e. path id ->
    subsequence, step -> position in the stream, exactly as documented in include/mcamd.h;
  * the bulk fill equals rocRAND's subsequence-0 stream.
fp64: 1e-13 (libm-level differences only); fp32: 4e-6 absolute per normal (hardware v_sin/v_cos/v_log vs ocml)."""
import ctypes as C
import importlib
import math
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
pkg = importlib.import_module("monte-carlo-project-cuda_amd")
capi = pkg.capi


@pytest.fixture(scope="module")
def rr(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available to build the rocRAND helper")
    so = tmp_path_factory.mktemp("rr") / "librrcheck.so"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-w", "-shared", "-fPIC",
                           os.path.join(ROOT, "tests", "rocrand_device_check.hip"), "-o", str(so)])
    L = C.CDLL(str(so))
    L.rr_normal4.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]
    L.rr_normal_double2.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]
    return L


@pytest.fixture(scope="module")
def ctx():
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()          # one explicit stream shared by torch and the library (see test_gpu_parity.py)
    torch.cuda.set_stream(stream)
    c = capi.Context(0, stream.cuda_stream)
    yield c
    c.close()
    torch.cuda.set_stream(torch.cuda.default_stream())


def test_bulk_fill_equals_rocrand_subsequence_zero(ctx, rr):
    blocks = 50_000
    ref32 = torch.empty(blocks * 4, dtype=torch.float32, device="cuda")
    ref64 = torch.empty(blocks * 2, dtype=torch.float64, device="cuda")
    assert rr.rr_normal4(1234, 0, 1, blocks, ref32.data_ptr()) == 0
    assert rr.rr_normal_double2(1234, 0, 1, blocks, ref64.data_ptr()) == 0
    got32, got64 = torch.empty_like(ref32), torch.empty_like(ref64)
    ctx.generate_normals(1234, got32.numel(), capi.F32, got32)
    ctx.generate_normals(1234, got64.numel(), capi.F64, got64)
    assert (got32 - ref32).abs().max().item() < 4e-6
    assert torch.allclose(got64, ref64, rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize("prec", [capi.F32, capi.F64])
def test_path_stream_is_rocrand_subsequence_per_global_path_id(ctx, rr, prec):
    # S0 = 1, r = v^2/2 (zero drift), v sqrt(dt) = 1  ->  S_n(path p) = exp(sum_{i<n} G_i(p)), G from subsequence p
    n_paths, n_steps, offset = 4096, 12, (1 << 32) + 17
    nb = 4 if prec == capi.F32 else 2
    blocks = n_steps // nb
    t = torch.float32 if prec == capi.F32 else torch.float64
    z = torch.empty(n_paths * n_steps, dtype=t, device="cuda")
    fn = rr.rr_normal4 if prec == capi.F32 else rr.rr_normal_double2
    assert fn(777, offset, n_paths, blocks, z.data_ptr()) == 0
    want = torch.exp(torch.cumsum(z.view(n_paths, n_steps).double(), dim=1)).T      # [step][path]
    v = math.sqrt(n_steps)   # T = 1, dt = 1/n_steps, v sqrt(dt) = 1
    opt = capi.make_option(S0=1.0, T=1.0, K=0.5, r=v * v / 2, v=v)
    traj = torch.empty(n_steps * n_paths, dtype=t, device="cuda")
    ctx.simulate_trajectories(opt, capi.make_sim(1 << 40, n_steps, prec, seed=777, path_offset=offset, n_paths_local=n_paths), traj)
    got = traj.view(n_steps, n_paths).double()
    rel = ((got - want) / want).abs().max().item()
    assert rel < (1e-12 if prec == capi.F64 else 5e-5), rel
