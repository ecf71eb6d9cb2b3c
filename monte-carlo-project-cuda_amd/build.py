"""Builds libmcamd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the
GPU box with the repository snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libmcamd.so")
SOURCES = ["price_f64.hip", "price_f32.hip", "store.hip", "aux.hip", "nmc.hip", "capi.cpp", "group.cpp"]
HEADERS = ["launch.hpp", "mc_device.hpp", "path_consts.hpp", "fast64.hpp", "tables64.inc", "tables64_consts.inc",
           "price_impl.hpp", "nmc_compact.hpp"]
ARCH = "gfx950"


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the engine is HIP-only and cannot be built without ROCm")
    return exe


def _flags(extra=()):
    return ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
            "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, *extra]


def build_id(extra=()) -> str:
    """Hash of everything the device code is made from (kernel sources, headers, the public header, the compile
    flags).  Compiled into the library (mcamd_build_id) and written beside the ISA slot counts
    (profiles/valu_slots.json), so bench.py can tell whether the counts describe the kernels it is running."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    with open(os.path.join(ROOT, "include", "mcamd.h"), "rb") as f:
        h.update(f.read())
    # every flag that can change the device code (include paths are machine-dependent and carry no code)
    h.update(" ".join(f for f in _flags(extra) if not f.startswith("-I")).encode())
    return h.hexdigest()[:16]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, force: bool, extra, obj_dir: str = OBJ) -> str:
    obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS] + [
        os.path.join(ROOT, "include", "mcamd.h"), os.path.abspath(__file__)]
    if src == "capi.cpp":   # carries the build id, a hash over EVERY source
        deps += [os.path.join(CSRC, s_) for s_ in SOURCES]
    if force or _stale(obj, deps):
        cmd = [hipcc(), *_flags(extra), "-c", os.path.join(CSRC, src), "-o", obj]
        if src.endswith(".cpp"):
            cmd[1:1] = ["-x", "hip", "-ffp-contract=off"]
        if src == "capi.cpp":
            cmd.append('-DMCAMD_BUILD_ID="' + build_id(extra) + '"')
        subprocess.check_call(cmd)
    return obj


def build(force: bool = False, extra_flags=(), jobs: int = 6) -> str:
    os.makedirs(OBJ, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, tuple(extra_flags)), SOURCES))
    if force or _stale(LIB, objs):
        subprocess.check_call([hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", *objs, "-o", LIB, "-ldl"])
    refresh_slot_counts(tuple(extra_flags))
    return LIB


def build_variant(tag: str, extra_flags, jobs: int = 6) -> str:
    """A second library beside the shipped one, built from the same sources with extra -D flags:
    monte-carlo-project-cuda_amd/libmcamd_<tag>.so (objects under csrc/build/<tag>/).  For same-box A/B measurements of
    a source-level variant (tools/ab_lib.py loads it through MCAMD_LIB); never loaded by the package itself, and its
    mcamd_build_id differs from the shipped library's because the flags are part of the id."""
    obj_dir = os.path.join(OBJ, tag)
    os.makedirs(obj_dir, exist_ok=True)
    lib = os.path.join(PKG, f"libmcamd_{tag}.so")
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, False, tuple(extra_flags), obj_dir), SOURCES))
    if _stale(lib, objs):
        subprocess.check_call([hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", *objs, "-o", lib, "-ldl"])
    return lib


def refresh_slot_counts(extra=()) -> bool:
    """Recounts the VALU issue slots of the shipped inner loops (tools/count_valu_slots.py -> profiles/valu_slots.json)
    whenever the committed counts were taken from other sources than the library just built.  Best effort: the tool
    is a heuristic reader of the compiler's assembly, and a failure there must not fail a build whose library has
    already linked — the counts then keep their old build id, which bench.py detects (roofline.frac null, both ids
    reported).  Returns whether the counts describe this build."""
    import json
    import sys
    path = os.path.join(ROOT, "profiles", "valu_slots.json")
    try:
        with open(path) as f:
            if json.load(f).get("build_id") == build_id(extra):
                return True
    except (OSError, ValueError):
        pass
    cmd = [sys.executable, os.path.join(ROOT, "tools", "count_valu_slots.py"), "--hipcc", hipcc()]
    for f in extra:
        cmd += ["--extra-flag=" + f]
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
        return True
    except (subprocess.CalledProcessError, OSError) as e:
        print(f"build.py: warning: VALU slot recount failed ({e}); profiles/valu_slots.json keeps its old build id and "
              "bench.py will report roofline.frac as null", file=sys.stderr)
        return False


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv))
