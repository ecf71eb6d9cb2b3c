"""MI355X-native Monte Carlo option-pricing engine: HIP kernels for gfx950 behind a C ABI
(include/mcamd.h), with the reference's call surface re-exposed in C++ (include/*.hpp).

The directory name carries a hyphen, so import it with
    importlib.import_module("monte-carlo-project-cuda_amd")
(see __graft_entry__.py).  `capi` is the ctypes binding of the C ABI; `build` compiles it.
"""
from . import build as _build  # noqa: F401
from . import capi  # noqa: F401
from . import sharding  # noqa: F401

build = _build.build
