"""vofod_amd — MI355X-native implementation of VoFOD's per-scan point-cloud hot path.

The compute lives in `csrc/` (hand-written gfx950 HIP kernels + C++ host driver behind
the C-ABI of include/vofod.h, built into `csrc/libvofod_hip.so`).  This package is the
host-side mirror of the nodelet interface on top of that library.  There is no CPU
fallback: `library()` raises if the HIP extension has not been built.
"""
from __future__ import annotations

from pathlib import Path

from . import capi
from .detector import ScanData, VoFOD, VofodError, default_params  # noqa: F401

LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libvofod_hip.so"
_lib: capi.Library | None = None


def library() -> capi.Library:
    """The product library (HIP).  Raises ImportError when it is not built/loadable."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). vofod_amd has no CPU fallback."
            )
        _lib = capi.Library(LIB_PATH, "vofod_")
    return _lib
