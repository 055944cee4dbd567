"""Build and load the C-ABI shared library (guassianhand_amd/libgh_raster.so).

The library is built in-tree with hipcc for gfx950 and loaded with ctypes. There is NO CPU fallback:
if the library is missing or fails to load, every rasteriser entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
from typing import Optional

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libgh_raster.so")
# Profiling / ablation tooling only: load another build of the same C-ABI (tools/abl/*.so). Still a HIP library — there
# is no CPU path behind this switch either.
_OVERRIDE = os.environ.get("GH_RASTER_LIB")
SOURCES = ("gh_api.hip", "gh_preprocess.hip", "gh_binning.hip", "gh_render.hip", "gh_uv.hip", "gh_sh.hip", "gh_knn.hip", "gh_loss.hip", "gh_select.hip")
HEADERS = ("gh_internal.h", os.path.join("..", "..", "include", "gh_raster.h"))
# -ffp-contract=off: FMAs only where the source says fmaf() (arithmetic contract)
# -fno-slp-vectorize: keeps the DPP butterflies as v_add_f32_dpp instead of v_mov_dpp + v_pk_add_f32
HIPCC_FLAGS = ("--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-std=c++17")

_lib: Optional[C.CDLL] = None


class GhLibraryError(RuntimeError):
    pass


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise GhLibraryError("hipcc not found: cannot build libgh_raster.so")
    return exe


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into libgh_raster.so (cross-compiles without a GPU)."""
    if not force and not is_stale():
        return LIB_PATH
    cmd = [_hipcc(), *HIPCC_FLAGS, "-o", LIB_PATH + ".tmp", *[os.path.join(CSRC, s) for s in SOURCES]]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(" ".join(cmd))
        print(res.stdout, res.stderr)
    if res.returncode != 0:
        raise GhLibraryError(f"hipcc failed ({res.returncode}):\n{res.stderr[-4000:]}")
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


def lib() -> C.CDLL:
    """The loaded library. torch must be imported first so both share one HIP runtime
    (torch's bundled libamdhip64.so and /opt/rocm's carry the same SONAME)."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (loads libamdhip64.so.7 into the process before our library binds to it)
        path = _OVERRIDE or LIB_PATH
        if not os.path.exists(path):
            raise GhLibraryError(
                f"{path} is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the rasteriser)")
        try:
            L = C.CDLL(path)
        except OSError as e:  # pragma: no cover
            raise GhLibraryError(f"cannot load {path}: {e}") from e
        _abi.declare(L)
        # the structs of the C-ABI grow between 0.x minor versions: a library built from another header (GH_RASTER_LIB pointing at
        # a build of another commit, a stale .so) would read this mirror's structs with the wrong layout
        want = (_abi.GH_VERSION_MAJOR << 16) | _abi.GH_VERSION_MINOR
        have = int(L.gh_version())
        if have != want:
            raise GhLibraryError(f"{path} is C-ABI v{have >> 16}.{have & 0xFFFF}, this package speaks v{want >> 16}.{want & 0xFFFF}: "
                                 "rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
        _lib = L
    return _lib


def loaded_path() -> Optional[str]:
    return (_OVERRIDE or LIB_PATH) if _lib is not None else None
