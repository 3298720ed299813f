"""ctypes binding of libgsloc_hip.so (C ABI: include/gsloc_hip.h).

There is no CPU fallback: if the shared library is missing or a call returns a
non-zero status the operators raise.  ``build_library()`` drives ``make`` in
``csrc/`` (hipcc, --offload-arch=gfx950); it is what ``__graft_entry__.build``
calls.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsloc_hip.so")
_lib: Optional[ctypes.CDLL] = None

P = c_void_p  # every device pointer travels as void*

_SIGNATURES = {
    "gsl_version": (c_char_p, []),
    "gsl_status_string": (c_char_p, [c_int]),
    "gsl_dev_poison_lds": (c_int, [ctypes.c_uint32, P]),
    "gsl_project_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_float, c_float, c_float, c_float,
                                P, P, P, P, P, P]),
    "gsl_project_bwd_ws_bytes": (c_size_t, [c_int]),
    "gsl_project_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_float, P, P, P, P, P, P, P, P, P, P, P,
                                P, c_size_t, P]),
    "gsl_sh_fwd": (c_int, [c_int, P, P, P, c_int, c_int, P, P]),
    "gsl_sh_bwd": (c_int, [c_int, P, P, P, c_int, c_int, P, P, P, P]),
    "gsl_isect_ws_bytes": (c_size_t, [c_int]),
    "gsl_isect_count": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_size_t, P]),
    "gsl_isect_fill": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int64,
                               P, P, P, P, c_size_t, P]),
    "gsl_tile_sort": (c_int, [P, c_int, c_int, c_int64, P, P, P, c_int64, P]),
    "gsl_fused_ws_bytes": (c_size_t, [c_int, c_int]),
    "gsl_fused_project": (c_int, [P, P, P, P, P, c_int, c_int, P, P, c_int, c_int, c_int, c_float, c_float, c_float,
                                  c_float, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, c_size_t,
                                  P, P, c_int, P, P, P]),
    "gsl_fused_bin": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int64, P, P, P, P, c_size_t, c_int,
                              P, c_int, P, P, c_int, P, P, P]),
    "gsl_long_sort": (c_int, [P, c_int, c_int, c_int, c_int, c_int64, P, c_int, P, P, c_int, P, c_size_t, c_int, c_int, P,
                              P]),
    "gsl_fused_raster_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int64,
                                     P, P, P, c_int, c_int, P, P, P, P, c_int, P, c_int, P, P, P, P]),
    "gsl_fused_raster_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int64,
                                     P, P, P, P, P, P, c_int, c_int, P, P, P, P, c_int, P, P]),
    "gsl_long_ws_bytes": (c_size_t, [c_int]),
    "gsl_long_segment": (c_int, []),
    "gsl_long_sort_segment": (c_int, []),
    "gsl_long_raster_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int64,
                                    P, P, P, c_int, c_int, P, P, c_int, P, c_size_t, c_int, c_int, P]),
    "gsl_long_raster_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int64,
                                    P, P, P, P, P, P, c_int, c_int, P, P, c_int, P, c_int, P]),
    "gsl_loss_ws_bytes": (c_size_t, [c_int, c_int]),
    "gsl_loss_n_partials": (c_int, [c_int, c_int, c_int, c_int]),
    "gsl_tracking_loss": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_float, c_float, P, P, P, P, c_size_t,
                                  P]),
    "gsl_pose_init": (c_int, [P, P, P, c_float, c_float, P, P, P]),
    "gsl_normal_ws_bytes": (c_size_t, [c_int, c_int]),
    "gsl_normal_loss": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float,
                                P, P, P, c_size_t, P]),
    "gsl_fused_viewmat_rows": (P, [P, c_int]),
    "gsl_pose_step": (c_int, [P, P, P, P, c_int, P, P, c_int, P, P, P, c_int, c_int, c_float, c_float, c_float, c_float, c_float,
                              c_float, c_float, c_float, c_float, c_int, c_int, c_int, c_int, P, P, P, P]),
    "gsl_pack_pose_reduce": (c_int, [P, P, c_int, P, P, P, c_int, P, P, P]),
    "gsl_knn_ws_bytes": (c_size_t, [c_int]),
    "gsl_knn_cells": (c_int, []),
    "gsl_knn_count": (c_int, [P, c_int, P, P, c_size_t, P]),
    "gsl_knn_query": (c_int, [P, c_int, P, P, c_int, P, P, c_size_t, P]),
    "gsl_fused_project_bwd": (c_int, [P, P, P, P, P, c_int, c_int, P, P, c_int, c_int, c_int, c_float, c_int, c_int,
                                      P, P, P, P, P, P, P, P, P, P, P, c_size_t, c_int, P, P, P, P, c_int, c_int,
                                      c_int, c_int, c_int64, P, P, c_int, P, P]),
    "gsl_tiny_raster_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int64,
                                    P, P, P, P, P, P, P, c_int, c_int, P, c_int, P, c_float, c_float, P, P, P]),
    "gsl_isect_emit": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "gsl_isect_offsets": (c_int, [P, c_int64, c_int, c_int, c_int, P, P]),
    "gsl_rasterize_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P,
                                  c_int64, P, P, P, P]),
    "gsl_vacc_bytes": (c_size_t, [c_int, c_int]),
    "gsl_rasterize_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P,
                                  c_int64, P, P, P, P, P, P]),
    "gsl_vacc_unpack": (c_int, [P, c_int, c_int, P, P, P, P, P]),
}


def library_path() -> str:
    return _LIB_PATH


def exported_symbols():
    """Names every build of the library must export (== include/gsloc_hip.h)."""
    return sorted(_SIGNATURES)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libgsloc_hip.so (in-tree)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-j", str(min(8, os.cpu_count() or 1)), "all"]
    if force:
        subprocess.run(["make", "-C", csrc, "clean"], check=True, capture_output=not verbose)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("building libgsloc_hip.so failed:\n" + res.stderr[-4000:])
    return _LIB_PATH


def load_library() -> ctypes.CDLL:
    """Load (once) and type the C ABI.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"{_LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C gsplatloc_amd/csrc`."
        )
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load_library().gsl_status_string(status).decode()
        raise RuntimeError(f"{what} failed: {msg} (status {status})")


def ptr(t) -> Optional[int]:
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream() -> int:
    """Raw handle of torch's current HIP stream (the private getter when there is one: a tenth of the cost of building
    a torch.cuda.Stream object, and this is called once per launch)."""
    import torch

    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return raw(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
