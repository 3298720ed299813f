"""ctypes front end of the C restatement (oracle/csrc/gsplat_oracle.c).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``); PARITY UNPINNED like the rest of the oracle.
Two builds of the same source: ``f64`` (the checker, pinned against the autograd oracle in
tests/test_c_oracle.py) and ``f32`` (the CPU baseline bench.py times, same arithmetic type as the HIP path).
``build()`` compiles both with gcc + OpenMP into oracle/_build/ (git-ignored, travels to the GPU box).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_double, c_float, c_int, c_int64, c_void_p
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "gsplat_oracle.c")
_OUT = os.path.join(_HERE, "_build")
_LIBS: Dict[str, ctypes.CDLL] = {}
_MODES = {"RGB": (True, 0), "D": (False, 1), "ED": (False, 2), "RGB+D": (True, 1), "RGB+ED": (True, 2)}


def library_path(precision: str) -> str:
    assert precision in ("f32", "f64")
    return os.path.join(_OUT, f"libgso_{precision}.so")


def build(force: bool = False) -> None:
    os.makedirs(_OUT, exist_ok=True)
    for precision, ctype in (("f64", "double"), ("f32", "float")):
        out = library_path(precision)
        if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(_SRC):
            continue
        cmd = ["gcc", "-O3", "-std=c99", "-fPIC", "-fopenmp", "-shared", "-Wall", "-Wextra", f"-DGSO_REAL={ctype}",
               _SRC, "-o", out, "-lm"]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("building the C oracle failed:\n" + res.stderr[-4000:])


def load(precision: str = "f64") -> ctypes.CDLL:
    if precision not in _LIBS:
        path = library_path(precision)
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(_SRC):
            build()
        lib = ctypes.CDLL(path)
        real = c_double if precision == "f64" else c_float
        assert lib.gso_real_bytes() == ctypes.sizeof(real)
        P = c_void_p
        lib.gso_set_threads.argtypes = [c_int]
        lib.gso_project_fwd.argtypes = [P, P, P, P, P, c_int, c_int, c_int, real, real, real, real, P, P, P, P, P]
        lib.gso_project_bwd.argtypes = [P, P, P, P, P, c_int, c_int, c_int, real, P, P, P, P, P, P, P, P, P]
        lib.gso_sh_fwd.argtypes = [c_int, P, P, P, c_int, c_int, P]
        lib.gso_sh_bwd.argtypes = [c_int, P, P, P, c_int, c_int, P, P, P]
        lib.gso_isect.restype = c_int64
        lib.gso_isect.argtypes = [P, P, P, c_int, c_int, c_int, c_int, P, c_int64, P, P, P]
        lib.gso_raster_fwd.argtypes = [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int64, P, P, P]
        lib.gso_raster_bwd.argtypes = [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int64, P, P,
                                       P, P, P, P, P, P]
        lib.gso_rasterization.argtypes = [P, P, P, P, P, c_int, c_int, c_int, P, P, c_int, c_int, c_int, c_int, real, real,
                                          real, real, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P]
        _LIBS[precision] = lib
    return _LIBS[precision]


def _arr(x, dtype) -> Optional[np.ndarray]:
    if x is None:
        return None
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=dtype)


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


def rasterization(means, quats, scales, opacities, colors, viewmat, K, width: int, height: int,
                  sh_degree: Optional[int] = None, render_mode: str = "RGB+ED", v_render=None, v_alphas=None,
                  near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0, eps2d: float = 0.3,
                  tile_size: int = 16, antialiased: bool = False, precision: str = "f64",
                  threads: Optional[int] = None) -> Dict[str, np.ndarray]:
    """gsplat.rasterization for one camera (viewmat [4,4], K [3,3]); with ``v_render`` [H,W,D] (and optionally
    ``v_alphas`` [H,W]) also the backward.  Returns numpy arrays: render, alphas, n_isects and, after a backward,
    v_means, v_quats, v_scales, v_opacities, v_colors, v_viewmat."""
    lib = load(precision)
    dt = np.float64 if precision == "f64" else np.float32
    if threads:
        lib.gso_set_threads(int(threads))
    want_rgb, depth_mode = _MODES[render_mode]
    means, quats, scales = _arr(means, dt), _arr(quats, dt), _arr(scales, dt)
    opacities, viewmat, K = _arr(opacities, dt), _arr(viewmat, dt).reshape(4, 4), _arr(K, dt).reshape(3, 3)
    N = means.shape[0]
    color_mode, K_sh, deg = 0, 0, 0
    cols = None
    if want_rgb:
        cols = _arr(colors, dt)
        if sh_degree is None:
            color_mode = 2
            assert cols.shape == (N, 3)
        else:
            color_mode, deg, K_sh = 1, int(sh_degree), cols.shape[1]
            assert cols.shape == (N, K_sh, 3)
    D = (3 if want_rgb else 0) + (1 if depth_mode else 0)
    render = np.zeros((height, width, D), dtype=dt)
    alphas = np.zeros((height, width), dtype=dt)
    n_is = c_int64(0)
    back = v_render is not None
    vr = _arr(v_render, dt).reshape(height, width, D) if back else None
    va = (_arr(v_alphas, dt).reshape(height, width) if v_alphas is not None else np.zeros((height, width), dtype=dt)) if back else None
    out = {}
    if back:
        out = {"v_means": np.zeros((N, 3), dt), "v_quats": np.zeros((N, 4), dt), "v_scales": np.zeros((N, 3), dt),
               "v_opacities": np.zeros(N, dt), "v_viewmat": np.zeros((4, 4), dt),
               "v_colors": (np.zeros_like(cols) if want_rgb else None)}
    rc = lib.gso_rasterization(
        _p(means), _p(quats), _p(scales), _p(opacities), _p(cols), color_mode, deg, K_sh, _p(viewmat), _p(K), N, width,
        height, tile_size, eps2d, near_plane, far_plane, radius_clip, int(antialiased), depth_mode, int(back), _p(vr),
        _p(va), _p(render), _p(alphas), _p(out.get("v_means")), _p(out.get("v_quats")), _p(out.get("v_scales")),
        _p(out.get("v_opacities")), _p(out.get("v_colors")), _p(out.get("v_viewmat")), ctypes.byref(n_is))
    if rc != 0:
        raise RuntimeError(f"gso_rasterization failed ({rc})")
    out.update(render=render, alphas=alphas, n_isects=int(n_is.value))
    return out
