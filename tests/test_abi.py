"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports exactly
the symbols include/gsloc_hip.h declares; argument validation returns error codes
without touching a GPU."""
import os
import re

from gsplatloc_amd import _lib


def _declared(repo_root):
    txt = open(os.path.join(repo_root, "include", "gsloc_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gsl_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_agree(repo_root):
    assert _declared(repo_root) == _lib.exported_symbols()


def test_headers_are_plain_c(repo_root):
    """The boundary is a C ABI: both headers must compile as C99 without any HIP / C++ / torch type."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    for h in ("gsloc_hip.h", "gsloc_icp.h"):
        res = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                              os.path.join(repo_root, "include", h)], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr


def test_library_exports_every_symbol(repo_root):
    if not os.path.exists(_lib.library_path()):
        _lib.build_library()
    lib = _lib.load_library()
    for name in _declared(repo_root):
        assert hasattr(lib, name), name
    assert lib.gsl_version().decode().startswith("gsloc_hip")
    assert lib.gsl_status_string(-2).decode() == "workspace too small"


def test_bad_arguments_are_rejected_without_a_gpu():
    lib = _lib.load_library()
    # null pointers / bad sizes are caught on the host before any launch
    assert lib.gsl_project_fwd(None, None, None, None, None, 5, 64, 48, 0.3, 0.01, 1e10, 0.0, None, None, None, None,
                               None, None) == -1
    assert lib.gsl_project_fwd(None, None, None, None, None, 5, 0, 48, 0.3, 0.01, 1e10, 0.0, None, None, None, None,
                               None, None) == -1
    assert lib.gsl_rasterize_fwd(None, None, None, None, None, 7, 64, 48, 16, 4, 3, 0, 3, None, None, 0, None, None,
                                 None, None) == -1
    assert lib.gsl_isect_count(None, None, 0, 16, 4, 3, 0, 4, None, None, None, None, 0, None) == -1  # ty1 > tile_h
    assert lib.gsl_sh_fwd(4, None, None, None, 1, 25, None, None) == -1
    # more Gaussians than the packed gradient rows can address with 32-bit byte offsets (include/gsloc_hip.h, "Limits"):
    # means, quats, scales, opacities, colors, sh_degree, K_sh, viewmat, K, N, width, height, eps2d, near, far,
    # radius_clip, antialiased, tile_w, tile_h, ty0, ty1, then 17 pointers / sizes
    assert lib.gsl_fused_project(None, None, None, None, None, 1, 4, None, None, (1 << 26) + 1, 64, 48, 0.3, 0.01, 1e10,
                                 0.0, 0, 4, 3, 0, 3, None, None, None, None, None, None, None, None, None, 0, None, None,
                                 0, None, None, None) == -1
    assert lib.gsl_project_bwd_ws_bytes(1000) == 4 * 12 * 4
    assert lib.gsl_isect_ws_bytes(100) == 800


def test_ops_fail_loudly_without_gpu_tensors():
    import pytest
    import torch

    import gsplatloc_amd as A

    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    m = torch.zeros(4, 3)
    with pytest.raises(AssertionError, match="no CPU path"):
        A.fully_fused_projection(m, None, torch.zeros(4, 4), torch.zeros(4, 3), torch.eye(4)[None], torch.eye(3)[None],
                                 32, 32)
