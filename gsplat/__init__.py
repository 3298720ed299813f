"""Drop-in ``gsplat`` namespace for GsplatLoc on MI355X.

``from gsplat import rasterization`` (/root/reference/src/my_gsplat/model.py:5,
/root/reference/src/my_gsplat/geometry.py:4) resolves to the HIP-backed operators of
``gsplatloc_amd`` when this repository is on ``sys.path``.
"""
from gsplatloc_amd.legacy import project_gaussians, rasterize_gaussians  # noqa: F401
from gsplatloc_amd.ops import (  # noqa: F401
    fully_fused_projection,
    isect_offset_encode,
    isect_tiles,
    rasterize_to_pixels,
    spherical_harmonics,
)
from gsplatloc_amd.rendering import rasterization  # noqa: F401

__version__ = "1.3.0+gsloc.hip"
