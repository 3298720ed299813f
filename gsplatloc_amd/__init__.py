"""gsplatloc_amd -- MI355X-native Gaussian-splat rasterizer for GsplatLoc pose tracking.

Host side: PyTorch-ROCm tensors and autograd glue (this package).
Device side: hand-written HIP kernels for gfx950 behind the C ABI of
``include/gsloc_hip.h`` (``libgsloc_hip.so``, loaded with ctypes).

The public operator surface mirrors the gsplat API that GsplatLoc's
``src/my_gsplat`` calls (/root/reference/src/my_gsplat/model.py:195-213):
``rasterization`` plus the stage operators and the legacy pair
``project_gaussians`` / ``rasterize_gaussians``.  The top-level ``gsplat``
package of this repository re-exports them so ``from gsplat import
rasterization`` resolves here.
"""
from ._lib import build_library, library_path, load_library  # noqa: F401
from .ops import (  # noqa: F401
    fully_fused_projection,
    isect_offset_encode,
    isect_tiles,
    rasterize_to_pixels,
    spherical_harmonics,
)
from .rendering import rasterization  # noqa: F401
from .legacy import project_gaussians, rasterize_gaussians  # noqa: F401

__version__ = "0.1.0"
