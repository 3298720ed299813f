"""``rasterization`` -- the one gsplat entry point GsplatLoc calls.

Call sites replaced: /root/reference/src/my_gsplat/model.py:195-213 (grad,
"RGB+ED") and /root/reference/src/my_gsplat/geometry.py:117-132 (no_grad,
"ED").  Signature, defaults, return triple and meta keys follow
``gsplat.rendering.rasterization`` of gsplat 1.3.0 (IDX:14954); the stages run
on the HIP kernels of libgsloc_hip (see ``ops.py``).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor
from typing_extensions import Literal

import os

from .fused import cached_rasterization, fused_rasterization, fused_supported
from .ops import (
    fully_fused_projection,
    isect_offset_encode,
    isect_tiles,
    rasterize_to_pixels,
    spherical_harmonics,
)


def rasterization(
    means: Tensor,  # [N, 3]
    quats: Tensor,  # [N, 4]
    scales: Tensor,  # [N, 3]
    opacities: Tensor,  # [N]
    colors: Tensor,  # [(C,) N, D] or [(C,) N, K, 3]
    viewmats: Tensor,  # [C, 4, 4]
    Ks: Tensor,  # [C, 3, 3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = True,
    tile_size: int = 16,
    backgrounds: Optional[Tensor] = None,
    render_mode: Literal["RGB", "D", "ED", "RGB+D", "RGB+ED"] = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: Literal["classic", "antialiased"] = "classic",
    channel_chunk: int = 32,
    distributed: bool = False,
    ortho: bool = False,
    covars: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor, Dict]:
    """Rasterize a set of 3D Gaussians to a batch of C image planes.

    Returns (render_colors [C,H,W,X], render_alphas [C,H,W,1], meta).  X is D for
    "RGB", 1 for "D"/"ED", D+1 for "RGB+D"/"RGB+ED"; "ED" divides the accumulated
    depth by alpha ("expected depth").

    ``packed`` only changes gsplat's internal memory layout, never the rendered
    result; this implementation always computes densely and returns the dense
    ([C,N,...]) meta tensors.
    """
    meta: Dict = {}
    N = means.shape[0]
    C = viewmats.shape[0]
    assert means.shape == (N, 3), means.shape
    assert quats.shape == (N, 4), quats.shape
    assert scales.shape == (N, 3), scales.shape
    assert opacities.shape == (N,), opacities.shape
    assert viewmats.shape == (C, 4, 4), viewmats.shape
    assert Ks.shape == (C, 3, 3), Ks.shape
    assert render_mode in ["RGB", "D", "ED", "RGB+D", "RGB+ED"], render_mode
    if covars is not None:
        raise NotImplementedError("covars input is not supported; pass quats and scales")
    if distributed:
        raise NotImplementedError("gsplat's Gaussian-sharded distributed mode is not part of GsplatLoc's path; "
                                  "use gsplatloc_amd.parallel for screen-tile parallelism")
    if ortho:
        raise NotImplementedError("orthographic cameras are not supported")
    if absgrad:
        raise NotImplementedError("absgrad is not supported (GsplatLoc: absgrad=False, model.py:124)")
    if sparse_grad:
        raise NotImplementedError("sparse_grad is not supported (GsplatLoc: sparse_grad=False, model.py:122)")

    if sh_degree is None:  # colours are final values: [N, D] shared by the cameras, or [C, N, D]
        assert (colors.dim() == 2 and colors.shape[0] == N) or (
            colors.dim() == 3 and colors.shape[:2] == (C, N)), colors.shape
    else:  # colours are SH coefficients with K >= (sh_degree + 1)^2 bands: [N, K, 3] or [C, N, K, 3]
        assert (colors.dim() == 3 and colors.shape[0] == N and colors.shape[2] == 3) or (
            colors.dim() == 4 and colors.shape[:2] == (C, N) and colors.shape[3] == 3), colors.shape
        assert (sh_degree + 1) ** 2 <= colors.shape[-2], colors.shape

    # Hot path: one camera, no background -> the fused five-launch pipeline (csrc/fused.hip).
    if os.environ.get("GSLOC_DISABLE_FUSED", "0") != "1" and fused_supported(
            N, C, colors, sh_degree, width, height, tile_size, backgrounds, render_mode):
        # the same call signature again and again (a tracker's loop): keep the context, see fused.py
        call = cached_rasterization if (N > 0 and os.environ.get("GSLOC_DROPIN_CACHE", "1") != "0") else fused_rasterization
        render, alphas, meta = call(
            means, quats, scales, opacities, colors, viewmats[0], Ks[0], width, height, sh_degree=sh_degree,
            render_mode=render_mode, eps2d=eps2d, near_plane=near_plane, far_plane=far_plane,
            radius_clip=radius_clip, antialiased=(rasterize_mode == "antialiased"))
        return render[None], alphas[None], meta

    # General path (several cameras, a background colour): the stage operators, one after the other.
    # Features wider than ``channel_chunk`` (gsplat's default 32, the widest compositing kernel here) are composited in
    # chunks of that many channels, as gsplat does; every chunk yields the same alphas, the first one's are returned.
    want_depth = render_mode in ("D", "ED", "RGB+D", "RGB+ED")
    want_rgb = render_mode.startswith("RGB")
    radii, means2d, depths, conics, compensations = fully_fused_projection(
        means, None, quats, scales, viewmats, Ks, width, height, eps2d=eps2d, packed=False, near_plane=near_plane,
        far_plane=far_plane, radius_clip=radius_clip, sparse_grad=False,
        calc_compensations=(rasterize_mode == "antialiased"))
    opac = opacities[None].expand(C, N)
    if compensations is not None:  # "antialiased": opacity scaled by sqrt(det Sigma / det(Sigma + eps2d I))
        opac = opac * compensations
    opac = opac.contiguous()

    # per-camera features [C, N, D]: RGB (direct, or SH evaluated along the view direction, shifted by 0.5 and
    # clamped at 0 as the CUDA backends do), then the camera-space depth as the last channel
    feats = []
    if want_rgb:
        if sh_degree is None:
            rgb = colors if colors.dim() == 3 else colors[None].expand(C, N, colors.shape[-1])
        else:
            cam_pos = torch.linalg.inv(viewmats)[:, :3, 3]                      # [C, 3]
            view_dirs = means[None] - cam_pos[:, None]                          # [C, N, 3]
            coeffs = colors if colors.dim() == 4 else colors[None].expand(C, *colors.shape)
            rgb = (spherical_harmonics(sh_degree, view_dirs, coeffs, masks=radii > 0) + 0.5).clamp_min(0.0)
        feats.append(rgb)
    if want_depth:
        feats.append(depths[..., None])
    feats = torch.cat(feats, dim=-1) if len(feats) > 1 else feats[0]
    bg = backgrounds
    if bg is not None and want_depth:  # the depth channel composites over 0
        bg = torch.cat([bg, bg.new_zeros(C, 1)], dim=-1) if want_rgb else bg.new_zeros(C, 1)

    tile_width, tile_height = -(-width // tile_size), -(-height // tile_size)
    tiles_per_gauss, isect_ids, flatten_ids = isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height,
                                                          packed=False, n_cameras=C)
    isect_offsets = isect_offset_encode(isect_ids, C, tile_width, tile_height)
    chunk = min(int(channel_chunk), 32)
    assert chunk >= 1, channel_chunk
    if feats.shape[-1] > chunk:
        parts, render_alphas = [], None
        for lo in range(0, feats.shape[-1], chunk):
            part, part_alphas = rasterize_to_pixels(means2d, conics, feats[..., lo:lo + chunk].contiguous(), opac, width,
                                                    height, tile_size, isect_offsets, flatten_ids,
                                                    backgrounds=None if bg is None else bg[..., lo:lo + chunk].contiguous())
            parts.append(part)
            render_alphas = part_alphas if render_alphas is None else render_alphas
        render_colors = torch.cat(parts, dim=-1)
    else:
        render_colors, render_alphas = rasterize_to_pixels(means2d, conics, feats, opac, width, height, tile_size,
                                                           isect_offsets, flatten_ids, backgrounds=bg)
    if render_mode in ("ED", "RGB+ED"):  # expected depth: accumulated depth over accumulated alpha
        expected = render_colors[..., -1:] / render_alphas.clamp(min=1e-10)
        render_colors = torch.cat([render_colors[..., :-1], expected], dim=-1)

    meta.update({
        "camera_ids": None, "gaussian_ids": None, "radii": radii, "means2d": means2d, "depths": depths,
        "conics": conics, "opacities": opac, "tile_width": tile_width, "tile_height": tile_height,
        "tiles_per_gauss": tiles_per_gauss, "isect_ids": isect_ids, "flatten_ids": flatten_ids,
        "isect_offsets": isect_offsets, "width": width, "height": height, "tile_size": tile_size, "n_cameras": C,
    })
    return render_colors, render_alphas, meta
