"""Legacy (gsplat v0.1-style) operator pair kept for signature compatibility.

``project_gaussians`` (IDX:14774) and ``rasterize_gaussians`` (IDX:14765) of
``gsplat.cuda_legacy._wrapper`` in the fork GsplatLoc installs; the north-star
contract asks for these names.  GsplatLoc itself calls ``rasterization``
(/root/reference/src/my_gsplat/model.py:195-213), so the arithmetic here follows the v1
kernels that call executes (SURVEY.md decision D1): v1 tile rectangle, radius floor
0.01, exact view-matrix gradient.  Single camera, 16x16 tiles.
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
from torch import Tensor

from .ops import (
    fully_fused_projection,
    isect_offset_encode,
    isect_tiles,
    rasterize_to_pixels,
)


def _cov3d_triu(scales: Tensor, glob_scale: float, quats: Tensor) -> Tensor:
    q = torch.nn.functional.normalize(quats, dim=-1)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)
    M = R * (glob_scale * scales)[:, None, :]
    S = M @ M.transpose(-1, -2)
    i, j = torch.triu_indices(3, 3)
    return S[:, i, j]


def project_gaussians(
    means3d: Tensor,  # [N, 3]
    scales: Tensor,  # [N, 3]
    glob_scale: float,
    quats: Tensor,  # [N, 4] wxyz
    viewmat: Tensor,  # [4, 4] world -> camera
    fx: float,
    fy: float,
    cx: float,
    cy: float,
    img_height: int,
    img_width: int,
    block_width: int,
    clip_thresh: float = 0.01,
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """Returns (xys [N,2], depths [N], radii [N] i32, conics [N,3], compensation [N],
    num_tiles_hit [N] i32, cov3d [N,6])."""
    assert block_width == 16, "block_width must be 16 (the kernels' tile size)"
    assert means3d.shape[-1] == 3 and scales.shape[-1] == 3 and quats.shape[-1] == 4
    assert viewmat.shape[-2:] == (4, 4) or viewmat.shape[-2:] == (3, 4), viewmat.shape
    if viewmat.shape[-2] == 3:
        viewmat = torch.cat([viewmat, torch.tensor([[0, 0, 0, 1.0]], device=viewmat.device)], dim=0)
    K = torch.tensor([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]], dtype=torch.float32, device=means3d.device)
    radii, xys, depths, conics, comp = fully_fused_projection(
        means3d, None, quats, scales * glob_scale, viewmat[None], K[None], img_width, img_height,
        near_plane=clip_thresh, calc_compensations=True)
    tw = (img_width + block_width - 1) // block_width
    th = (img_height + block_width - 1) // block_width
    num_tiles_hit, _, _ = isect_tiles(xys.detach(), radii, depths.detach(), block_width, tw, th, sort=False)
    with torch.no_grad():
        cov3d = _cov3d_triu(scales, glob_scale, quats)
    return xys[0], depths[0], radii[0], conics[0], comp[0], num_tiles_hit[0], cov3d


def rasterize_gaussians(
    xys: Tensor,  # [N, 2]
    depths: Tensor,  # [N]
    radii: Tensor,  # [N]
    conics: Tensor,  # [N, 3]
    num_tiles_hit: Tensor,  # [N]
    colors: Tensor,  # [N, D]
    opacity: Tensor,  # [N, 1]
    img_height: int,
    img_width: int,
    block_width: int,
    background: Optional[Tensor] = None,
    return_alpha: Optional[bool] = False,
) -> Union[Tensor, Tuple[Tensor, Tensor]]:
    """Returns out_img [H,W,D] (and out_alpha [H,W] when return_alpha)."""
    assert block_width == 16, "block_width must be 16 (the kernels' tile size)"
    assert xys.ndim == 2 and xys.shape[1] == 2, xys.shape
    assert colors.ndim == 2 and colors.shape[0] == xys.shape[0], colors.shape
    if colors.dtype == torch.uint8:
        colors = colors.float() / 255
    if opacity.ndim == 2:
        opacity = opacity[:, 0]
    if background is not None:
        assert background.shape[0] == colors.shape[-1], f"incorrect shape of background color tensor"
        background = background[None]
    tw = (img_width + block_width - 1) // block_width
    th = (img_height + block_width - 1) // block_width
    _, isect_ids, flatten_ids = isect_tiles(xys[None].detach(), radii[None], depths[None].detach(), block_width, tw,
                                            th, sort=True)
    offsets = isect_offset_encode(isect_ids, 1, tw, th)
    out, alpha = rasterize_to_pixels(xys[None], conics[None], colors[None], opacity[None], img_width, img_height,
                                     block_width, offsets, flatten_ids, backgrounds=background)
    if return_alpha:
        return out[0], alpha[0, ..., 0]
    return out[0]
