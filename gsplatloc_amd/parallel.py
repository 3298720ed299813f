"""Screen-tile parallelism across the GPUs of one node.

The render path shards by tile rows: pixels are independent in the forward pass and the pose
gradient is a sum over pixels, so rank r renders and back-propagates only tile rows
[ty0_r, ty1_r) and ONE all-reduce (sum) of 16 floats per iteration -- the 12 entries of
d loss / d viewmat plus loss scalars -- rebuilds the exact gradient on every rank (RCCL over xGMI;
64 bytes, latency-bound).  Every rank then applies the same tiny pose update, so no broadcast
is needed.  Nothing like this exists in the reference (single process, single GPU:
/root/reference/src/my_gsplat/gs_trainer_total.py:45-282).

Gaussians are pre-bucketed once per frame: a rank keeps only the Gaussians whose tile-row range
(plus a guard band) can reach its strip, so projection and binning shrink with the strip as well.
The pose moves by far less than a tile during a frame's optimisation; ``guard_tiles`` covers it.
"""
from __future__ import annotations

from typing import Optional, List, Sequence, Tuple

import torch
from torch import Tensor


def strip_rows(tile_offsets: Tensor, tile_w: int, tile_h: int, world: int) -> List[Tuple[int, int]]:
    """Split tile rows into `world` contiguous strips with balanced intersection counts.
    tile_offsets: [tile_w*tile_h + 1] exclusive scan from a full-frame binning pass."""
    offs = tile_offsets.detach().to("cpu", torch.int64)
    row_start = offs[0:tile_w * tile_h + 1:tile_w].contiguous()  # offset at the start of each tile row (+ total)
    total = int(row_start[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        # first row whose start offset reaches the target, keeping at least one row per strip when possible
        idx = int(torch.searchsorted(row_start, torch.tensor(target, dtype=torch.float64).to(torch.int64), right=False))
        idx = max(idx, bounds[-1] + 1) if bounds[-1] + 1 <= tile_h else bounds[-1]
        bounds.append(min(idx, tile_h))
    bounds.append(tile_h)
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return [(bounds[i], bounds[i + 1]) for i in range(world)]


def gaussians_for_strip(means2d: Tensor, radii: Tensor, rows: Tuple[int, int], tile_size: int = 16,
                        guard_tiles: int = 1) -> Tensor:
    """Indices of the Gaussians whose splat (centre +- radius) can touch tile rows
    [rows[0]-guard, rows[1]+guard).  means2d [N,2], radii [N] from a projection at the frame's
    initial pose."""
    y = means2d[:, 1]
    r = radii.to(means2d.dtype)
    lo = (rows[0] - guard_tiles) * tile_size
    hi = (rows[1] + guard_tiles) * tile_size
    keep = (radii > 0) & (y + r >= lo) & (y - r < hi)
    return keep.nonzero(as_tuple=True)[0]


def all_reduce_pose(buf16: Tensor, group=None) -> Tensor:
    """Sum the 16-float pose-gradient / loss buffer over the ranks (in place)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf16, op=dist.ReduceOp.SUM, group=group)
    return buf16


def halo_rows(rows: Tuple[int, int], tile_h: int) -> Tuple[int, int]:
    """Tile rows a rank must BIN so that the 3x3 Sobel of its own pixel rows is exact: its strip plus the tile
    row that holds the one halo pixel row on each interior side.  Only that pixel row of the extra tile row is
    composited (see halo_pixel_rows)."""
    return max(rows[0] - 1, 0), min(rows[1] + 1, tile_h)


def halo_pixel_rows(rows: Tuple[int, int], height: int, tile_size: int = 16) -> Tuple[int, int]:
    """Pixel rows a rank composites and back-propagates: its own rows plus ONE pixel row of halo on each interior
    side (the support of the 3x3 Sobel, /root/reference/src/my_gsplat/loss.py:51-52)."""
    r0, r1 = rows[0] * tile_size, min(rows[1] * tile_size, height)
    return max(r0 - 1, 0), min(r1 + 1, height)


def strip_tracking_loss(depths: Tensor, depths_gt: Tensor, rows: Tuple[int, int], height: int,
                        depth_lambda: float = 0.8, normal_lambda: float = 0.0, tile_size: int = 16,
                        K: Optional[Tensor] = None):
    """This rank's share of the tracker's loss (/root/reference/src/my_gsplat/gs_trainer_total.py:105-150):
    the L1 depth and L1 Sobel-edge terms of the pixel rows it owns, normalised by the FULL image size, so
    that the shares of all ranks add up to the single-GPU loss and their pose gradients to its gradient.
    ``depths`` [1,H,W,1] must be valid on the owned rows and one pixel row beyond (see halo_rows).
    Returns (total_share, depth_share, silhouette_share)."""
    from .my_gsplat.loss import sobel

    r0, r1 = rows[0] * tile_size, min(rows[1] * tile_size, height)
    if r1 <= r0:
        z = depths.sum() * 0.0
        return z, z, z
    h0, h1 = max(r0 - 1, 0), min(r1 + 1, height)  # one pixel row of halo
    d = depths[:, h0:h1]
    g = depths_gt[:, h0:h1]
    mask = (d != 0).float()
    dm, gm = d * mask, g * mask
    P = float(depths.shape[1] * depths.shape[2] * depths.shape[0] * depths.shape[3])
    own = slice(r0 - h0, r0 - h0 + (r1 - r0))
    depth_share = (dm[:, own] - gm[:, own]).abs().sum() / P
    ea = sobel(dm.permute(0, 3, 1, 2))
    eb = sobel(gm.permute(0, 3, 1, 2))
    sil_share = (ea[:, :, own] - eb[:, :, own]).abs().sum() / P
    total = depth_share * depth_lambda + sil_share * (1 - depth_lambda - normal_lambda)
    if normal_lambda != 0.0:
        # the (switched-off) normal term of the reference is a mean over image rows of row-wise cosines
        # (loss.py:62-101 with dim=1 on [H,W,3] maps): a rank owns the cosines of its rows
        import torch.nn.functional as F
        from .my_gsplat.geometry import depth_to_normal
        keep = torch.zeros_like(depths)
        keep[:, h0:h1] = 1.0
        m_all = (depths != 0).float() * keep
        na = depth_to_normal((depths * m_all)[0, :, :, 0], K)
        nb = depth_to_normal((depths_gt * m_all)[0, :, :, 0], K)
        cos = F.cosine_similarity(na[r0:r1], nb[r0:r1], dim=1)  # [rows, 3]
        total = total + normal_lambda * ((r1 - r0) / float(height) - cos.sum() / (3.0 * height))
    return total, depth_share, sil_share
