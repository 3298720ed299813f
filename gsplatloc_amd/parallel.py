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

from typing import List, Sequence, Tuple

import torch
from torch import Tensor


def strip_rows(tile_offsets: Tensor, tile_w: int, tile_h: int, world: int) -> List[Tuple[int, int]]:
    """Split tile rows into `world` contiguous strips with balanced intersection counts.
    tile_offsets: [tile_w*tile_h + 1] exclusive scan from a full-frame binning pass."""
    offs = tile_offsets.detach().to("cpu", torch.int64)
    row_start = offs[0:tile_w * tile_h + 1:tile_w]  # offset at the start of each tile row (+ total)
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
