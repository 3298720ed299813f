"""Tracking losses with the interface of /root/reference/src/my_gsplat/loss.py (compute_depth_loss :10-30,
compute_silhouette_loss :33-59, compute_normal_consistency_loss :62-101).  kornia.filters.sobel, which the
reference calls at loss.py:51-52, is restated here as one 2-channel convolution.  The tracker's per-iteration
loss does not run through these functions in GraphTracker (csrc/tracker.hip fuses depth L1, Sobel L1 and their
adjoint); they are the autograd form used by PoseTracker and by the tests that check the fused kernel.
"""
from typing import Callable, Dict, Literal

import torch
from torch import Tensor
from torch.nn import functional as F

from .geometry import depth_to_normal

_PIXEL_DISTANCES: Dict[str, Callable[[Tensor, Tensor], Tensor]] = {"l1": F.l1_loss, "mse": F.mse_loss}


def _distance(a: Tensor, b: Tensor, loss_type: str, complaint: str) -> Tensor:
    try:
        return _PIXEL_DISTANCES[loss_type](a, b)
    except KeyError:
        raise ValueError(complaint) from None


def sobel(x: Tensor, normalized: bool = True, eps: float = 1e-6) -> Tensor:
    """Gradient magnitude of every channel of x [B,C,H,W] as kornia.filters.sobel computes it: replicate
    padding by one pixel, the 3x3 Sobel pair (divided by 8 when ``normalized``), sqrt(gx^2 + gy^2 + eps)."""
    b, c, h, w = x.shape
    col = torch.tensor([1.0, 2.0, 1.0], dtype=x.dtype, device=x.device)
    dif = torch.tensor([-1.0, 0.0, 1.0], dtype=x.dtype, device=x.device)
    pair = torch.stack([torch.outer(col, dif), torch.outer(dif, col)])  # d/dx, d/dy
    if normalized:
        pair = pair / 8.0
    planes = F.pad(x.reshape(b * c, 1, h, w), (1, 1, 1, 1), mode="replicate")
    gx, gy = F.conv2d(planes, pair[:, None]).unbind(dim=1)
    return torch.sqrt(gx * gx + gy * gy + eps).reshape(b, c, h, w)


def compute_depth_loss(depth_A: Tensor, depth_B: Tensor, *, loss_type: Literal["l1", "mse"] = "l1") -> Tensor:
    """Mean absolute (or squared) difference of two depth images of any common shape."""
    return _distance(depth_A, depth_B, loss_type, "Invalid loss type. Use 'mse' or 'l1'.")


def compute_silhouette_loss(depth_A: Tensor, depth_B: Tensor, *, loss_type: Literal["l1", "mse"] = "l1") -> Tensor:
    """Distance between the Sobel edge maps of two depth images [B,H,W,1]."""
    assert depth_A.dim() == 4 and depth_B.dim() == 4
    edges = [sobel(d.permute(0, 3, 1, 2)) for d in (depth_A, depth_B)]
    return _distance(edges[0], edges[1], loss_type, "Invalid loss type. Use 'mse', 'l1', or 'huber'.")


def compute_normal_consistency_loss(depth_real: Tensor, depth_rendered: Tensor, *, K: Tensor,
                                    loss_type: Literal["cosine", "l1", "mse"] = "cosine") -> Tensor:
    """Disagreement of the normal maps derived from two depth images [H,W] or [1,H,W] (not used by the
    tracker: its normal_lambda is 0)."""
    normals = [depth_to_normal(d.squeeze(0) if d.dim() == 3 else d, K=K) for d in (depth_real, depth_rendered)]
    if loss_type == "cosine":
        return 1 - F.cosine_similarity(normals[0], normals[1], dim=1).mean()
    return _distance(normals[0], normals[1], loss_type, "Invalid loss type. Use 'cosine', 'l1', or 'mse'.")
