"""Mirror of /root/reference/src/my_gsplat/loss.py; kornia.filters.sobel restated (loss.py:51-52)."""
from typing import Literal

import torch
from torch import Tensor
from torch.nn import functional as F

from .geometry import depth_to_normal


def sobel(x: Tensor, normalized: bool = True, eps: float = 1e-6) -> Tensor:
    """kornia.filters.sobel on [B,C,H,W]: replicate-pad 1, Sobel pair (/8 when normalised),
    sqrt(gx^2 + gy^2 + eps)."""
    kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]], dtype=x.dtype, device=x.device)
    ky = kx.t().contiguous()
    if normalized:
        kx, ky = kx / kx.abs().sum(), ky / ky.abs().sum()
    b, c, h, w = x.shape
    xp = F.pad(x.reshape(b * c, 1, h, w), (1, 1, 1, 1), mode="replicate")
    g = F.conv2d(xp, torch.stack([kx, ky])[:, None])
    return torch.sqrt(g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1] + eps).reshape(b, c, h, w)


def compute_depth_loss(depth_A: Tensor, depth_B: Tensor, *, loss_type: Literal["l1", "mse"] = "l1") -> Tensor:
    """loss.py:10-30."""
    if loss_type == "l1":
        return F.l1_loss(depth_A, depth_B)
    elif loss_type == "mse":
        return F.mse_loss(depth_A, depth_B)
    raise ValueError("Invalid loss type. Use 'mse' or 'l1'.")


def compute_silhouette_loss(depth_A: Tensor, depth_B: Tensor, *, loss_type: Literal["l1", "mse"] = "l1") -> Tensor:
    """loss.py:33-59: distance between Sobel edge maps; inputs [B,H,W,1]."""
    assert depth_A.dim() == 4 and depth_B.dim() == 4
    edge_A = sobel(depth_A.permute(0, 3, 1, 2))
    edge_B = sobel(depth_B.permute(0, 3, 1, 2))
    if loss_type == "l1":
        return F.l1_loss(edge_A, edge_B)
    elif loss_type == "mse":
        return F.mse_loss(edge_A, edge_B)
    raise ValueError("Invalid loss type. Use 'mse', 'l1', or 'huber'.")


def compute_normal_consistency_loss(depth_real: Tensor, depth_rendered: Tensor, *, K: Tensor,
                                    loss_type: Literal["cosine", "l1", "mse"] = "cosine") -> Tensor:
    """loss.py:62-101 (unused by the tracker: normal_lambda = 0)."""
    if depth_real.dim() == 3:
        depth_real = depth_real.squeeze(0)
    if depth_rendered.dim() == 3:
        depth_rendered = depth_rendered.squeeze(0)
    n_real = depth_to_normal(depth_real, K=K)
    n_rend = depth_to_normal(depth_rendered, K=K)
    if loss_type == "cosine":
        return 1 - F.cosine_similarity(n_real, n_rend, dim=1).mean()
    elif loss_type == "l1":
        return F.l1_loss(n_real, n_rend)
    elif loss_type == "mse":
        return F.mse_loss(n_real, n_rend)
    raise ValueError("Invalid loss type. Use 'cosine', 'l1', or 'mse'.")
