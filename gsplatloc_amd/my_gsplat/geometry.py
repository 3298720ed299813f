"""Geometry helpers of the tracker with the interface of /root/reference/src/my_gsplat/geometry.py
(construct_full_pose :12-20, transform_points :23-41, init_gs_scales :44-66, compute_depth_gt :69-135,
depth_to_points :138-161, depth_to_normal :164-197).  kornia.geometry.depth_to_3d_v2 (geometry.py:158) is
restated; the rasterizer is this package's HIP implementation; the k-NN behind the scales runs on the GPU
(csrc/knn.hip) when the points live there.
"""
from typing import Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from ..rendering import rasterization
from .utils import knn, rgb_to_sh


def construct_full_pose(rotation: Tensor, translation: Tensor) -> Tensor:
    """4x4 rigid transform [R t; 0 0 0 1] on the device of ``rotation``, differentiable in both blocks."""
    pose = torch.eye(4, dtype=rotation.dtype, device=rotation.device)
    pose[:3, :3] = rotation
    pose[:3, 3] = translation
    return pose


def transform_points(matrix: Tensor, points: Tensor) -> Tensor:
    """Apply a 4x4 rigid transform to points [N,3]."""
    assert matrix.shape == (4, 4)
    assert points.dim() == 2 and points.shape[1] == 3
    return torch.addmm(matrix[:3, 3], points, matrix[:3, :3].t())


def init_gs_scales(points: Tensor, k: int = 5, eps: float = 1e-24) -> Tensor:
    """Isotropic initial scale per point [N,3]: root of the mean of the squared values ``knn`` returns for the
    k-1 nearest other points.  (``knn`` already returns squared distances, as small_gicp does; the reference
    squares them again -- SURVEY.md A.7 -- and that as-coded behaviour is kept.)"""
    neighbour_terms = knn(points, k)[:, 1:] ** 2
    radius = torch.sqrt(neighbour_terms.mean(dim=-1) + eps)
    return radius.unsqueeze(-1).repeat(1, 3)


def _point_gaussians(points: Tensor, rgbs: Tensor, sh_degree: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """The Gaussians the reference puts on a point cloud: kNN scales, identity rotation, opacity
    sigmoid(logit(1)) = 1, SH coefficients with the DC band from the colours."""
    n, dev = points.shape[0], points.device
    quats = torch.zeros(n, 4, device=dev)
    quats[:, 0] = 1.0
    coefficients = torch.zeros(n, (sh_degree + 1) ** 2, 3, device=dev)
    coefficients[:, 0, :] = rgb_to_sh(rgbs)
    return init_gs_scales(points), quats, torch.ones(n, device=dev), coefficients


@torch.no_grad()
def compute_depth_gt(points: Tensor, rgbs: Tensor, Ks: Tensor, c2w: Tensor, height: int, width: int) -> Tensor:
    """Expected-depth ("ED") render [H,W] of a point cloud [N,3] seen from ``c2w`` [1,4,4] with ``Ks`` [1,3,3]:
    the target image of the tracker's loss."""
    sh_degree = 1
    scales, quats, opacities, coefficients = _point_gaussians(points, rgbs, sh_degree)
    render, _, _ = rasterization(means=points, quats=quats, scales=scales, opacities=opacities, colors=coefficients,
                                 sh_degree=sh_degree, viewmats=torch.linalg.inv(c2w), Ks=Ks, width=width,
                                 height=height, far_plane=1e10, near_plane=1e-2, render_mode="ED",
                                 rasterize_mode="classic", packed=False)
    return render[0, :, :, 0]


def depth_to_points(depth: Tensor, K: Tensor, include_homogeneous: bool = False) -> Tensor:
    """Back-project a depth image [H,W] through K: pixel (u, v) on the integer grid -> z * K^-1 (u, v, 1),
    row-major [H*W, 3] (or [H*W, 4] with a trailing 1)."""
    H, W = depth.shape
    v, u = torch.meshgrid(torch.arange(H, device=depth.device, dtype=depth.dtype),
                          torch.arange(W, device=depth.device, dtype=depth.dtype), indexing="ij")
    x = (u - K[0, 2]) / K[0, 0] * depth
    y = (v - K[1, 2]) / K[1, 1] * depth
    pts = torch.stack([x, y, depth], dim=-1).view(-1, 3)
    return F.pad(pts, (0, 1), value=1) if include_homogeneous else pts


def depth_to_normal(depth: Tensor, K: Tensor) -> Tensor:
    """Unit normals [H,W,3] of the back-projected surface: cross product of the central differences along
    the image columns and rows, borders replicated."""
    H, W = depth.shape
    surface = depth_to_points(depth, K).view(1, H, W, 3)
    padded = F.pad(surface, (0, 0, 1, 1, 1, 1), mode="replicate")
    along_x = padded[:, 1:-1, 2:] - padded[:, 1:-1, :-2]
    along_y = padded[:, 2:, 1:-1] - padded[:, :-2, 1:-1]
    return F.normalize(torch.cross(along_x, along_y, dim=-1), p=2, dim=-1)[0]
