"""Mirror of /root/reference/src/my_gsplat/geometry.py on the HIP rasterizer."""
import torch
import torch.nn.functional as F
from torch import Tensor

from ..rendering import rasterization
from .utils import knn, rgb_to_sh


def construct_full_pose(rotation: Tensor, translation: Tensor) -> Tensor:
    """geometry.py:12-20: [R|t] into a 4x4 with gradients to both."""
    pose = torch.eye(4, dtype=rotation.dtype, device=rotation.device)
    pose[:3, :3] = rotation
    pose[:3, 3] = translation
    return pose


def transform_points(matrix: Tensor, points: Tensor) -> Tensor:
    """geometry.py:23-41."""
    assert matrix.shape == (4, 4)
    assert len(points.shape) == 2 and points.shape[1] == 3
    return torch.addmm(matrix[:3, 3], points, matrix[:3, :3].t())


def init_gs_scales(points: Tensor, k: int = 5, eps: float = 1e-24) -> Tensor:
    """geometry.py:44-66: isotropic scale from the k-1 nearest neighbours."""
    dist2_avg = (knn(points, k)[:, 1:] ** 2).mean(dim=-1)
    dist_avg = torch.sqrt(dist2_avg + eps)
    return dist_avg.unsqueeze(-1).repeat(1, 3)


@torch.no_grad()
def compute_depth_gt(points: Tensor, rgbs: Tensor, Ks: Tensor, c2w: Tensor, height: int, width: int) -> Tensor:
    """geometry.py:69-135: expected-depth ("ED") render of a point cloud -> [H, W]."""
    N = points.shape[0]
    dev = points.device
    opacities = torch.sigmoid(torch.logit(torch.full((N,), 1.0, device=dev)))
    scales = init_gs_scales(points)
    quats = torch.tensor([1.0, 0.0, 0.0, 0.0], device=dev).repeat(N, 1)
    sh_degree = 1
    colors = torch.zeros((N, (sh_degree + 1) ** 2, 3), device=dev)
    colors[:, 0, :] = rgb_to_sh(rgbs)
    render_colors, _, _ = rasterization(
        means=points, quats=quats, scales=scales, opacities=opacities, colors=colors, sh_degree=sh_degree,
        viewmats=torch.linalg.inv(c2w), Ks=Ks, width=width, height=height, far_plane=1e10, near_plane=1e-2,
        render_mode="ED", rasterize_mode="classic", packed=False)
    return render_colors.squeeze(0).squeeze(-1)


def depth_to_points(depth: Tensor, K: Tensor, include_homogeneous: bool = False) -> Tensor:
    """geometry.py:138-161 (kornia depth_to_3d_v2): back-project on the integer pixel grid, row-major."""
    H, W = depth.shape
    v, u = torch.meshgrid(torch.arange(H, device=depth.device, dtype=depth.dtype),
                          torch.arange(W, device=depth.device, dtype=depth.dtype), indexing="ij")
    x = (u - K[0, 2]) / K[0, 0] * depth
    y = (v - K[1, 2]) / K[1, 1] * depth
    pts = torch.stack([x, y, depth], dim=-1).view(-1, 3)
    if include_homogeneous:
        pts = F.pad(pts, (0, 1), value=1)
    return pts


def depth_to_normal(depth: Tensor, K: Tensor) -> Tensor:
    """geometry.py:164-197: normals from central differences of the back-projected points."""
    H, W = depth.shape
    points = depth_to_points(depth, K).view(H, W, 3).unsqueeze(0)
    pp = F.pad(points, (0, 0, 1, 1, 1, 1), mode="replicate")
    dx = pp[:, 1:-1, 2:, :] - pp[:, 1:-1, :-2, :]
    dy = pp[:, 2:, 1:-1, :] - pp[:, :-2, 1:-1, :]
    normal = F.normalize(torch.cross(dx, dy, dim=-1), p=2, dim=-1)
    return normal.squeeze(0)
