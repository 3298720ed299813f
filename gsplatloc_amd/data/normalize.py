"""PCA normalisation of a frame pair (mirror of /root/reference/src/data/normalize.py:8-124)."""
import torch
from torch import Tensor


@torch.no_grad()
def align_principle_axes(point_cloud: Tensor) -> Tensor:
    """normalize.py:8-50: median-centred PCA frame; eigenvectors by descending eigenvalue, first axis
    flipped if the frame is left-handed.  Returns the 4x4 world->normalised transform."""
    centre = point_cloud.median(dim=0).values
    spread, axes = torch.linalg.eigh(torch.cov((point_cloud - centre).T))  # ascending eigenvalues
    axes = axes.flip(dims=(1,))  # largest spread first
    if torch.linalg.det(axes) < 0:  # keep the frame right-handed
        axes[:, 0].neg_()
    T = torch.eye(4, device=point_cloud.device, dtype=point_cloud.dtype)
    T[:3, :3] = axes.T
    T[:3, 3] = -(axes.T @ centre)
    return T


@torch.no_grad()
def transform_points(matrix: Tensor, points: Tensor) -> Tensor:
    """normalize.py:53-72."""
    return points @ matrix[:3, :3].T + matrix[:3, 3]


@torch.no_grad()
def transform_cameras(matrix: Tensor, c2w: Tensor):
    """normalize.py:75-104: T @ c2w, rotation re-normalised by the norm of its first row (returned as the
    scale factor; 1 for a rigid T)."""
    moved = matrix.unsqueeze(0) @ c2w
    scale = moved[:, 0, :3].norm(dim=1, keepdim=True)
    moved[:, :3, :3] = moved[:, :3, :3] / scale.unsqueeze(-1)
    return moved, scale


@torch.no_grad()
def normalize_pair(tar_points: Tensor, tar_pose: Tensor, src_points: Tensor, src_pose: Tensor):
    """normalize.py:107-124 (normalize_2C): PCA frame of the target cloud applied to both clouds and poses."""
    T = align_principle_axes(tar_points)
    tar_pose_n, scale = transform_cameras(T, tar_pose.unsqueeze(0))
    src_pose_n, _ = transform_cameras(T, src_pose.unsqueeze(0))
    return (transform_points(T, tar_points), tar_pose_n.squeeze(0), transform_points(T, src_points),
            src_pose_n.squeeze(0), scale)
