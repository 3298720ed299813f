"""Mirror of /root/reference/src/my_gsplat/utils.py.  ``knn`` used small_gicp's KdTree
(utils.py:16-22); here a scipy cKDTree on the host gives the same k nearest neighbours
(per-frame setup, not on the per-iteration path).  small_gicp's batch_knn_search returns SQUARED
distances and the reference uses them as returned: ``squared=True`` keeps that."""
import numpy as np
import torch
from torch import Tensor


def knn(x: Tensor, K: int = 4, squared: bool = True) -> Tensor:
    from scipy.spatial import cKDTree

    x_np = x.detach().cpu().numpy().astype(np.float64)
    d, _ = cKDTree(x_np).query(x_np, k=K, workers=-1)
    if squared:
        d = d * d
    return torch.from_numpy(d).to(dtype=torch.float32, device=x.device)


def remove_outliers(points: Tensor, k: int = 10, std_ratio: float = 10.0, verbose: bool = False):
    """utils.py:25-50."""
    distances = knn(points, k)
    dist_avg = torch.sqrt((distances[:, 1:] ** 2).mean(dim=-1))
    threshold = dist_avg.mean() + std_ratio * dist_avg.std()
    inlier_mask = dist_avg < threshold
    cleaned = points[inlier_mask]
    if verbose:
        print(f"Original points: {len(points)}\nPoints after cleaning: {len(cleaned)}")
    return cleaned, inlier_mask


def rgb_to_sh(rgb: Tensor) -> Tensor:
    """utils.py:53-55."""
    C0 = 0.28209479177387814
    return (rgb - 0.5) / C0
