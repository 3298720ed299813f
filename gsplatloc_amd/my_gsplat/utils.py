"""Mirror of /root/reference/src/my_gsplat/utils.py.  ``knn`` used small_gicp's KdTree
(utils.py:16-22); here a scipy cKDTree on the host gives the same k nearest neighbours
(per-frame setup, not on the per-iteration path).  small_gicp's batch_knn_search returns SQUARED
distances and the reference uses them as returned: ``squared=True`` keeps that."""
import numpy as np
import torch
from torch import Tensor


def knn_device(x: Tensor, K: int) -> Tensor:
    """Exact K nearest neighbours (self included) of a device point cloud: SQUARED distances [N,K],
    ascending -- csrc/knn.hip (uniform grid, shell search), no host round trip."""
    from .._lib import check, current_stream, load_library, ptr

    lib = load_library()
    assert x.is_cuda and x.dim() == 2 and x.shape[1] == 3 and 1 <= K <= 8
    x = x.detach().to(torch.float32).contiguous()
    N = x.shape[0]
    bbox = torch.cat([x.amin(0), x.amax(0)]).contiguous()
    ws_bytes = lib.gsl_knn_ws_bytes(N)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    st = current_stream()
    check(lib.gsl_knn_count(ptr(x), N, ptr(bbox), ptr(ws), ws_bytes, st), "gsl_knn_count")
    cells = lib.gsl_knn_cells()
    counts = ws[:cells * 4].view(torch.int32)
    incl = torch.cumsum(counts, 0, dtype=torch.int32)
    dists = torch.empty(N, K, dtype=torch.float32, device=x.device)
    check(lib.gsl_knn_query(ptr(x), N, ptr(bbox), ptr(incl), K, ptr(dists), ptr(ws), ws_bytes, st), "gsl_knn_query")
    return dists


def knn(x: Tensor, K: int = 4, squared: bool = True) -> Tensor:
    if x.is_cuda and K <= 8:
        d2 = knn_device(x, K)
        return d2 if squared else torch.sqrt(d2)
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
