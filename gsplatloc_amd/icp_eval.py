"""ICP baseline over a sequence -- the role of /root/reference/src/icps_eval.py:26-85 +
ICPExperiment.run (/root/reference/src/eval/experiment.py:76-149) without W&B: register every scan
against its predecessor starting from the predecessor's ground-truth pose, and report the per-frame
translation / rotation error and their RMSE (ATE / AAE as in eval/utils.py:113-119).

    python -m gsplatloc_amd.icp_eval --dataset Replica --rooms office0 --root datasets/Replica --method GICP

CPU only (BASELINE.json configs[0]: "plumbing, no GPU").

As coded, the reference sets ``pre_pose`` and ``pose_gt`` from the *same* frame (experiment.py:86-87), so its
initial guess ``pose_gt @ inv(pre_pose)`` is always the identity and the accumulated estimate is reset to the
current frame's own ground truth before the relative motion is applied.  The evident intent -- reset to the
*previous* frame's ground truth, identity initial guess -- is what ``run_icp_sequence`` does.
"""
from __future__ import annotations

import argparse
import json
import math
import time
from typing import Dict, Iterable, List, Optional

import numpy as np

from .component import Scan2ScanICP


def translation_error(est: np.ndarray, gt: np.ndarray) -> float:
    """calculate_translation_error_np (eval/utils.py:14-31)."""
    return float(np.linalg.norm(est[:3, 3] - gt[:3, 3]))


def rotation_error(est: np.ndarray, gt: np.ndarray) -> float:
    """calculate_rotation_error_np (eval/utils.py:34-57): angle of R_est^T R_gt in degrees."""
    c = (np.trace(est[:3, :3].T @ gt[:3, :3]) - 1.0) / 2.0
    return float(np.degrees(np.arccos(np.clip(c, -1.0, 1.0))))


def rmse(values: Iterable[float]) -> float:
    v = list(values)
    return math.sqrt(sum(x * x for x in v) / max(len(v), 1))


def run_icp_sequence(frames, registration_type: str = "GICP", max_images: int = 2000, knn: int = 20,
                     max_corresponding_distance: float = 0.1, voxel_downsampling_resolutions: float = 0.0,
                     num_threads: int = 8, stride: int = 1) -> Dict:
    """``frames``: iterable of objects with ``.points`` ([n,3] camera-frame points, tensor or array) and
    ``.pose`` (4x4 camera-to-world), e.g. gsplatloc_amd.data.Replica / TUM items.  ``stride`` subsamples the
    points of every scan (1 = all, as the reference)."""
    icp = Scan2ScanICP(max_corresponding_distance=max_corresponding_distance,
                       voxel_downsampling_resolutions=voxel_downsampling_resolutions, knn=knn,
                       num_threads=num_threads, registration_type=registration_type)
    eTs: List[float] = []
    eRs: List[float] = []
    iters: List[int] = []
    prev_pose: Optional[np.ndarray] = None
    t0 = time.perf_counter()
    # the readers signal "out of range" with ValueError (as the reference's do), which the sequence iteration
    # protocol does not treat as the end: walk sized collections by index
    walk = (frames[i] for i in range(min(len(frames), max_images))) if hasattr(frames, "__len__") else frames
    for i, frame in enumerate(walk):
        pts = np.asarray(frame.points.cpu() if hasattr(frame.points, "cpu") else frame.points, dtype=np.float64)
        pose_gt = np.asarray(frame.pose.cpu() if hasattr(frame.pose, "cpu") else frame.pose, dtype=np.float64)
        pts = pts.reshape(-1, pts.shape[-1])[::stride, :3]
        pts = pts[np.isfinite(pts).all(axis=1) & (pts[:, 2] > 0)]
        if i == 0:
            icp.align(pts, pose_gt)
        else:
            icp.T_world_camera = prev_pose
            icp.align(pts, np.identity(4))
            est = icp.T_world_camera
            eTs.append(translation_error(est, pose_gt))
            eRs.append(rotation_error(est, pose_gt))
            iters.append(icp.last_result.iterations)
        prev_pose = pose_gt
        if i >= max_images - 1:
            break
    dt = time.perf_counter() - t0
    return {"method": registration_type, "frames": len(eTs) + 1, "ATE": rmse(eTs), "AAE": rmse(eRs), "eT": eTs,
            "eR": eRs, "iterations": iters, "seconds": dt, "frames_per_s": (len(eTs) + 1) / max(dt, 1e-9)}


def main(argv=None) -> None:
    from .data import get_data_set

    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--dataset", choices=["Replica", "TUM"], default="Replica")
    ap.add_argument("--rooms", nargs="+", default=["office0"])
    ap.add_argument("--root", default=None, help="dataset folder (default datasets/<dataset>)")
    ap.add_argument("--method", choices=["ICP", "PLANE_ICP", "GICP"], default="GICP")
    ap.add_argument("--max-images", type=int, default=2000)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--out", default=None)
    args = ap.parse_args(argv)
    report = {}
    for room in args.rooms:
        kw = {"input_folder": args.root} if args.root else {}
        data = get_data_set(args.dataset, room, **kw)
        res = run_icp_sequence(data, args.method, args.max_images, num_threads=args.threads, stride=args.stride)
        report[room] = {k: res[k] for k in ("method", "frames", "ATE", "AAE", "seconds", "frames_per_s")}
        print(room, json.dumps(report[room]))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
