"""Dataset readers on synthetic files written in the Replica / TUM on-disk formats (no dataset ships in the
container; formats: /root/reference/src/data/dataset.py:78-321, /root/reference/datasets/Replica/cam_params.json)."""
import json
import os

import numpy as np
import pytest
import torch

from gsplatloc_amd.synthetic import frame_pair, perturbed_pose, replica_intrinsics, room_depth


def write_replica(root, name="room0", W=160, H=120, n=3):
    from PIL import Image
    d = root / name / "results"
    d.mkdir(parents=True)
    K = replica_intrinsics(W, H)
    cam = {"camera": {"w": W, "h": H, "fx": float(K[0, 0]), "fy": float(K[1, 1]), "cx": float(K[0, 2]),
                      "cy": float(K[1, 2]), "scale": 6553.5}}
    (root / "cam_params.json").write_text(json.dumps(cam))
    poses = []
    rng = np.random.default_rng(0)
    for i in range(n):
        c2w = perturbed_pose(0.3 * i, 0.01 * i, seed=3)
        depth = room_depth(W, H, K, c2w).numpy()
        Image.fromarray(np.round(depth * 6553.5).astype(np.uint16)).save(d / f"depth{i:06d}.png")
        Image.fromarray(rng.integers(0, 255, (H, W, 3), dtype=np.uint8)).save(d / f"frame{i:06d}.jpg")
        poses.append(c2w.double().numpy())
    with open(root / name / "traj.txt", "w") as f:
        for p in poses:
            f.write(" ".join(f"{v:.10f}" for v in p.reshape(-1)) + "\n")
    return K, poses


def test_replica_reader_roundtrip(tmp_path):
    from gsplatloc_amd.data import Replica
    K, poses = write_replica(tmp_path)
    ds = Replica("room0", input_folder=tmp_path)
    assert len(ds) == 3 and "Replica dataset" in str(ds)
    f = ds[1]
    assert f.depth.shape == (120, 160) and f.rgb.shape == (120, 160, 3)
    np.testing.assert_allclose(f.pose, poses[1], atol=1e-9)
    ref = room_depth(160, 120, K, torch.from_numpy(poses[1]).float()).numpy()
    np.testing.assert_allclose(f.depth, ref, atol=1.0 / 6553.5)
    assert len(ds[0:2]) == 2
    with pytest.raises(ValueError):
        ds[3]
    with pytest.raises(TypeError):
        ds["a"]


def test_tum_reader_association(tmp_path):
    from PIL import Image
    from gsplatloc_amd.data import TUM
    d = tmp_path / "rgbd_dataset_freiburg1_desk"
    (d / "rgb").mkdir(parents=True)
    (d / "depth").mkdir()
    W, H, crop = 64, 48, 4
    (d / "cam_params.json").write_text(json.dumps({"camera": {"w": W, "h": H, "fx": 50.0, "fy": 50.0, "cx": 31.5,
                                                               "cy": 23.5, "scale": 5000.0, "crop_edge": crop}}))
    ts = [1.00, 1.02, 1.10, 1.30]
    with open(d / "rgb.txt", "w") as fr, open(d / "depth.txt", "w") as fd, open(d / "groundtruth.txt", "w") as fg:
        fr.write("# color images\n"); fd.write("# depth maps\n"); fg.write("# timestamp tx ty tz qx qy qz qw\n")
        for i, t in enumerate(ts):
            Image.fromarray(np.full((H, W, 3), 10 * i, np.uint8)).save(d / "rgb" / f"{t:.2f}.png")
            Image.fromarray(np.full((H, W), 5000 * (i + 1), np.uint16)).save(d / "depth" / f"{t:.2f}.png")
            fr.write(f"{t:.4f} rgb/{t:.2f}.png\n")
            fd.write(f"{t + 0.01:.4f} depth/{t:.2f}.png\n")
            fg.write(f"{t + 0.005:.4f} {0.1 * i} 0 0 0 0 0 1\n")
    ds = TUM("freiburg1_desk", input_folder=tmp_path, frame_rate=32)
    # frames closer than 1/32 s to the last kept one are dropped: 1.00, (1.02 dropped), 1.10, 1.30
    assert len(ds) == 3
    f0, f1 = ds[0], ds[1]
    assert f0.depth.shape == (H - 2 * crop, W - 2 * crop)
    assert np.allclose(f0.depth, 1.0) and np.allclose(f1.depth, 3.0)
    assert np.allclose(f0.pose, np.eye(4)) and np.allclose(f1.pose[:3, 3], [0.2, 0, 0])  # relative to the first frame
    assert ds.K[0, 2] == 31.5 - crop


def test_pca_normalisation_is_rigid():
    from gsplatloc_amd.data import align_principle_axes, transform_cameras
    g = torch.Generator().manual_seed(0)
    pts = torch.randn(500, 3, generator=g) * torch.tensor([3.0, 1.0, 0.2]) + torch.tensor([1.0, -2.0, 0.5])
    T = align_principle_axes(pts)
    R = T[:3, :3]
    assert torch.allclose(R @ R.T, torch.eye(3), atol=1e-5) and torch.det(R) > 0
    out = (pts @ R.T + T[:3, 3])
    c = torch.median(pts, dim=0).values
    assert (R @ c + T[:3, 3]).abs().max() < 1e-5             # the per-axis median maps to the origin
    v = out.var(0)
    assert v[0] > v[1] > v[2]                                # axes by descending variance
    c2w, s = transform_cameras(T, torch.eye(4)[None])
    assert torch.allclose(s, torch.ones_like(s), atol=1e-5)


@pytest.mark.gpu
def test_sequence_evaluation_on_synthetic_replica(tmp_path):
    """End to end: Replica files -> Parser (PCA, query-depth render) -> GraphTracker -> ATE/AAE report."""
    from gsplatloc_amd.data import Parser
    from gsplatloc_amd.eval import evaluate_room
    write_replica(tmp_path, n=3)
    parser = Parser("Replica", "room0", normalize=True, input_folder=tmp_path)
    d = parser[0]
    assert d.tar_points.shape == (160 * 120, 3) and d.src_depth.shape == (1, 120, 160, 1)
    r = evaluate_room(parser, num_iters=150, max_frames=2)
    assert r["frames"] == 2 and r["frames_with_result"] == 2
    assert r["ATE"] < 0.006 and np.isfinite(r["AAE"])       # start error is 1 cm per frame pair
