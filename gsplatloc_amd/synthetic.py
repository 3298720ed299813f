"""Seeded synthetic workloads: no dataset is needed or read.

* ``random_scene``   -- "Random-N": u,v uniform over the image, z ~ U(1,5) m, isotropic
  scale s = sigma_px * z / fx, identity quaternions, opacity 1 (as GSModel builds them,
  /root/reference/src/my_gsplat/model.py:156-165), SH degree 1 with a random DC term.
* ``replica_intrinsics`` -- /root/reference/datasets/Replica/cam_params.json:3-9.
* ``perturbed_pose`` -- GT = identity, initial pose = GT rotated by ``rot_deg`` about a seeded
  axis and shifted by ``trans`` metres.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

SH_C0 = 0.28209479177387814


def replica_intrinsics(W: int = 1200, H: int = 680, dtype=torch.float32) -> torch.Tensor:
    fx = 600.0 * W / 1200.0
    fy = 600.0 * H / 680.0
    return torch.tensor([[fx, 0.0, (W - 1) / 2.0], [0.0, fy, (H - 1) / 2.0], [0.0, 0.0, 1.0]], dtype=dtype)


def random_scene(N: int, W: int, H: int, seed: int = 42, sigma_px: float = 1.0, device="cpu",
                 order: str = "random") -> Dict:
    """order="raster" sorts the Gaussians by the pixel they project to at the identity pose (row-major),
    the order GsplatLoc's clouds have: one Gaussian per pixel of the previous depth frame
    (/root/reference/src/data/Image.py:29,35)."""
    g = torch.Generator().manual_seed(seed)
    K = replica_intrinsics(W, H)
    fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    u = torch.rand(N, generator=g) * W
    v = torch.rand(N, generator=g) * H
    z = 1.0 + 4.0 * torch.rand(N, generator=g)
    if order == "raster":
        perm = torch.argsort(torch.floor(v) * W + u)
        u, v, z = u[perm], v[perm], z[perm]
    means = torch.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], -1)
    quats = torch.tensor([1.0, 0.0, 0.0, 0.0]).repeat(N, 1)
    scales = (max(sigma_px, 1e-4) * z / fx)[:, None].repeat(1, 3)
    opacities = torch.ones(N)
    rgbs = torch.rand(N, 3, generator=g)
    sh = torch.zeros(N, 4, 3)
    sh[:, 0, :] = (rgbs - 0.5) / SH_C0
    out = dict(means=means, quats=quats, scales=scales, opacities=opacities, sh=sh, rgbs=rgbs, K=K, W=W, H=H, N=N)
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in out.items()}


def depth_upstream(H: int, W: int, seed: int = 1, channels: int = 4) -> torch.Tensor:
    """Upstream gradient of the parity statements and of the bench's timed step: white noise on the depth channel
    (float32 draw, seeded), zeros elsewhere.  ONE definition for bench.py and tests/test_gpu_configs.py: a pose gradient
    is a sum of ~1e6 terms of random sign, so another draw (or the same seed drawn in another dtype) moves its relative
    error by 2-3 x -- round 3's bench line and its test log disagreed for that reason only."""
    g = torch.Generator().manual_seed(seed)
    v = torch.zeros(H, W, channels)
    v[..., channels - 1] = torch.randn(H, W, generator=g)
    return v


def perturbed_pose(rot_deg: float = 0.5, trans: float = 0.01, seed: int = 7) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    ax = torch.randn(3, generator=g, dtype=torch.float64)
    ax = ax / ax.norm()
    th = math.radians(rot_deg)
    Kx = torch.tensor([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]], dtype=torch.float64)
    R = torch.eye(3, dtype=torch.float64) + math.sin(th) * Kx + (1 - math.cos(th)) * (Kx @ Kx)
    d = torch.randn(3, generator=g, dtype=torch.float64)
    d = d / d.norm() * trans
    c2w = torch.eye(4, dtype=torch.float64)
    c2w[:3, :3] = R
    c2w[:3, 3] = d
    return c2w.float()


def room_depth(W: int, H: int, K: torch.Tensor, c2w: torch.Tensor, half=(2.5, 1.5, 3.0)) -> torch.Tensor:
    """Depth image [H,W] (z along the optical axis) of an axis-aligned box room |x|<=hx, |y|<=hy, |z|<=hz seen
    from inside through camera ``c2w`` -- a stand-in for a Replica depth frame (no dataset in the container)."""
    dt = torch.float64
    K, c2w = K.to(dt), c2w.to(dt)
    v, u = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    d_cam = torch.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], torch.ones_like(u)], -1)  # z = 1
    d_w = d_cam @ c2w[:3, :3].T
    o = c2w[:3, 3]
    t_best = torch.full((H, W), float("inf"), dtype=dt)
    for ax in range(3):
        for sgn in (-1.0, 1.0):
            t = (sgn * half[ax] - o[ax]) / d_w[..., ax]
            t = torch.where(t > 1e-6, t, torch.full_like(t, float("inf")))
            t_best = torch.minimum(t_best, t)
    return t_best.float()  # ray parameter with d_cam.z = 1 == depth along the optical axis


def frame_pair(W: int = 160, H: int = 120, rot_deg: float = 0.5, trans: float = 0.01, seed: int = 7) -> Dict:
    """Synthetic GsplatLoc frame pair (/root/reference/src/data/dataset.py:345-383 in miniature):
    target cloud = back-projection of frame 0's depth (pose = identity), query depth = frame 1's depth,
    ground-truth pose of frame 1 = identity perturbed by (rot_deg, trans)."""
    K = replica_intrinsics(W, H)
    c2w0 = torch.eye(4)
    c2w1 = perturbed_pose(rot_deg, trans, seed)
    d0 = room_depth(W, H, K, c2w0)
    d1 = room_depth(W, H, K, c2w1)
    g = torch.Generator().manual_seed(seed)
    rgb = torch.rand(H * W, 3, generator=g)
    return dict(K=K, W=W, H=H, depth0=d0, depth1=d1, c2w0=c2w0, c2w1=c2w1, rgb=rgb)


def depth_frame_scene(W: int = 640, H: int = 480, stride: int = 1, holes: bool = False, device="cuda", seed: int = 3,
                      hole_frac: float = 0.08, pile: bool = False) -> Dict:
    """Workloads S / T of SURVEY.md 8(d): the Gaussians GsplatLoc builds from one depth frame
    (/root/reference/src/data/Image.py:29-35, my_gsplat/geometry.py:44-66,138-161): every ``stride``-th pixel of a
    synthetic room depth image back-projected in raster order, isotropic scales from the 4 nearest neighbours as the
    reference codes them (device k-NN), opacity 1, SH degree 1 with a random DC colour.  ``holes`` zeroes rectangular
    patches of the depth image as a TUM frame has them: those points sit at the camera origin and are culled by the
    near plane.  Rendered from the frame pair's second pose, chosen so that the camera has moved FORWARD: the invalid
    points then lie behind the near plane as SURVEY.md A.7 describes the usual case (with a sideways or backward
    step they are splatted as one enormous pile at the image centre -- handled, tests/test_gpu_configs.py has that
    frame, but it measures the pile, not the frame; ``pile=True`` selects exactly that step: the invalid points pass
    the near plane and land on one spot, ~23 k entries in one tile list at 640x480).  Needs the GPU (k-NN kernels)."""
    from .my_gsplat.geometry import depth_to_points, init_gs_scales

    fp = frame_pair(W, H, rot_deg=0.4, trans=0.015, seed=seed)
    c2w1 = fp["c2w1"].clone()
    in_front = float((-c2w1[:3, :3].T @ c2w1[:3, 3])[2]) > 0  # world origin in front of the second camera
    if in_front != pile:  # step the other way
        c2w1[:3, 3] = -c2w1[:3, 3]
    fp["c2w1"] = c2w1
    depth = fp["depth0"].clone()
    if holes:
        g = torch.Generator().manual_seed(seed)
        for _ in range(int(hole_frac * W * H / (24 * 18))):
            x0 = int(torch.randint(0, W - 24, (1,), generator=g))
            y0 = int(torch.randint(0, H - 18, (1,), generator=g))
            depth[y0:y0 + 18, x0:x0 + 24] = 0.0
    K = fp["K"].to(device)
    pts = depth_to_points(depth.to(device), K)[::stride].contiguous()
    rgb = fp["rgb"].to(device)[::stride].contiguous()
    valid = pts[:, 2] > 0
    scales = torch.full_like(pts, 1e-6)
    scales[valid] = init_gs_scales(pts[valid].contiguous())
    N = pts.shape[0]
    sh = torch.zeros(N, 4, 3, device=device)
    sh[:, 0, :] = (rgb - 0.5) / SH_C0
    quats = torch.tensor([1.0, 0.0, 0.0, 0.0], device=device).repeat(N, 1).contiguous()
    return dict(means=pts, quats=quats, scales=scales.contiguous(), opacities=torch.ones(N, device=device), sh=sh, K=K,
                W=W, H=H, N=N, viewmat=torch.linalg.inv(fp["c2w1"]).to(device).contiguous())


def write_replica_sequence(root, W, H, n):
    """Replica layout (results/depthNNNNNN.png at scale 6553.5, frameNNNNNN.jpg, traj.txt, cam_params.json) of a
    camera drifting through the synthetic room: ~1 cm and ~0.4 deg between frames."""
    import json

    import numpy as np
    from PIL import Image

    d = root / "room0" / "results"
    d.mkdir(parents=True)
    K = replica_intrinsics(W, H)
    cam = {"camera": {"w": W, "h": H, "fx": float(K[0, 0]), "fy": float(K[1, 1]), "cx": float(K[0, 2]),
                      "cy": float(K[1, 2]), "scale": 6553.5}}
    (root / "cam_params.json").write_text(json.dumps(cam))
    rng = np.random.default_rng(5)
    poses = []
    c2w = np.eye(4)
    for i in range(n):
        if i:
            ax = rng.normal(size=3)
            ax /= np.linalg.norm(ax)
            th = np.radians(0.4)
            Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)
            step = np.eye(4)
            step[:3, :3] = R
            t = rng.normal(size=3)
            step[:3, 3] = 0.01 * t / np.linalg.norm(t)
            c2w = c2w @ step
        depth = room_depth(W, H, K, torch.from_numpy(c2w).float()).numpy()
        Image.fromarray(np.round(depth * 6553.5).astype(np.uint16)).save(d / f"depth{i:06d}.png")
        Image.fromarray(rng.integers(0, 255, (H, W, 3), dtype=np.uint8)).save(d / f"frame{i:06d}.jpg")
        poses.append(c2w.copy())
    with open(root / "room0" / "traj.txt", "w") as f:
        for p in poses:
            f.write(" ".join(f"{v:.12f}" for v in p.reshape(-1)) + "\n")


def write_tum_sequence(root, W, H, n, hole_frac: float = 0.06, crop_edge: int = 8, name: str = "freiburg1_desk"):
    """TUM RGB-D layout (rgbd_dataset_<name>/{rgb/*.png, depth/*.png at scale 5000, rgb.txt, depth.txt, groundtruth.txt
    with `timestamp tx ty tz qx qy qz qw`, cam_params.json with crop_edge}) of the same drifting camera, 15 Hz, the
    depth and pose timestamps a few milliseconds off the colour ones, and rectangular patches of INVALID (zero) depth
    in every frame as a Kinect frame has them."""
    import json

    import numpy as np
    from PIL import Image
    from scipy.spatial.transform import Rotation

    d = root / f"rgbd_dataset_{name}"
    (d / "rgb").mkdir(parents=True)
    (d / "depth").mkdir()
    K = replica_intrinsics(W, H)
    cam = {"camera": {"w": W, "h": H, "fx": float(K[0, 0]), "fy": float(K[1, 1]), "cx": float(K[0, 2]),
                      "cy": float(K[1, 2]), "scale": 5000.0, "crop_edge": crop_edge}}
    (d / "cam_params.json").write_text(json.dumps(cam))
    rng = np.random.default_rng(9)
    c2w = np.eye(4)
    t0 = 1305031450.0
    with open(d / "rgb.txt", "w") as fr, open(d / "depth.txt", "w") as fd, open(d / "groundtruth.txt", "w") as fg:
        fr.write("# color images\n")
        fd.write("# depth maps\n")
        fg.write("# timestamp tx ty tz qx qy qz qw\n")
        for i in range(n):
            if i:
                ax = rng.normal(size=3)
                ax /= np.linalg.norm(ax)
                step = np.eye(4)
                step[:3, :3] = Rotation.from_rotvec(np.radians(0.4) * ax).as_matrix()
                t = rng.normal(size=3)
                step[:3, 3] = 0.01 * t / np.linalg.norm(t)
                c2w = c2w @ step
            depth = room_depth(W, H, K, torch.from_numpy(c2w).float()).numpy()
            for _ in range(int(hole_frac * W * H / (24 * 18))):
                x0, y0 = int(rng.integers(0, W - 24)), int(rng.integers(0, H - 18))
                depth[y0:y0 + 18, x0:x0 + 24] = 0.0
            ts = t0 + i / 15.0
            Image.fromarray(np.round(depth * 5000.0).astype(np.uint16)).save(d / "depth" / f"{ts:.6f}.png")
            Image.fromarray(rng.integers(0, 255, (H, W, 3), dtype=np.uint8)).save(d / "rgb" / f"{ts:.6f}.png")
            q = Rotation.from_matrix(c2w[:3, :3]).as_quat()  # x y z w
            fr.write(f"{ts:.6f} rgb/{ts:.6f}.png\n")
            fd.write(f"{ts + 0.004:.6f} depth/{ts:.6f}.png\n")
            fg.write(f"{ts + 0.002:.6f} {c2w[0, 3]:.9f} {c2w[1, 3]:.9f} {c2w[2, 3]:.9f} {q[0]:.9f} {q[1]:.9f} {q[2]:.9f} {q[3]:.9f}\n")
