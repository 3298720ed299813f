"""Seeded synthetic inputs shared by the tests (inputs only; no oracle code)."""
import math

import numpy as np
import torch


def random_scene(N, W, H, seed=42, sigma_px=1.0, dtype=torch.float64, fx=None, aniso=False, opacity=None):
    """Random-N scene of SURVEY.md 8(d): u,v uniform on the image, z~U(1,5)."""
    g = torch.Generator().manual_seed(seed)
    fx = fx if fx is not None else 0.5 * W
    fy = fx
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    u = torch.rand(N, generator=g, dtype=torch.float64) * W
    v = torch.rand(N, generator=g, dtype=torch.float64) * H
    z = 1.0 + 4.0 * torch.rand(N, generator=g, dtype=torch.float64)
    means = torch.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], -1)
    if aniso:
        quats = torch.randn(N, 4, generator=g, dtype=torch.float64)
        scales = (sigma_px * z / fx)[:, None] * (0.5 + torch.rand(N, 3, generator=g, dtype=torch.float64))
    else:
        quats = torch.tensor([1.0, 0, 0, 0], dtype=torch.float64).repeat(N, 1)
        scales = (sigma_px * z / fx)[:, None].repeat(1, 3)
    if opacity is None:
        opac = torch.ones(N, dtype=torch.float64)
    else:
        opac = opacity[0] + (opacity[1] - opacity[0]) * torch.rand(N, generator=g, dtype=torch.float64)
    rgbs = torch.rand(N, 3, generator=g, dtype=torch.float64)
    K = torch.tensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=torch.float64)
    return dict(means=means.to(dtype), quats=quats.to(dtype), scales=scales.to(dtype), opacities=opac.to(dtype),
                rgbs=rgbs.to(dtype), K=K.to(dtype), W=W, H=H)


def small_pose(rot_deg=0.5, trans=0.01, seed=7, dtype=torch.float64):
    """c2w = GT (identity) perturbed by rot_deg about a seeded axis and `trans` metres."""
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
    return c2w.to(dtype)


def sh_from_rgb(rgbs):
    sh = torch.zeros(rgbs.shape[0], 4, 3, dtype=rgbs.dtype)
    sh[:, 0, :] = (rgbs - 0.5) / 0.28209479177387814
    return sh


def frustum_clamp_scene(dtype=torch.float64):
    """12 large Gaussians centred up to 1.7x the half field of view off-axis (the clamped branch of the EWA Jacobian:
    gsplat limits x/z, y/z to 1.3 tan(fov/2)) plus one behind the camera, one beyond far_plane = 20 and one below
    radius_clip = 2.5.  64x48 image."""
    W, H, fx = 64, 48, 40.0
    K = torch.tensor([[fx, 0, 31.5], [0, fx, 23.5], [0, 0, 1]], dtype=torch.float64)
    g = torch.Generator().manual_seed(31)
    tanx, tany = 0.5 * W / fx, 0.5 * H / fx
    z = torch.full((12,), 2.0, dtype=torch.float64)
    side = torch.tensor([1.5, -1.5, 1.6, -1.7, 0.2, -0.3, 1.45, -1.55, 0.0, 0.1, 1.35, -1.4], dtype=torch.float64)
    means = torch.stack([side * tanx * z, torch.roll(side, 3) * tany * z, z], -1)
    scales = torch.full((12, 3), 0.9, dtype=torch.float64) * (0.7 + 0.6 * torch.rand(12, 3, generator=g, dtype=torch.float64))
    means = torch.cat([means, torch.tensor([[0.0, 0.0, -0.5], [0.1, 0.0, 50.0], [0.0, 0.1, 2.0]], dtype=torch.float64)])
    scales = torch.cat([scales, torch.tensor([[0.1] * 3, [0.1] * 3, [1e-3] * 3], dtype=torch.float64)])
    N = means.shape[0]
    quats = torch.randn(N, 4, generator=g, dtype=torch.float64)
    opac = 0.3 + 0.6 * torch.rand(N, generator=g, dtype=torch.float64)
    rgb = torch.rand(N, 3, generator=g, dtype=torch.float64)
    V = torch.linalg.inv(small_pose(2.0, 0.05, seed=4))
    v_render = torch.randn(H, W, 4, generator=g, dtype=torch.float64)
    v_alphas = torch.randn(H, W, generator=g, dtype=torch.float64)
    t = lambda x: x.to(dtype)  # noqa: E731
    return dict(means=t(means), quats=t(quats), scales=t(scales), opacities=t(opac), rgbs=t(rgb), K=t(K), V=t(V), W=W,
                H=H, tan=(tanx, tany), v_render=t(v_render), v_alphas=t(v_alphas),
                kw=dict(near_plane=0.01, far_plane=20.0, radius_clip=2.5))
