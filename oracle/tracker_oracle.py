"""CPU restatement of GsplatLoc's pose-tracking glue around the rasterizer.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PARITY UNPINNED for the
kornia 0.7.2 functions (kornia is not importable here; their published
definitions are restated and the reference's call sites are cited).

Follows:
  * pose parametrisation   /root/reference/src/my_gsplat/model.py:27-116
                           /root/reference/src/my_gsplat/transform.py:50-84
                           /root/reference/src/my_gsplat/geometry.py:12-20
  * losses                 /root/reference/src/my_gsplat/loss.py:10-59
  * inner loop             /root/reference/src/my_gsplat/gs_trainer_total.py:79-267
  * pose errors            /root/reference/src/eval/utils.py:122-168
  * per-frame construction /root/reference/src/my_gsplat/geometry.py:44-161
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from . import gsplat_oracle as G


# ----------------------------------------------------------- kornia restated
def normalize_quaternion(q: Tensor, eps: float = 1e-12) -> Tensor:
    """kornia.geometry.normalize_quaternion (call site transform.py:65)."""
    return F.normalize(q, p=2.0, dim=-1, eps=eps)


def quaternion_to_rotation_matrix(q: Tensor) -> Tensor:
    """kornia.geometry.quaternion_to_rotation_matrix, wxyz (transform.py:66)."""
    qn = normalize_quaternion(q)
    w, x, y, z = qn.unbind(-1)
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = torch.ones_like(w)
    m = torch.stack(
        [one - (tyy + tzz), txy - twz, txz + twy,
         txy + twz, one - (txx + tzz), tyz - twx,
         txz - twy, tyz + twx, one - (txx + tyy)], dim=-1)
    return m.reshape(q.shape[:-1] + (3, 3))


def rotation_matrix_to_quaternion(R: Tensor, eps: float = 1e-8) -> Tensor:
    """kornia.geometry.rotation_matrix_to_quaternion, returns wxyz (transform.py:84)."""
    tiny = torch.finfo(R.dtype).tiny

    def sdiv(n, d):
        return n / torch.clamp(d, min=tiny)

    v = R.reshape(R.shape[:-2] + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.chunk(v, 9, dim=-1)
    trace = m00 + m11 + m22

    def c0():
        sq = torch.sqrt(trace + 1.0 + eps) * 2.0
        return torch.cat((0.25 * sq, sdiv(m21 - m12, sq), sdiv(m02 - m20, sq), sdiv(m10 - m01, sq)), -1)

    def c1():
        sq = torch.sqrt(1.0 + m00 - m11 - m22 + eps) * 2.0
        return torch.cat((sdiv(m21 - m12, sq), 0.25 * sq, sdiv(m01 + m10, sq), sdiv(m02 + m20, sq)), -1)

    def c2():
        sq = torch.sqrt(1.0 + m11 - m00 - m22 + eps) * 2.0
        return torch.cat((sdiv(m02 - m20, sq), sdiv(m01 + m10, sq), 0.25 * sq, sdiv(m12 + m21, sq)), -1)

    def c3():
        sq = torch.sqrt(1.0 + m22 - m00 - m11 + eps) * 2.0
        return torch.cat((sdiv(m10 - m01, sq), sdiv(m02 + m20, sq), sdiv(m12 + m21, sq), 0.25 * sq), -1)

    w2 = torch.where(m11 > m22, c2(), c3())
    w1 = torch.where((m00 > m11) & (m00 > m22), c1(), w2)
    return torch.where(trace > 0.0, c0(), w1)


def sobel(x: Tensor, normalized: bool = True, eps: float = 1e-6) -> Tensor:
    """kornia.filters.sobel on [B,C,H,W] (call site loss.py:51-52): replicate pad 1,
    cross-correlate with the Sobel pair (/8 when normalised), sqrt(gx^2+gy^2+eps)."""
    kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]], dtype=x.dtype)
    ky = kx.t().contiguous()
    if normalized:
        kx, ky = kx / kx.abs().sum(), ky / ky.abs().sum()
    b, c, h, w = x.shape
    xp = F.pad(x.reshape(b * c, 1, h, w), (1, 1, 1, 1), mode="replicate")
    g = F.conv2d(xp, torch.stack([kx, ky])[:, None])
    gx, gy = g[:, 0], g[:, 1]
    return torch.sqrt(gx * gx + gy * gy + eps).reshape(b, c, h, w)


# -------------------------------------------------------- my_gsplat restated
def construct_full_pose(R: Tensor, t: Tensor) -> Tensor:
    """geometry.py:12-20."""
    pose = torch.eye(4, dtype=R.dtype)
    pose[:3, :3] = R
    pose[:3, 3] = t
    return pose


def camera_forward(quat: Tensor, t: Tensor) -> Tensor:
    """CameraOptModule_quat_tans.forward, model.py:79-82."""
    return construct_full_pose(quaternion_to_rotation_matrix(normalize_quaternion(quat)), t)


def compute_depth_loss(a: Tensor, b: Tensor) -> Tensor:
    """loss.py:25-26 (l1)."""
    return F.l1_loss(a, b)


def compute_silhouette_loss(a: Tensor, b: Tensor) -> Tensor:
    """loss.py:51-55 (l1 of Sobel magnitudes), inputs [B,H,W,1]."""
    return F.l1_loss(sobel(a.permute(0, 3, 1, 2)), sobel(b.permute(0, 3, 1, 2)))


def depth_to_normal(depth: Tensor, K: Tensor) -> Tensor:
    """geometry.py:164-197 with depth_to_points :138-161 (kornia.geometry.depth_to_3d_v2: pixel (u, v) on the integer
    grid -> depth * K^-1 (u, v, 1)): central differences of the back-projected surface with replicated borders,
    cross product, F.normalize.  depth [H,W] -> [H,W,3]."""
    H, W = depth.shape
    v, u = torch.meshgrid(torch.arange(H, dtype=depth.dtype), torch.arange(W, dtype=depth.dtype), indexing="ij")
    pts = torch.stack([(u - K[0, 2]) / K[0, 0] * depth, (v - K[1, 2]) / K[1, 1] * depth, depth], -1)[None]
    pp = F.pad(pts, (0, 0, 1, 1, 1, 1), mode="replicate")
    dx = pp[:, 1:-1, 2:, :] - pp[:, 1:-1, :-2, :]
    dy = pp[:, 2:, 1:-1, :] - pp[:, :-2, 1:-1, :]
    return F.normalize(torch.cross(dx, dy, dim=-1), p=2, dim=-1)[0]


def compute_normal_consistency_loss(depth_real: Tensor, depth_rendered: Tensor, K: Tensor) -> Tensor:
    """loss.py:62-101, loss_type "cosine", as coded: the maps are [H,W,3] and the similarity is taken over dim=1, i.e.
    along each image row per component."""
    return 1 - F.cosine_similarity(depth_to_normal(depth_real, K), depth_to_normal(depth_rendered, K), dim=1).mean()


def tracking_loss(depths: Tensor, depths_gt: Tensor, depth_lambda: float = 0.8, normal_lambda: float = 0.0,
                  K: Tensor = None):
    """gs_trainer_total.py:105-150.  depths, depths_gt: [1,H,W,1].  The normal term is the call the reference keeps
    commented out (:138-143; normal_lambda = 0 in data/base.py:28); evaluated only for a non-zero weight."""
    mask = (depths != 0).to(depths.dtype).detach()
    dl = compute_depth_loss(depths * mask, depths_gt * mask)
    sl = compute_silhouette_loss(depths * mask, depths_gt * mask)
    total = dl * depth_lambda + sl * (1 - depth_lambda - normal_lambda)
    if normal_lambda != 0.0:
        nl = compute_normal_consistency_loss((depths * mask)[0, :, :, 0], (depths_gt * mask)[0, :, :, 0], K.to(depths.dtype))
        total = total + nl * normal_lambda
    return total, dl, sl


def calculate_translation_error(est: Tensor, gt: Tensor) -> float:
    """eval/utils.py:122-141."""
    return float(torch.norm(est[:3, 3] - gt[:3, 3]))


def calculate_rotation_error(est: Tensor, gt: Tensor) -> float:
    """eval/utils.py:144-168: angle of R_est R_gt^T in degrees."""
    Rr = est[:3, :3] @ gt[:3, :3].T
    c = torch.clamp((torch.trace(Rr) - 1.0) / 2.0, -1.0, 1.0)
    return float(torch.acos(c) * 180.0 / math.pi)


def depth_to_points(depth: Tensor, K: Tensor) -> Tensor:
    """geometry.py:138-161 (kornia depth_to_3d_v2): integer pixel grid, row-major."""
    H, W = depth.shape
    v, u = torch.meshgrid(torch.arange(H, dtype=depth.dtype), torch.arange(W, dtype=depth.dtype), indexing="ij")
    x = (u - K[0, 2]) / K[0, 0] * depth
    y = (v - K[1, 2]) / K[1, 1] * depth
    return torch.stack([x, y, depth], dim=-1).reshape(-1, 3)


def knn_dists(points: Tensor, k: int, squared: bool) -> Tensor:
    """utils.py:16-22: k nearest (self included) distances via a KD-tree.
    small_gicp's batch_knn_search returns SQUARED distances."""
    from scipy.spatial import cKDTree

    p = points.detach().double().numpy()
    d, _ = cKDTree(p).query(p, k=k)
    d = torch.from_numpy(d).to(points.dtype)
    return d * d if squared else d


def init_gs_scales(points: Tensor, k: int = 5, eps: float = 1e-24, as_coded: bool = True) -> Tensor:
    """geometry.py:44-66.  as_coded=True feeds squared distances (the library's
    return value) through the reference's **2/mean/sqrt, i.e. sqrt(mean(d^4))."""
    d = knn_dists(points, k, squared=as_coded)[:, 1:]
    return torch.sqrt((d ** 2).mean(dim=-1) + eps)[:, None].repeat(1, 3)


def rgb_to_sh(rgb: Tensor) -> Tensor:
    """utils.py:53-55."""
    return (rgb - 0.5) / 0.28209479177387814


@dataclass
class TrackResult:
    losses: List[float] = field(default_factory=list)
    best_loss: float = float("inf")
    best_eT: float = float("inf")
    best_eR: float = float("inf")
    final_c2w: Optional[Tensor] = None
    steps: int = 0


def gs_forward(means, quats, scales, opacities, sh, c2w, K, W, H, render_mode="RGB+ED"):
    """GSModel.forward, model.py:180-215 (opacities already post-sigmoid)."""
    return G.rasterization(
        means=means, quats=quats, scales=scales, opacities=opacities, colors=sh, sh_degree=1,
        viewmats=torch.linalg.inv(c2w)[None], Ks=K[None], width=W, height=H, packed=False,
        absgrad=False, sparse_grad=False, far_plane=1e10, near_plane=1e-2, render_mode=render_mode,
        rasterize_mode="classic",
    )


class _CRasterize(torch.autograd.Function):
    """The rasterizer call of ``gs_forward`` through the C restatement (oracle/csrc/gsplat_oracle.c, float64)
    instead of the autograd oracle: same arithmetic (tests/test_c_oracle.py pins one against the other), fast
    enough for BASELINE.json's frame sizes.  Differentiable in the view matrix only -- all the tracker needs."""

    @staticmethod
    def forward(ctx, viewmat, means, quats, scales, opacities, sh, K, W, H, threads, precision="f64"):
        from . import c_oracle

        ctx.args = (means, quats, scales, opacities, sh, K, W, H, threads, precision)
        ctx.save_for_backward(viewmat)
        out = c_oracle.rasterization(means, quats, scales, opacities, sh, viewmat, K, W, H, sh_degree=1,
                                     render_mode="RGB+ED", precision=precision, threads=threads)
        return (torch.from_numpy(out["render"]).double()[None], torch.from_numpy(out["alphas"]).double()[None, ..., None])

    @staticmethod
    def backward(ctx, v_render, v_alphas):
        from . import c_oracle

        means, quats, scales, opacities, sh, K, W, H, threads, precision = ctx.args
        (viewmat,) = ctx.saved_tensors
        out = c_oracle.rasterization(means, quats, scales, opacities, sh, viewmat, K, W, H, sh_degree=1,
                                     render_mode="RGB+ED", v_render=v_render[0], v_alphas=v_alphas[0, ..., 0],
                                     precision=precision, threads=threads)
        return (torch.from_numpy(out["v_viewmat"]).double(),) + (None,) * 10


def gs_forward_c(means, quats, scales, opacities, sh, c2w, K, W, H, threads=None):
    """``gs_forward`` (RGB+ED) with the C restatement as the rasterizer; float64, gradient to ``c2w`` only."""
    render, alphas = _CRasterize.apply(torch.linalg.inv(c2w), means, quats, scales, opacities, sh, K, W, H, threads)
    return render, alphas, {}


def track_frame(
    means: Tensor, scales: Tensor, rgbs: Tensor, depth_gt: Tensor, K: Tensor, W: int, H: int,
    init_c2w: Tensor, gt_c2w: Tensor, max_steps: int = 200, patience: int = 200, min_step: int = 100,
    quat_lr: float = 5e-4, trans_lr: float = 1e-3, wd: float = 1e-3, verbose: bool = False,
    engine: str = "autograd", threads: Optional[int] = None, stop_after: Optional[int] = None,
) -> TrackResult:
    """Runner.train's per-frame body, gs_trainer_total.py:53-267.  ``engine="c"`` renders through the C
    restatement (full-size frames); ``stop_after`` ends the loop early without touching the schedule
    (gamma still follows ``max_steps``), for comparing the first iterations of a long run."""
    N = means.shape[0]
    dt = means.dtype
    quats = torch.tensor([1.0, 0, 0, 0], dtype=dt).repeat(N, 1)
    opac = torch.sigmoid(torch.logit(torch.full((N,), 1.0, dtype=dt)))
    sh = torch.zeros(N, 4, 3, dtype=dt)
    sh[:, 0, :] = rgb_to_sh(rgbs)
    q = torch.nn.Parameter(rotation_matrix_to_quaternion(init_c2w[:3, :3].contiguous()))
    t = torch.nn.Parameter(init_c2w[:3, 3].clone())
    opt_q = torch.optim.Adam([q], lr=quat_lr, weight_decay=wd)
    opt_t = torch.optim.Adam([t], lr=trans_lr, weight_decay=wd)
    gamma = 0.2 ** (1.0 / max_steps)
    sch = [torch.optim.lr_scheduler.ExponentialLR(o, gamma=gamma) for o in (opt_q, opt_t)]
    res = TrackResult()
    counter = 0
    for step in range(max_steps):
        opt_q.zero_grad(set_to_none=True)
        opt_t.zero_grad(set_to_none=True)
        c2w = camera_forward(q, t)
        if engine == "c":
            renders, _, _ = gs_forward_c(means, quats, scales, opac, sh, c2w, K, W, H, threads)
        else:
            renders, _, _ = gs_forward(means, quats, scales, opac, sh, c2w, K, W, H)
        depths = renders[..., 3:4]
        total, dl, sl = tracking_loss(depths, depth_gt)
        total.backward()
        lv = float(total.detach())
        res.losses.append(lv)
        eT = calculate_translation_error(c2w.detach(), gt_c2w)
        eR = calculate_rotation_error(c2w.detach(), gt_c2w)
        if step > min_step:
            if lv < res.best_loss:
                res.best_loss, res.best_eT, res.best_eR = lv, eT, eR
                counter = 0
            else:
                counter += 1
        if verbose:
            print(f"step {step} loss {lv:.6e} eT {eT:.3e} eR {eR:.3e}")
        res.steps = step + 1
        res.final_c2w = c2w.detach().clone()
        if counter >= patience or (stop_after is not None and step + 1 >= stop_after):
            break
        opt_q.step()
        opt_t.step()
        for s in sch:
            s.step()
    return res
