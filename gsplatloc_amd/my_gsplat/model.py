"""Pose parameters and Gaussian model of the tracker, with the public interface of
/root/reference/src/my_gsplat/model.py (CameraConfig :18-23, CameraOptModule_quat_tans :27-116, GsConfig
:119-133, GSModel :136-215): same constructor arguments, attribute names and results, so the reference's
trainer code runs on it unchanged.  The nerfview viewer hook is not part of the path and is left out.

Differences that do not change results:
* ``optimizers`` is the reference's list of two Adam objects (quaternion first, translation second:
  model.py:93-116), so the scheduler construction over ``optimizers[0]`` / ``optimizers[1]``
  (gs_trainer_total.py:65-72) runs unchanged;
* the activated opacities and the concatenated SH coefficients are constants of a frame: GSModel builds
  them once and rebuilds only if ``opacities`` / ``sh0`` / ``shN`` were replaced or modified in place
  (the reference re-runs sigmoid and cat on every iteration).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
from torch import Tensor, nn
from torch.optim import Adam, Optimizer

from ..rendering import rasterization
from .geometry import construct_full_pose, init_gs_scales
from .transform import normalize_quaternion, quat_to_rotation_matrix, rotation_matrix_to_quaternion
from .utils import rgb_to_sh


@dataclass(frozen=True)
class CameraConfig:
    trans_lr: float = 1e-3
    quat_lr: float = 5 * 1e-4
    quat_opt_reg: float = 1e-3
    trans_opt_reg: float = 1e-3


def _split_pose(pose: Tensor) -> Tuple[Tensor, Tensor]:
    """4x4 camera-to-world -> (wxyz quaternion [4], translation [3]), detached copies."""
    if not (torch.is_tensor(pose) and pose.shape == (4, 4)):
        raise ValueError("fake new pose")
    pose = pose.detach()
    return rotation_matrix_to_quaternion(pose[:3, :3].contiguous()), pose[:3, 3].clone()


class CameraOptModule_quat_tans(nn.Module):
    """Camera-to-world pose as a (not necessarily unit) wxyz quaternion and a translation; ``forward()`` is
    the 4x4 matrix of the normalised quaternion.  ``update_pose(None)`` extrapolates with a constant-velocity
    model from the pose at the previous call; ``update_pose(T)`` re-seeds the pose and the optimiser state."""

    def __init__(self, init_pose: Tensor, *, config: CameraConfig = CameraConfig()):
        super().__init__()
        self.config = config
        quat, trans = _split_pose(init_pose)
        self.quat_cur = nn.Parameter(quat)
        self.t_cur = nn.Parameter(trans)
        self.prev_quat, self.prev_t = quat.clone(), trans.clone()
        self.optimizers: List[Optimizer] = self._create_optimizers()

    # ---- pose
    def forward(self) -> Tensor:
        return construct_full_pose(quat_to_rotation_matrix(self.quat_cur), self.t_cur)

    def predict_next_pose(self) -> Tuple[Tensor, Tensor]:
        """Constant velocity in parameter space: x_next = x + (x - x_prev); remembers x as the new x_prev."""
        quat, trans = self.quat_cur.detach(), self.t_cur.detach()
        nxt = normalize_quaternion(quat + (quat - self.prev_quat)), trans + (trans - self.prev_t)
        self.prev_quat, self.prev_t = quat.clone(), trans.clone()
        return nxt

    def update_pose(self, new_pose: Optional[Tensor] = None) -> None:
        with torch.no_grad():
            if new_pose is None:
                quat, trans = self.predict_next_pose()
                self.quat_cur.data, self.t_cur.data = quat, trans
            else:
                quat, trans = _split_pose(new_pose)
                self.quat_cur.data, self.t_cur.data = quat, trans
                self.optimizers = self._create_optimizers()  # fresh moments for the new starting point

    # ---- optimiser
    def _create_optimizers(self) -> List[Optimizer]:
        cfg = self.config
        return [
            Adam([{"params": [self.quat_cur], "lr": cfg.quat_lr, "name": "quat"}], weight_decay=cfg.quat_opt_reg),
            Adam([{"params": [self.t_cur], "lr": cfg.trans_lr, "name": "trans"}], weight_decay=cfg.trans_opt_reg),
        ]

    def optimizer_step(self) -> None:
        for opt in self.optimizers:
            opt.step()

    def optimizer_clean(self) -> None:
        for opt in self.optimizers:
            opt.zero_grad(set_to_none=True)


@dataclass
class GsConfig:
    init_opa: float = 1.0
    sparse_grad: bool = False
    packed: bool = False
    absgrad: bool = False
    antialiased: bool = False
    sh_degree: int = 1  # degree of the spherical harmonics
    near_plane: float = 1e-2
    far_plane: float = 1e10


class GSModel(nn.Module):
    """One isotropic Gaussian per point: identity rotation, scale from the k-NN distances (or ``scales``),
    opacity logit(init_opa), SH of degree ``sh_degree`` with only the DC band set from the colours.
    ``forward`` renders through ``gsplat.rasterization``'s signature (here: the HIP implementation)."""

    def __init__(self, points: Tensor, colors: Tensor, *, config: GsConfig = GsConfig(), scales: Optional[Tensor] = None):
        super().__init__()
        self.config = config
        n, dev = points.shape[0], points.device
        self.means3d = points
        self.colors = colors
        self.scales = scales if scales is not None else init_gs_scales(points)
        self.quats = torch.zeros(n, 4, device=dev)
        self.quats[:, 0] = 1.0
        self.opacities = torch.logit(torch.full((n,), float(config.init_opa), device=dev))
        bands = (config.sh_degree + 1) ** 2
        self.sh0 = rgb_to_sh(colors).reshape(n, 1, 3).to(dev)
        self.shN = torch.zeros(n, bands - 1, 3, device=dev)
        self._activated = None  # (key, opacities after sigmoid, [N,K,3] coefficients)

    def __len__(self) -> int:
        return self.means3d.shape[0]

    @property
    def device(self) -> torch.device:
        return self.means3d.device

    def _render_constants(self) -> Tuple[Tensor, Tensor]:
        src = (self.opacities, self.sh0, self.shN)
        if any(t.requires_grad for t in src):  # differentiable inputs: a cached graph could not be reused
            return torch.sigmoid(self.opacities), torch.cat([self.sh0, self.shN], dim=1)
        key = tuple((id(t), t._version) for t in src)
        if self._activated is None or self._activated[0] != key:
            self._activated = (key, torch.sigmoid(self.opacities), torch.cat([self.sh0, self.shN], dim=1))
        return self._activated[1], self._activated[2]

    def forward(self, camtoworlds: Tensor, Ks: Tensor, width: int, height: int, render_mode: str = "RGB+ED"):
        cfg = self.config
        assert self.means3d.shape[0] == self.opacities.shape[0]
        opacities, coefficients = self._render_constants()
        return rasterization(means=self.means3d, quats=self.quats, scales=self.scales, opacities=opacities,
                             colors=coefficients, sh_degree=cfg.sh_degree, viewmats=torch.linalg.inv(camtoworlds),
                             Ks=Ks, width=width, height=height, packed=cfg.packed, absgrad=cfg.absgrad,
                             sparse_grad=cfg.sparse_grad, far_plane=cfg.far_plane, near_plane=cfg.near_plane,
                             render_mode=render_mode, rasterize_mode="classic")
