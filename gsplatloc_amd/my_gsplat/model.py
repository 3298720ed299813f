"""Mirror of /root/reference/src/my_gsplat/model.py (minus the nerfview viewer callback)."""
from dataclasses import dataclass

import torch
from torch import Tensor, nn
from torch.optim import Adam, Optimizer

from ..rendering import rasterization
from .geometry import construct_full_pose, init_gs_scales
from .transform import normalize_quaternion, quat_to_rotation_matrix, rotation_matrix_to_quaternion
from .utils import rgb_to_sh


@dataclass(frozen=True)
class CameraConfig:  # model.py:18-23
    trans_lr: float = 1e-3
    quat_lr: float = 5 * 1e-4
    quat_opt_reg: float = 1e-3
    trans_opt_reg: float = 1e-3


class CameraOptModule_quat_tans(nn.Module):
    """model.py:27-116: pose = (wxyz quaternion, translation), one Adam per parameter."""

    def __init__(self, init_pose: Tensor, *, config: CameraConfig = CameraConfig()):
        super().__init__()
        self.config = config
        self.quat_cur = nn.Parameter(rotation_matrix_to_quaternion(init_pose[:3, :3].contiguous()))
        self.t_cur = nn.Parameter(init_pose[:3, 3].clone())
        self.prev_quat = self.quat_cur.detach().clone()
        self.prev_t = self.t_cur.detach().clone()
        self.optimizers = self._create_optimizers()

    def update_pose(self, new_pose: Tensor | None = None):
        with torch.no_grad():
            if torch.is_tensor(new_pose) and new_pose.shape == (4, 4):
                self.quat_cur.data = rotation_matrix_to_quaternion(new_pose[:3, :3].contiguous())
                self.t_cur.data = new_pose[:3, 3]
                self.optimizers = self._create_optimizers()
            elif new_pose is None:
                self.quat_cur.data, self.t_cur.data = self.predict_next_pose()
            else:
                raise ValueError("fake new pose")

    def predict_next_pose(self):
        predicted_quaternion = normalize_quaternion(self.quat_cur + (self.quat_cur - self.prev_quat))
        predicted_translation = self.t_cur + (self.t_cur - self.prev_t)
        self.prev_quat, self.prev_t = self.quat_cur.detach().clone(), self.t_cur.detach().clone()
        return predicted_quaternion, predicted_translation

    def forward(self) -> Tensor:
        return construct_full_pose(quat_to_rotation_matrix(self.quat_cur), self.t_cur)

    def optimizer_step(self):
        for optimizer in self.optimizers:
            optimizer.step()

    def optimizer_clean(self):
        for optimizer in self.optimizers:
            optimizer.zero_grad(set_to_none=True)

    def _create_optimizers(self) -> list[Optimizer]:
        params = [("quat", self.quat_cur, self.config.quat_lr), ("trans", self.t_cur, self.config.trans_lr)]
        return [Adam([{"params": param, "lr": lr, "name": name}],
                     weight_decay=(self.config.quat_opt_reg if name == "quat" else self.config.trans_opt_reg))
                for name, param, lr in params]


@dataclass
class GsConfig:  # model.py:119-133
    init_opa: float = 1.0
    sparse_grad: bool = False
    packed: bool = False
    absgrad: bool = False
    antialiased: bool = False
    sh_degree: int = 1
    near_plane: float = 1e-2
    far_plane: float = 1e10


class GSModel(nn.Module):
    """model.py:136-215: one isotropic Gaussian per point, opacity logit(1.0), identity quaternions,
    SH degree 1 with only the DC term set."""

    def __init__(self, points: Tensor, colors: Tensor, *, config: GsConfig = GsConfig(), scales: Tensor | None = None):
        super().__init__()
        self.config = config
        self.means3d = points
        self.opacities = torch.logit(torch.full((points.shape[0],), self.config.init_opa, device=self.device))
        self.scales = init_gs_scales(points) if scales is None else scales
        self.quats = torch.tensor([1.0, 0.0, 0.0, 0.0], device=self.device).repeat(points.shape[0], 1)
        sh = torch.zeros((points.shape[0], (self.config.sh_degree + 1) ** 2, 3), device=self.device)
        sh[:, 0, :] = rgb_to_sh(colors)
        self.colors = colors
        self.sh0 = sh[:, :1, :]
        self.shN = sh[:, 1:, :]

    def __len__(self):
        return self.means3d.shape[0]

    def forward(self, camtoworlds: Tensor, Ks: Tensor, width: int, height: int, render_mode: str = "RGB+ED"):
        assert self.means3d.shape[0] == self.opacities.shape[0]
        opacities = torch.sigmoid(self.opacities)
        colors = torch.cat([self.sh0, self.shN], 1)
        return rasterization(
            means=self.means3d, quats=self.quats, scales=self.scales, opacities=opacities, colors=colors,
            sh_degree=self.config.sh_degree, viewmats=torch.linalg.inv(camtoworlds), Ks=Ks, width=width,
            height=height, packed=self.config.packed, absgrad=self.config.absgrad,
            sparse_grad=self.config.sparse_grad, far_plane=self.config.far_plane,
            near_plane=self.config.near_plane, render_mode=render_mode, rasterize_mode="classic")

    @property
    def device(self):
        return self.means3d.device
