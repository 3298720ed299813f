"""Replica / TUM readers and the frame-pair Parser (mirror of /root/reference/src/data/dataset.py:17-383,
Image.py:14-35).  Arrays stay numpy until the Parser moves a pair to the device."""
from __future__ import annotations

import json
import re
from collections.abc import Sequence
from dataclasses import dataclass
from pathlib import Path
from typing import Literal

import numpy as np
import torch
from torch import Tensor

from ..my_gsplat.geometry import compute_depth_gt, depth_to_points, transform_points
from .normalize import normalize_pair


def _natural_key(p: Path):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", p.name)]


def _imread(path: Path) -> np.ndarray:
    from PIL import Image

    with Image.open(path) as im:
        return np.array(im)


@dataclass
class RGBDFrame:
    rgb: np.ndarray  # [H,W,3] 0..255
    depth: np.ndarray  # [H,W] metres
    K: np.ndarray  # [3,3]
    pose: np.ndarray  # [4,4] camera-to-world

    @property
    def points(self) -> np.ndarray:
        """[H*W,3] camera-frame back-projection of every pixel (RGBDImage.points, Image.py:111-118); pixels
        without depth stay at the origin, as in the reference -- consumers filter z > 0."""
        h, w = self.depth.shape
        v, u = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
        z = self.depth.astype(np.float64)
        x = (u - self.K[0, 2]) / self.K[0, 0] * z
        y = (v - self.K[1, 2]) / self.K[1, 1] * z
        return np.stack([x, y, z], axis=-1).reshape(-1, 3)


class BaseDataset(Sequence):
    """dataset.py:17-76: camera config (json/yaml), optional crop_edge, K."""

    def __init__(self, input_folder: str, cfg_file: str):
        assert Path(input_folder).exists(), f"Path {input_folder} does not exist."
        assert Path(cfg_file).exists(), f"Path {cfg_file} does not exist."
        self.input_folder = Path(input_folder)
        cfg_path = Path(cfg_file)
        with open(cfg_path) as f:
            if cfg_path.suffix in (".yaml", ".yml"):
                import yaml
                cfg = yaml.safe_load(f)
            else:
                cfg = json.load(f)
        self.cfg = cfg["camera"]
        self.scale = self.cfg["scale"]
        self.distortion = np.array(self.cfg["distortion"]) if "distortion" in self.cfg else None
        self.crop_edge = self.cfg.get("crop_edge", 0)
        if self.crop_edge:
            self.cfg["h"] -= 2 * self.crop_edge
            self.cfg["w"] -= 2 * self.crop_edge
            self.cfg["cx"] -= self.crop_edge
            self.cfg["cy"] -= self.crop_edge
        self.K = np.eye(3)
        self.K[0, 0], self.K[1, 1], self.K[0, 2], self.K[1, 2] = (self.cfg["fx"], self.cfg["fy"], self.cfg["cx"],
                                                                   self.cfg["cy"])

    def __getitem__(self, index):
        if isinstance(index, int):
            if index >= len(self) or index < 0:
                raise ValueError(f"Index {index} out of range (0 to {len(self) - 1})")
            return self._get_one(index)
        elif isinstance(index, slice):
            return [self._get_one(i) for i in range(*index.indices(len(self)))]
        raise TypeError(f"index must be int or slice but now is {type(index)}")

    def _crop(self, a: np.ndarray) -> np.ndarray:
        c = self.crop_edge
        return a[c:-c, c:-c] if c > 0 else a


class Replica(BaseDataset):
    """dataset.py:78-160: <root>/<name>/{results/frame*.jpg, results/depth*.png, traj.txt}."""

    def __init__(self, name: str = "room0", *, input_folder: Path | str = "datasets/Replica",
                 cfg_file: Path | str | None = None):
        self.name = name
        input_folder = Path(input_folder)
        cfg_file = Path(cfg_file) if cfg_file is not None else input_folder / "cam_params.json"
        super().__init__((input_folder / name).as_posix(), cfg_file.as_posix())
        self._color_paths = sorted(self.input_folder.rglob("frame*.jpg"), key=_natural_key)
        self._depth_paths = sorted(self.input_folder.rglob("depth*.png"), key=_natural_key)
        if len(self._color_paths) == 0 or len(self._depth_paths) == 0:
            raise FileNotFoundError(f"No images found in {self.input_folder}. Please check the path.")
        if len(self._color_paths) != len(self._depth_paths):
            raise ValueError(f"Number of color and depth images do not match in {self.input_folder}.")
        self._num_img = len(self._color_paths)
        with open(self.input_folder / "traj.txt") as f:
            lines = f.readlines()
        self._poses = [np.array(list(map(float, lines[i].split()))).reshape(4, 4) for i in range(self._num_img)]

    def __str__(self):
        return f"Replica dataset: {self.name}\n in {self.input_folder}"

    def __len__(self):
        return self._num_img

    def _get_one(self, index: int) -> RGBDFrame:
        depth = _imread(self._depth_paths[index]).astype(np.float64) / self.scale
        rgb = _imread(self._color_paths[index]).astype(np.float64)[..., :3]
        return RGBDFrame(rgb, depth, self.K, self._poses[index])


class TUM(BaseDataset):
    """dataset.py:163-321: timestamp association (max_dt 0.08 s), frame-rate thinning, poses relative to
    the first frame, crop_edge."""

    def __init__(self, name: str = "freiburg1_desk", *, input_folder: Path | str = "datasets/TUM",
                 frame_rate: int = 32):
        self.name = "rgbd_dataset_" + name
        data_dir = Path(input_folder) / self.name
        super().__init__(data_dir.as_posix(), (data_dir / "cam_params.json").as_posix())
        self._color_paths, self._depth_paths, self._poses = self._load_tum_data(frame_rate)
        self._num_img = len(self._color_paths)

    def __str__(self):
        return f"TUM dataset: {self.name}\n in {self.input_folder}"

    def __len__(self):
        return self._num_img

    def _get_one(self, index: int) -> RGBDFrame:
        depth = self._crop(_imread(self._depth_paths[index]).astype(np.float32)) / self.scale
        rgb = self._crop(_imread(self._color_paths[index]).astype(np.float64)[..., :3])
        return RGBDFrame(rgb, depth, self.K, self._poses[index])

    @staticmethod
    def _parse_list(filepath: Path, skiprows: int = 0) -> np.ndarray:
        return np.loadtxt(filepath, delimiter=" ", dtype=np.str_, skiprows=skiprows, comments="#")

    @staticmethod
    def _associate_frames(tstamp_image, tstamp_depth, tstamp_pose, max_dt: float = 0.08):
        associations = []
        for i, t in enumerate(tstamp_image):
            j = int(np.argmin(np.abs(tstamp_depth - t)))
            k = int(np.argmin(np.abs(tstamp_pose - t)))
            if np.abs(tstamp_depth[j] - t) < max_dt and np.abs(tstamp_pose[k] - t) < max_dt:
                associations.append((i, j, k))
        return associations

    @staticmethod
    def _get_frame_indices(associations, tstamp_image, frame_rate: int):
        indices = [0]
        for i in range(1, len(associations)):
            if tstamp_image[associations[i][0]] - tstamp_image[associations[indices[-1]][0]] > 1.0 / frame_rate:
                indices.append(i)
        return indices

    @staticmethod
    def _pose_matrix_from_quaternion(pvec: np.ndarray) -> np.ndarray:
        from scipy.spatial.transform import Rotation

        pose = np.eye(4)
        pose[:3, :3] = Rotation.from_quat(pvec[3:]).as_matrix()
        pose[:3, 3] = pvec[:3]
        return pose

    def _load_tum_data(self, frame_rate: int):
        d = self.input_folder
        pose_list = d / ("groundtruth.txt" if (d / "groundtruth.txt").is_file() else "pose.txt")
        image_data = self._parse_list(d / "rgb.txt")
        depth_data = self._parse_list(d / "depth.txt")
        pose_data = self._parse_list(pose_list, skiprows=1)
        pose_vecs = pose_data[:, 1:].astype(np.float64)
        t_img, t_dep, t_pose = (a[:, 0].astype(np.float64) for a in (image_data, depth_data, pose_data))
        associations = self._associate_frames(t_img, t_dep, t_pose)
        color_paths, depth_paths, poses = [], [], []
        inv_pose = None
        for ix in self._get_frame_indices(associations, t_img, frame_rate):
            i, j, k = associations[ix]
            color_paths.append(d / image_data[i, 1])
            depth_paths.append(d / depth_data[j, 1])
            c2w = self._pose_matrix_from_quaternion(pose_vecs[k])
            if inv_pose is None:
                inv_pose = np.linalg.inv(c2w)
                c2w = np.eye(4)
            else:
                c2w = inv_pose @ c2w
            poses.append(c2w.astype(np.float32))
        return color_paths, depth_paths, poses


def get_data_set(name: Literal["TUM", "Replica"], room: str, **kw):
    if name == "TUM":
        return TUM(room, **kw)
    elif name == "Replica":
        return Replica(room, **kw)
    raise ValueError("data set name should be in ['TUM,Replica']")


@dataclass
class AlignData:
    """data/base.py:109-125."""
    pca_factor: Tensor
    colors: Tensor  # [N,3]
    pixels: Tensor  # [1,H,W,3]
    tar_points: Tensor  # [N,3]
    src_points: Tensor
    src_depth: Tensor  # [1,H,W,1]
    tar_c2w: Tensor
    src_c2w: Tensor
    tar_nums: int


class Parser:
    """dataset.py:333-383: frame pair (i, i+1) -> tracker inputs.  Both clouds are placed with the TARGET
    pose (dataset.py:349-350); with normalize=True the pair is moved to the target cloud's PCA frame and
    the query depth is re-rendered ("ED") from the source cloud at the target pose."""

    def __init__(self, data_set: Literal["Replica", "TUM"] = "Replica", name: str = "room0", normalize: bool = False,
                 device="cuda", **dataset_kw):
        self._data = get_data_set(data_set, name, **dataset_kw)
        self.device = torch.device(device)
        self.K = torch.as_tensor(self._data.K, dtype=torch.float32, device=self.device)
        self.normalize = normalize

    def __len__(self):
        return len(self._data) - 1

    def _frame(self, index: int):
        f = self._data[index]
        t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=self.device)  # noqa: E731
        depth, rgb, pose = t(f.depth), t(f.rgb), t(f.pose)
        return depth, rgb, pose, depth_to_points(depth, self.K), (rgb / 255.0).reshape(-1, 3)

    @torch.no_grad()
    def __getitem__(self, index: int) -> AlignData:
        if index >= len(self) or index < 0:
            raise IndexError(index)
        t_depth, t_rgb, t_pose, t_pts, t_col = self._frame(index)
        s_depth, s_rgb, s_pose, s_pts, s_col = self._frame(index + 1)
        t_pts = transform_points(t_pose, t_pts)
        s_pts = transform_points(t_pose, s_pts)
        pca_factor = torch.scalar_tensor(1.0, device=self.device)
        if self.normalize:
            t_pts, t_pose, s_pts, s_pose, pca_factor = normalize_pair(t_pts, t_pose, s_pts, s_pose)
            h, w = s_depth.shape
            s_depth = compute_depth_gt(s_pts, s_col, self.K.unsqueeze(0), c2w=t_pose.unsqueeze(0), height=h,
                                       width=w) / pca_factor
            s_depth = s_depth.reshape(h, w)
        return AlignData(pca_factor=pca_factor, colors=t_col, pixels=(s_rgb / 255.0).unsqueeze(0), tar_points=t_pts,
                         src_points=s_pts, src_depth=s_depth.unsqueeze(-1).unsqueeze(0), tar_c2w=t_pose,
                         src_c2w=s_pose, tar_nums=t_pts.shape[0])
