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


def _camera_block(path: Path) -> dict:
    """The "camera" mapping of a json / yaml camera file."""
    text = path.read_text()
    if path.suffix.lower() in (".yaml", ".yml"):
        import yaml

        return dict(yaml.safe_load(text)["camera"])
    return dict(json.loads(text)["camera"])


class BaseDataset(Sequence):
    """Camera description shared by the readers (role of dataset.py:17-76): depth scale, optional distortion,
    optional crop_edge (the intrinsics and the image size shrink with the cropped border), K."""

    def __init__(self, input_folder: str, cfg_file: str):
        root, cam_file = Path(input_folder), Path(cfg_file)
        for what in (root, cam_file):
            if not what.exists():
                raise AssertionError(f"Path {what} does not exist.")
        self.input_folder = root
        cam = _camera_block(cam_file)
        border = int(cam.get("crop_edge", 0))
        if border:  # a border of `border` pixels is cut off every side: principal point and size move with it
            cam.update(h=cam["h"] - 2 * border, w=cam["w"] - 2 * border, cx=cam["cx"] - border, cy=cam["cy"] - border)
        self.cfg, self.crop_edge, self.scale = cam, border, cam["scale"]
        self.distortion = np.asarray(cam["distortion"]) if "distortion" in cam else None
        self.K = np.array([[cam["fx"], 0.0, cam["cx"]], [0.0, cam["fy"], cam["cy"]], [0.0, 0.0, 1.0]])

    def __getitem__(self, index):
        if isinstance(index, slice):
            return [self._get_one(i) for i in range(*index.indices(len(self)))]
        if not isinstance(index, (int, np.integer)):
            raise TypeError(f"frames are addressed by int or slice, not {type(index).__name__}")
        if not 0 <= index < len(self):  # (ValueError, as the reference's readers raise: callers iterate until it)
            raise ValueError(f"frame {index} does not exist: the sequence has {len(self)} frames")
        return self._get_one(int(index))

    def _crop(self, a: np.ndarray) -> np.ndarray:
        c = self.crop_edge
        return a[c:-c, c:-c] if c > 0 else a


class Replica(BaseDataset):
    """dataset.py:78-160: <root>/<name>/{results/frame*.jpg, results/depth*.png, traj.txt}."""

    def __init__(self, name: str = "room0", *, input_folder: Path | str = "datasets/Replica",
                 cfg_file: Path | str | None = None):
        self.name = name
        base = Path(input_folder)
        cam_file = base / "cam_params.json" if cfg_file is None else Path(cfg_file)
        super().__init__(str(base / name), str(cam_file))
        self._color_paths = sorted(self.input_folder.rglob("frame*.jpg"), key=_natural_key)
        self._depth_paths = sorted(self.input_folder.rglob("depth*.png"), key=_natural_key)
        if not self._color_paths or not self._depth_paths:
            raise FileNotFoundError(f"no frame*.jpg / depth*.png under {self.input_folder}")
        if len(self._color_paths) != len(self._depth_paths):
            raise ValueError(f"{len(self._color_paths)} colour images but {len(self._depth_paths)} depth images "
                             f"under {self.input_folder}")
        # traj.txt: one row-major 4x4 camera-to-world matrix per line
        rows = np.loadtxt(self.input_folder / "traj.txt", dtype=np.float64, ndmin=2)
        self._poses = [row.reshape(4, 4) for row in rows[: len(self._color_paths)]]

    def __str__(self):
        return f"Replica dataset: {self.name}\n in {self.input_folder}"

    def __len__(self):
        return len(self._color_paths)

    def _get_one(self, index: int) -> RGBDFrame:
        depth = _imread(self._depth_paths[index]).astype(np.float64) / self.scale
        rgb = _imread(self._color_paths[index]).astype(np.float64)[..., :3]
        return RGBDFrame(rgb, depth, self.K, self._poses[index])


def _nearest(stamps: np.ndarray, t: np.ndarray):
    """Per query time t[i]: index of the closest entry of `stamps` (the first one on a tie) and its distance."""
    gap = np.abs(stamps[None, :] - t[:, None])
    idx = gap.argmin(axis=1)
    return idx, gap[np.arange(len(t)), idx]


def _tum_pose(row: np.ndarray) -> np.ndarray:
    """TUM ground-truth row (tx ty tz qx qy qz qw) -> 4x4 camera-to-world."""
    x, y, z, w = row[3:7] / np.linalg.norm(row[3:7])
    c2w = np.eye(4)
    c2w[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
    c2w[:3, 3] = row[:3]
    return c2w


class TUM(BaseDataset):
    """Role of dataset.py:163-321: every colour image is paired with the depth image and the ground-truth pose closest
    in time (both within 0.08 s, else the image is dropped); the sequence is thinned to at most `frame_rate` frames
    per second; poses are expressed relative to the first kept frame; crop_edge from the camera file."""

    MAX_DT = 0.08

    def __init__(self, name: str = "freiburg1_desk", *, input_folder: Path | str = "datasets/TUM",
                 frame_rate: int = 32):
        self.name = "rgbd_dataset_" + name
        data_dir = Path(input_folder) / self.name
        super().__init__(str(data_dir), str(data_dir / "cam_params.json"))
        self._color_paths, self._depth_paths, self._poses = self._load_tum_data(frame_rate)

    def __str__(self):
        return f"TUM dataset: {self.name}\n in {self.input_folder}"

    def __len__(self):
        return len(self._color_paths)

    def _get_one(self, index: int) -> RGBDFrame:
        depth = self._crop(_imread(self._depth_paths[index]).astype(np.float32)) / self.scale
        rgb = self._crop(_imread(self._color_paths[index]).astype(np.float64)[..., :3])
        return RGBDFrame(rgb, depth, self.K, self._poses[index])

    @staticmethod
    def _table(path: Path, header_rows: int = 0) -> np.ndarray:
        """A TUM text table ("# comment" lines, blank-separated columns) as an array of strings."""
        return np.loadtxt(path, dtype=np.str_, comments="#", skiprows=header_rows, ndmin=2)

    def _load_tum_data(self, frame_rate: int):
        d = self.input_folder
        truth = d / "groundtruth.txt"
        images, depths = self._table(d / "rgb.txt"), self._table(d / "depth.txt")
        track = self._table(truth if truth.is_file() else d / "pose.txt", header_rows=1)
        t_img, t_dep, t_pose = (tab[:, 0].astype(np.float64) for tab in (images, depths, track))
        j, dt_depth = _nearest(t_dep, t_img)
        k, dt_pose = _nearest(t_pose, t_img)
        paired = np.flatnonzero((dt_depth < self.MAX_DT) & (dt_pose < self.MAX_DT))
        # thinning: a frame is kept when more than 1 / frame_rate seconds have passed since the last kept one
        kept, last_t = [], None
        for i in paired:
            if last_t is None or t_img[i] - last_t > 1.0 / frame_rate:
                kept.append(int(i))
                last_t = t_img[i]
        world = [_tum_pose(track[k[i], 1:8].astype(np.float64)) for i in kept]
        to_first = np.linalg.inv(world[0]) if world else None
        poses = [(np.eye(4) if n == 0 else to_first @ m).astype(np.float32) for n, m in enumerate(world)]
        return [d / images[i, 1] for i in kept], [d / depths[j[i], 1] for i in kept], poses


def get_data_set(name: Literal["TUM", "Replica"], room: str, **kw):
    readers = {"TUM": TUM, "Replica": Replica}
    if name not in readers:
        raise ValueError(f"unknown data set {name!r}: one of {sorted(readers)}")
    return readers[name](room, **kw)


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


def _now(device) -> float:
    import time
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
    return time.perf_counter()


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
        self.phase_seconds = None  # dict: per-phase wall time of __getitem__ (set by a profiling caller; costs two syncs)
        # Frame i + 1 of pair i is frame i of pair i + 1: keep the last two frames' device tensors, and decode the frame
        # the NEXT pair will need on a host thread while the tracker optimises this one (PNG / JPEG decoding releases the
        # GIL; reading was 14 of a frame's 85 ms at 640x480 -- the largest item outside the optimisation loop)
        self._frames = {}
        self._pending = {}
        self._pool = None

    def __len__(self):
        return len(self._data) - 1

    def _prefetch(self, index: int) -> None:
        if index in self._frames or index in self._pending or not 0 <= index < len(self._data):
            return
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="gsloc-reader")
        self._pending[index] = self._pool.submit(self._data.__getitem__, index)

    def _frame(self, index: int):
        if index in self._frames:
            return self._frames[index]
        fut = self._pending.pop(index, None)
        f = fut.result() if fut is not None else self._data[index]
        t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=self.device)  # noqa: E731
        depth, rgb, pose = t(f.depth), t(f.rgb), t(f.pose)
        out = (depth, rgb, pose, depth_to_points(depth, self.K), (rgb / 255.0).reshape(-1, 3))
        self._frames[index] = out
        for k in [k for k in self._frames if k < index - 1]:  # (a pair needs two frames: keep the last two)
            del self._frames[k]
        return out

    @torch.no_grad()
    def __getitem__(self, index: int) -> AlignData:
        if index >= len(self) or index < 0:
            raise IndexError(index)
        ph = self.phase_seconds
        t0 = _now(self.device) if ph is not None else 0.0
        t_depth, t_rgb, t_pose, t_pts, t_col = self._frame(index)
        s_depth, s_rgb, s_pose, s_pts, s_col = self._frame(index + 1)
        self._prefetch(index + 2)  # what pair index + 1 will need; decoded while this pair is being tracked
        t_pts = transform_points(t_pose, t_pts)
        s_pts = transform_points(t_pose, s_pts)
        pca_factor = torch.scalar_tensor(1.0, device=self.device)
        if self.normalize:
            t_pts, t_pose, s_pts, s_pose, pca_factor = normalize_pair(t_pts, t_pose, s_pts, s_pose)
            t1 = _now(self.device) if ph is not None else 0.0
            h, w = s_depth.shape
            s_depth = compute_depth_gt(s_pts, s_col, self.K.unsqueeze(0), c2w=t_pose.unsqueeze(0), height=h,
                                       width=w) / pca_factor
            s_depth = s_depth.reshape(h, w)
            if ph is not None:
                t2 = _now(self.device)
                ph["read_backproject_normalise"] = ph.get("read_backproject_normalise", 0.0) + (t1 - t0)
                ph["depth_gt_render"] = ph.get("depth_gt_render", 0.0) + (t2 - t1)
        elif ph is not None:
            ph["read_backproject_normalise"] = ph.get("read_backproject_normalise", 0.0) + (_now(self.device) - t0)
        return AlignData(pca_factor=pca_factor, colors=t_col, pixels=(s_rgb / 255.0).unsqueeze(0), tar_points=t_pts,
                         src_points=s_pts, src_depth=s_depth.unsqueeze(-1).unsqueeze(0), tar_c2w=t_pose,
                         src_c2w=s_pose, tar_nums=t_pts.shape[0])
