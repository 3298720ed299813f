"""small_gicp-shaped Python API over libgsloc_icp.so (C ABI: include/gsloc_icp.h) -- the CPU ICP baseline.

The reference imports the third-party ``small_gicp`` package at /root/reference/src/component/tracker.py:5 and
/root/reference/src/my_gsplat/utils.py:3; this module offers the calls those two files make, with the same
names and argument meaning, so ``import gsplatloc_amd.small_gicp as small_gicp`` is a drop-in for them:

    PointCloud(points)                                   tracker.py:106, utils.py:19
    KdTree(cloud, num_threads).batch_knn_search(q, k)    tracker.py:107, utils.py:20-21
    estimate_normals_covariances(cloud, tree, num_neighbors, num_threads)     tracker.py:108-110
    preprocess_points(points, downsampling_resolution, num_neighbors, num_threads)   tracker.py:98-103
    align(target, source, target_tree, init_T_target_source, max_correspondence_distance,
          registration_type, num_threads, max_iterations) -> RegistrationResult       tracker.py:124-133

Host code only (g++/OpenMP); it is the comparison baseline of BASELINE.json configs[0], never on the GPU path.
PARITY UNPINNED: small_gicp's sources and outputs are absent from the reference tree; the algorithm is restated
from its published design (see the header).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, c_char_p, c_double, c_int, c_int32, c_int64, c_void_p
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsloc_icp.so")
_lib: Optional[ctypes.CDLL] = None

_TYPES = {"ICP": 0, "PLANE_ICP": 1, "GICP": 2}


class _Result(Structure):
    _fields_ = [("T", c_double * 16), ("H", c_double * 36), ("b", c_double * 6), ("error", c_double),
                ("converged", c_int32), ("iterations", c_int32), ("num_inliers", c_int64)]


_SIGNATURES = {
    "gsl_icp_version": (c_char_p, []),
    "gsl_icp_cloud_create": (c_void_p, [c_void_p, c_int64, c_int]),
    "gsl_icp_cloud_destroy": (None, [c_void_p]),
    "gsl_icp_cloud_size": (c_int64, [c_void_p]),
    "gsl_icp_cloud_read": (c_int, [c_void_p, c_int, c_void_p]),
    "gsl_icp_build_tree": (c_int, [c_void_p, c_int]),
    "gsl_icp_knn": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_int]),
    "gsl_icp_estimate_normals_covariances": (c_int, [c_void_p, c_int, c_int]),
    "gsl_icp_voxel_downsample": (c_void_p, [c_void_p, c_double, c_int]),
    "gsl_icp_align": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_int, c_int, c_int, POINTER(_Result)]),
}


def exported_symbols():
    return sorted(_SIGNATURES)


def library_path() -> str:
    return _LIB_PATH


def build_library(verbose: bool = False) -> str:
    """g++ -O3 -fopenmp -shared csrc_host/icp.cpp (in-tree)."""
    res = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc_host")], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout, res.stderr)
    if res.returncode != 0:
        raise RuntimeError("building libgsloc_icp.so failed:\n" + res.stderr[-4000:])
    return _LIB_PATH


def load_library() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} not found: run `make -C gsplatloc_amd/csrc_host` "
                               "(or __graft_entry__.build())")
        lib = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _check(status: int, what: str) -> None:
    if status != 0:
        reason = {-1: "bad argument", -2: "the cloud has no kd-tree", -3: "normals/covariances missing"}.get(status, "?")
        raise RuntimeError(f"{what} failed: {reason} ({status})")


def _rows(a, min_cols: int = 3) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if a.ndim != 2 or a.shape[1] < min_cols:
        raise ValueError(f"expected an [n, >={min_cols}] array, got {a.shape}")
    return a


class PointCloud:
    """Points (+ normals and covariances once estimated).  ``points`` [n, 3|4|...]: the first three columns."""

    def __init__(self, points=None, _handle: Optional[int] = None):
        self._lib = load_library()
        if _handle is not None:
            self._h = _handle
        else:
            a = _rows(points if points is not None else np.zeros((0, 3)))
            self._h = self._lib.gsl_icp_cloud_create(a.ctypes.data, a.shape[0], a.shape[1])
        if not self._h:
            raise RuntimeError("gsl_icp_cloud_create failed")
        self._has_tree = False

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.gsl_icp_cloud_destroy(h)

    def size(self) -> int:
        return int(self._lib.gsl_icp_cloud_size(self._h))

    __len__ = size

    def _read(self, what: int, cols: int) -> np.ndarray:
        out = np.empty((self.size(), cols), dtype=np.float64)
        _check(self._lib.gsl_icp_cloud_read(self._h, what, out.ctypes.data), "gsl_icp_cloud_read")
        return out

    def points(self) -> np.ndarray:
        """[n,4] homogeneous, as small_gicp returns them."""
        return np.concatenate([self._read(0, 3), np.ones((self.size(), 1))], axis=1)

    def normals(self) -> np.ndarray:
        return np.concatenate([self._read(1, 3), np.zeros((self.size(), 1))], axis=1)

    def covs(self) -> np.ndarray:
        """[n,4,4] with the 3x3 covariance in the upper-left block."""
        c = self._read(2, 9).reshape(-1, 3, 3)
        out = np.zeros((c.shape[0], 4, 4))
        out[:, :3, :3] = c
        return out


class KdTree:
    """Nearest-neighbour index over a PointCloud (built inside the cloud's native object)."""

    def __init__(self, cloud: PointCloud, num_threads: int = 1):
        if not isinstance(cloud, PointCloud):
            cloud = PointCloud(cloud)
        self.cloud = cloud
        _check(cloud._lib.gsl_icp_build_tree(cloud._h, int(num_threads)), "gsl_icp_build_tree")
        cloud._has_tree = True

    def batch_knn_search(self, queries, k: int, num_threads: int = 1) -> Tuple[np.ndarray, np.ndarray]:
        """(indices [m,k] int64, squared distances [m,k]) ascending; self matches included."""
        q = _rows(queries)
        idx = np.empty((q.shape[0], k), dtype=np.int64)
        d2 = np.empty((q.shape[0], k), dtype=np.float64)
        _check(self.cloud._lib.gsl_icp_knn(self.cloud._h, q.ctypes.data, q.shape[0], q.shape[1], int(k),
                                           idx.ctypes.data, d2.ctypes.data, int(num_threads)), "gsl_icp_knn")
        return idx, d2

    def nearest_neighbor_search(self, point) -> Tuple[bool, int, float]:
        idx, d2 = self.batch_knn_search(np.asarray(point, dtype=np.float64)[None, :], 1)
        return bool(idx[0, 0] >= 0), int(idx[0, 0]), float(d2[0, 0])


def estimate_normals_covariances(points: PointCloud, tree: Optional[KdTree] = None, num_neighbors: int = 20,
                                 num_threads: int = 1) -> None:
    if tree is None or not points._has_tree:
        KdTree(points, num_threads)
    _check(points._lib.gsl_icp_estimate_normals_covariances(points._h, int(num_neighbors), int(num_threads)),
           "gsl_icp_estimate_normals_covariances")


estimate_covariances = estimate_normals = estimate_normals_covariances


def voxelgrid_sampling(points, downsampling_resolution: float, num_threads: int = 1) -> PointCloud:
    cloud = points if isinstance(points, PointCloud) else PointCloud(points)
    h = cloud._lib.gsl_icp_voxel_downsample(cloud._h, float(downsampling_resolution), int(num_threads))
    if not h:
        raise RuntimeError("gsl_icp_voxel_downsample failed (resolution must be > 0)")
    return PointCloud(_handle=h)


def preprocess_points(points, downsampling_resolution: float = 0.25, num_neighbors: int = 10,
                      num_threads: int = 1) -> Tuple[PointCloud, KdTree]:
    """Voxel-average, build the tree, estimate normals and covariances."""
    down = voxelgrid_sampling(points, downsampling_resolution, num_threads)
    tree = KdTree(down, num_threads)
    estimate_normals_covariances(down, tree, num_neighbors, num_threads)
    return down, tree


class RegistrationResult:
    def __init__(self, r: _Result):
        self.T_target_source = np.array(r.T, dtype=np.float64).reshape(4, 4)
        self.H = np.array(r.H, dtype=np.float64).reshape(6, 6)
        self.b = np.array(r.b, dtype=np.float64)
        self.error = float(r.error)
        self.converged = bool(r.converged)
        self.iterations = int(r.iterations)
        self.num_inliers = int(r.num_inliers)

    def __repr__(self):
        return (f"RegistrationResult(converged={self.converged}, iterations={self.iterations}, "
                f"num_inliers={self.num_inliers}, error={self.error:.6g})")


def align(target: PointCloud, source: PointCloud, target_tree: Optional[KdTree] = None,
          init_T_target_source=None, max_correspondence_distance: float = 1.0, registration_type: str = "GICP",
          num_threads: int = 1, max_iterations: int = 20) -> RegistrationResult:
    """Register ``source`` onto ``target``.  registration_type: "ICP" | "PLANE_ICP" | "GICP"."""
    if registration_type not in _TYPES:
        raise ValueError(f"registration_type must be one of {sorted(_TYPES)} (got {registration_type!r})")
    if target_tree is None and not target._has_tree:
        KdTree(target, num_threads)
    T0 = np.ascontiguousarray(np.eye(4) if init_T_target_source is None else init_T_target_source, dtype=np.float64)
    if T0.shape != (4, 4):
        raise ValueError("init_T_target_source must be 4x4")
    res = _Result()
    _check(target._lib.gsl_icp_align(target._h, source._h, T0.ctypes.data, float(max_correspondence_distance),
                                     _TYPES[registration_type], int(max_iterations), int(num_threads),
                                     ctypes.byref(res)), "gsl_icp_align")
    return RegistrationResult(res)
