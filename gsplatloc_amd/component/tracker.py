"""Scan2ScanICP -- the reference's ICP baseline tracker (/root/reference/src/component/tracker.py:9-137) on
the in-repo small_gicp stand-in (gsplatloc_amd/small_gicp.py -> libgsloc_icp.so, CPU only).

Same constructor arguments, attributes and ``align`` behaviour as the reference's small_gicp backend:
every scan is registered against the previous one and ``T_world_camera`` accumulates the relative motion.
The open3d backend (COLORED_ICP / HYBRID, tracker.py:139-260, CUDA open3d) is not provided.
"""
from __future__ import annotations

from typing import Literal, Optional

import numpy as np

from .. import small_gicp


class Scan2ScanICP:
    def __init__(self, max_corresponding_distance: float = 0.1, voxel_downsampling_resolutions: float = 0.0,
                 knn: int = 20, num_threads: int = 32,
                 registration_type: Literal["ICP", "PLANE_ICP", "GICP", "COLORED_ICP", "HYBRID"] = "GICP",
                 implementation: Literal["small_gicp", "open3d"] = "small_gicp", error_threshold: float = 50.0):
        if implementation != "small_gicp" or registration_type in ("COLORED_ICP", "HYBRID"):
            raise NotImplementedError("only the small_gicp backend (ICP / PLANE_ICP / GICP) is provided; the "
                                      "reference's open3d CUDA backend is outside this build")
        self.voxel_downsampling_resolutions = voxel_downsampling_resolutions
        self.max_corresponding_distance = max_corresponding_distance
        self.num_threads = num_threads
        self.previous_pcd: Optional[small_gicp.PointCloud] = None
        self.previous_tree: Optional[small_gicp.KdTree] = None
        self.T_last_current = np.identity(4)
        self.T_world_camera = np.identity(4)
        self.registration_type = registration_type
        self.backend = implementation
        self.error_threshold = error_threshold
        self.knn = knn
        self.last_result: Optional[small_gicp.RegistrationResult] = None

    def align(self, raw_points, init_gt_pose=None, T_last_current=np.identity(4)):
        """tracker.py:64-84.  NB the reference's caller passes its initial guess as the *second* positional
        argument (experiment.py:108), i.e. into ``init_gt_pose``; that quirk is kept: after the first scan
        ``init_gt_pose`` is ignored and ``T_last_current`` seeds the registration."""
        return self.align_small_gicp(raw_points, init_gt_pose, T_last_current, self.knn)

    def align_small_gicp(self, raw_points, init_gt_pose=None, T_last_current=np.identity(4), knn: int = 20):
        """tracker.py:86-137."""
        if self.voxel_downsampling_resolutions > 0.0:
            downsampled, tree = small_gicp.preprocess_points(raw_points, self.voxel_downsampling_resolutions,
                                                             num_threads=self.num_threads, num_neighbors=knn)
        elif self.voxel_downsampling_resolutions == 0.0:
            downsampled = small_gicp.PointCloud(raw_points)
            tree = small_gicp.KdTree(downsampled, num_threads=self.num_threads)
            small_gicp.estimate_normals_covariances(downsampled, tree, num_neighbors=knn,
                                                    num_threads=self.num_threads)
        else:
            raise ValueError("voxel_downsampling_resolutions must greater than 0.0")
        if self.previous_pcd is None:  # first frame
            self.previous_pcd, self.previous_tree = downsampled, tree
            self.T_world_camera = init_gt_pose if init_gt_pose is not None else np.identity(4)
            return init_gt_pose
        result = small_gicp.align(self.previous_pcd, downsampled, self.previous_tree,
                                  init_T_target_source=T_last_current,
                                  max_correspondence_distance=self.max_corresponding_distance,
                                  registration_type=self.registration_type, num_threads=self.num_threads)
        self.last_result = result
        self.T_world_camera = self.T_world_camera @ result.T_target_source
        self.previous_pcd, self.previous_tree = downsampled, tree
        return self.T_world_camera
