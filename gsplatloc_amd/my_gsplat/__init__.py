"""Host-side mirror of GsplatLoc's ``src/my_gsplat`` package on top of the HIP rasterizer.

Same class / function names, arguments and semantics as the reference modules so that its
driver code reads unchanged:
  model.py     -> CameraConfig, CameraOptModule_quat_tans, GsConfig, GSModel
  geometry.py  -> construct_full_pose, transform_points, init_gs_scales, compute_depth_gt,
                  depth_to_points, depth_to_normal
  loss.py      -> compute_depth_loss, compute_silhouette_loss, compute_normal_consistency_loss
  transform.py -> quat_to_rotation_matrix, rotation_matrix_to_quaternion, rotation_6d_to_matrix,
                  matrix_to_rotation_6d
  utils.py     -> knn, remove_outliers, rgb_to_sh
  trainer.py   -> PoseTracker (the per-frame body of gs_trainer_total.Runner.train)
kornia, small_gicp, nerfview and W&B are not needed: the few functions the path takes from them
are restated here in PyTorch (plumbing around the HIP operators).
"""
from .geometry import (compute_depth_gt, construct_full_pose, depth_to_normal, depth_to_points,  # noqa: F401
                       init_gs_scales, transform_points)
from .loss import compute_depth_loss, compute_normal_consistency_loss, compute_silhouette_loss  # noqa: F401
from .model import CameraConfig, CameraOptModule_quat_tans, GsConfig, GSModel  # noqa: F401
from .trainer import PoseTracker, TrackerConfig, calculate_rotation_error, calculate_translation_error  # noqa: F401
from .transform import (matrix_to_rotation_6d, quat_to_rotation_matrix, rotation_6d_to_matrix,  # noqa: F401
                        rotation_matrix_to_quaternion)
from .utils import knn, remove_outliers, rgb_to_sh  # noqa: F401
