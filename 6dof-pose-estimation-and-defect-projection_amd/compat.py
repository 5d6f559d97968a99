"""Drop-in surface: `from pedp_hip.compat import *` where run.py says `from src import *`
(run.py:3, src/__init__.py:1-4), plus the `mycpp` slot (Utils.py:45-48, estimater.py:118).

The hot-path names, the global registration that precedes them under
`determine_pose(icp=True)`, and the message format to the viewer thread (`update_dash_data`); the Dash app, sensor and learned-model code stay the reference's own.
"""
import numpy as np

from . import _lib
from .geometry import (KDTreeSearchParamHybrid, LineSet, PinholeCameraIntrinsic, PointCloud,  # noqa: F401
                       RegistrationResult, TriangleMesh)
from .icp_refine import (background_removal, compute_average_normal, determine_pose, estimate_normals,  # noqa: F401
                         execute_global_registration, filter_largest_cluster, run_icp,
                         flip_plane_normal_if_needed, improve_result, perform_plane_segmentation,
                         predict_z_axis_adjustment, preprocess_source, preprocess_target, refine_pose_with_icp,
                         refine_registration, remove_plane, remove_points_below_plane, remove_statistical_outliers,
                         transform_object)
from .ray_projection import (align_to_surface, calc_coordinates, compute_rays, create_intersection_pcd,  # noqa: F401
                             heatmap_to_point3d, heatmap_to_points, intersect_rays_with_mesh, load_extrinsics,
                             pcd_from_point3d, project_debug_rays, ray_tracing)
from .registration import (CorrespondenceCheckerBasedOnDistance, CorrespondenceCheckerBasedOnEdgeLength,  # noqa: F401
                           CorrespondenceCheckerBasedOnNormal, Feature, ICPConvergenceCriteria,
                           RANSACConvergenceCriteria, TransformationEstimationPointToPlane,
                           TransformationEstimationPointToPoint, compute_fpfh_feature, get_rotation_matrix_from_xyz,
                           registration_icp, registration_ransac_based_on_feature_matching)


def cluster_poses(angle_diff, dist_diff, poses_in, symmetry_tfs):
    """mycpp.cluster_poses(angle_diff_deg, dist_diff, poses[N,4,4], symmetry_tfs[S,4,4]) ->
    list of kept 4x4 float32 poses (mycpp/src/app/pybind_api.cpp:24-68).  Prints the two
    lines the C++ prints (:26, :66)."""
    poses = np.ascontiguousarray(poses_in, dtype=np.float32).reshape(-1, 4, 4)
    print(f"num original candidates = {len(poses)}")
    keep = _lib.cluster_poses(angle_diff, dist_diff, poses, symmetry_tfs)
    print(f"num of pose after clustering: {len(keep)}")
    return [poses[k].copy() for k in keep]


from .depth_filters import bilateral_filter_depth, depth2xyzmap, depth2xyzmap_batch, erode_depth  # noqa: E402
from .viewer_wire import update_dash_data  # noqa: E402  (web_vis.py:203-217: the message to the viewer thread)


class _MyCpp:
    """`mycpp = pedp_hip.compat.mycpp` gives estimater.py its `mycpp.cluster_poses`."""
    cluster_poses = staticmethod(cluster_poses)


mycpp = _MyCpp()

__all__ = [
    "refine_registration", "improve_result", "predict_z_axis_adjustment", "refine_pose_with_icp", "determine_pose",
    "preprocess_source", "preprocess_target", "transform_object", "execute_global_registration", "run_icp",
    "perform_plane_segmentation", "flip_plane_normal_if_needed", "remove_plane", "remove_points_below_plane",
    "background_removal", "filter_largest_cluster", "remove_statistical_outliers", "estimate_normals",
    "compute_average_normal", "KDTreeSearchParamHybrid",
    "heatmap_to_points", "compute_rays", "intersect_rays_with_mesh", "create_intersection_pcd",
    "project_debug_rays", "load_extrinsics", "ray_tracing",
    "heatmap_to_point3d", "pcd_from_point3d", "calc_coordinates", "align_to_surface",
    "erode_depth", "bilateral_filter_depth", "depth2xyzmap", "depth2xyzmap_batch",
    "registration_icp", "TransformationEstimationPointToPlane", "TransformationEstimationPointToPoint",
    "ICPConvergenceCriteria", "get_rotation_matrix_from_xyz",
    "compute_fpfh_feature", "Feature", "registration_ransac_based_on_feature_matching", "RANSACConvergenceCriteria",
    "CorrespondenceCheckerBasedOnEdgeLength", "CorrespondenceCheckerBasedOnDistance", "CorrespondenceCheckerBasedOnNormal",
    "PointCloud", "TriangleMesh", "LineSet", "PinholeCameraIntrinsic", "RegistrationResult",
    "cluster_poses", "mycpp", "update_dash_data",
]
