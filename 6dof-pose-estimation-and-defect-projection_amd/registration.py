"""`o3d.pipelines.registration` surface the reference touches, backed by libpedp_hip.so.

Reference call sites (src/pose_estimation.py):
    :519-521  registration_icp(source, target, distance_threshold, transformation,
                               TransformationEstimationPointToPlane())
    :654-660  ... same with ICPConvergenceCriteria(max_iteration=1)
    :584      o3d.geometry.get_rotation_matrix_from_xyz([a, b, c])
Names, argument order and defaults are Open3D 0.18's; errors are RuntimeError like
Open3D's (a target without normals under point-to-plane).
"""
import numpy as np

from . import _lib
from .geometry import RegistrationResult, normals_of, points_of


class TransformationEstimationPointToPlane:
    """Reference default (pose_estimation.py:521, :658)."""
    code = _lib.POINT_TO_PLANE

    def __init__(self, kernel=None):
        if kernel is not None:
            raise NotImplementedError("robust kernels are not used by the reference and are not built")


class TransformationEstimationPointToPoint:
    """Kabsch / Umeyama without scaling (north_star variant; reference uses it only in the
    RANSAC global registration, pose_estimation.py:486)."""
    code = _lib.POINT_TO_POINT

    def __init__(self, with_scaling=False):
        if with_scaling:
            raise NotImplementedError("with_scaling=True is not used by the reference and is not built")
        self.with_scaling = False


class ICPConvergenceCriteria:
    def __init__(self, relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30):
        self.relative_fitness = relative_fitness
        self.relative_rmse = relative_rmse
        self.max_iteration = max_iteration


def get_rotation_matrix_from_xyz(rotation):
    """Rx(a) @ Ry(b) @ Rz(c), as Open3D's geometry helper (used at pose_estimation.py:584)."""
    a, b, c = (float(v) for v in rotation)
    ca, sa, cb, sb, cc, sc = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    return np.array([
        [cb * cc, -cb * sc, sb],
        [sa * sb * cc + ca * sc, -sa * sb * sc + ca * cc, -sa * cb],
        [-ca * sb * cc + sa * sc, ca * sb * sc + sa * cc, ca * cb],
    ])


def upload(cloud, ctx=None):
    """Device copy of a PointCloud-like object; pass the returned handle to registration_icp
    when the same cloud is registered many times (improve_result does ~50 calls per frame)."""
    if isinstance(cloud, _lib.Cloud):
        return cloud
    ctx = ctx or _lib.default_context()
    return _lib.Cloud(ctx, points_of(cloud), normals_of(cloud))


def registration_icp(source, target, max_correspondence_distance, init=None, estimation_method=None,
                     criteria=None, ctx=None, want_correspondences=False, allreduce=None, n_source_global=0):
    """ICP registration on the GPU.  `init` maps source into the target frame (the reference
    passes inv(model->scene), pose_estimation.py:572, :657).  Returns a RegistrationResult;
    correspondence_set is filled only on request (the reference never reads it)."""
    ctx = ctx or (source.ctx if isinstance(source, _lib.Cloud) else
                  target.ctx if isinstance(target, _lib.Cloud) else _lib.default_context())
    est = estimation_method if estimation_method is not None else TransformationEstimationPointToPoint()
    crit = criteria if criteria is not None else ICPConvergenceCriteria()
    T0 = np.eye(4) if init is None else np.asarray(init, dtype=np.float64)
    if T0.shape != (4, 4):
        raise RuntimeError("init must be a 4x4 matrix")
    d_src, d_tgt = upload(source, ctx), upload(target, ctx)
    if est.code == _lib.POINT_TO_PLANE and not d_tgt.has_normals:
        raise RuntimeError("TransformationEstimationPointToPlane requires target normals")
    out = _lib.icp(ctx, d_src, d_tgt, max_correspondence_distance, T0, estimator=est.code,
                   max_iteration=crit.max_iteration, relative_fitness=crit.relative_fitness,
                   relative_rmse=crit.relative_rmse, want_corr=want_correspondences, allreduce=allreduce,
                   n_source_global=n_source_global)
    res = RegistrationResult(out["T"])
    res.fitness = out["fitness"]
    res.inlier_rmse = out["inlier_rmse"]
    res.iterations = out["iters"]
    if want_correspondences:
        src_idx = np.nonzero(out["corr"] >= 0)[0].astype(np.int32)
        res.correspondence_set = np.stack([src_idx, out["corr"][src_idx]], axis=1)
    return res


def registration_icp_batch(source, target, radii, inits, estimation_method=None, criteria=None, ctx=None):
    """B independent registrations of one source against one target: `radii[b]` and `inits[b]` per
    registration, one estimation method and one set of criteria.  On device handles (`upload`) they
    run concurrently through pedp_icp_batched_ex; on anything else they are B calls of
    registration_icp.  Same results either way -- this is what lets improve_result try several
    randomised restarts at once (pedp_hip.icp_refine)."""
    est = estimation_method if estimation_method is not None else TransformationEstimationPointToPoint()
    crit = criteria if criteria is not None else ICPConvergenceCriteria()
    if not (isinstance(source, _lib.Cloud) and isinstance(target, _lib.Cloud)):
        return [registration_icp(source, target, r, T0, est, crit) for r, T0 in zip(radii, inits)]
    if est.code == _lib.POINT_TO_PLANE and not target.has_normals:
        raise RuntimeError("TransformationEstimationPointToPlane requires target normals")
    T, fit, rmse, its = _lib.icp_batched_ex(ctx or source.ctx, source, target, radii, np.asarray(inits, dtype=np.float64),
                                            estimator=est.code, max_iteration=crit.max_iteration,
                                            relative_fitness=crit.relative_fitness, relative_rmse=crit.relative_rmse)
    out = []
    for b in range(len(T)):
        res = RegistrationResult(T[b])
        res.fitness, res.inlier_rmse, res.iterations = float(fit[b]), float(rmse[b]), int(its[b])
        out.append(res)
    return out
