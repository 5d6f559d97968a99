"""`o3d.pipelines.registration` surface the reference touches, backed by libpedp_hip.so.

Reference call sites (src/pose_estimation.py):
    :519-521  registration_icp(source, target, distance_threshold, transformation,
                               TransformationEstimationPointToPlane())
    :654-660  ... same with ICPConvergenceCriteria(max_iteration=1)
    :584      o3d.geometry.get_rotation_matrix_from_xyz([a, b, c])
    :132-137  compute_fpfh_feature(cloud, KDTreeSearchParamHybrid(radius, max_nn))
    :482-501  registration_ransac_based_on_feature_matching(source, target, source_fpfh, target_fpfh, False,
                  distance_threshold, TransformationEstimationPointToPoint(False), 3,
                  [CorrespondenceCheckerBasedOnEdgeLength, ...BasedOnDistance, ...BasedOnNormal],
                  RANSACConvergenceCriteria(iterations, confidence))
Names, argument order and defaults are Open3D 0.18's; errors are RuntimeError like
Open3D's (a target without normals under point-to-plane).
"""
import numpy as np

from . import _lib
from .geometry import PointCloud, RegistrationResult, normals_of, points_of


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


def forget_device_copy(cloud):
    """Drop a holder's kept device copy (it is dropped by itself whenever the holder's points or normals change)."""
    if hasattr(cloud, "_device_copy"):
        cloud._device_copy = None


def upload(cloud, ctx=None):
    """Device copy of a PointCloud-like object; pass the returned handle to registration_icp
    when the same cloud is registered many times (improve_result does ~50 calls per frame).
    A PointCloud holder keeps its device copy: the model cloud of a camera loop -- the same holder in every frame's
    z search and restarts -- is uploaded, ordered and packed once, not twice per frame.  The kept copy is EXACT: a holder
    owns immutable arrays (geometry.PointCloud: in-place edits raise, setters and transforms replace the arrays and bump
    the holder's version), so the copy is valid exactly as long as the version it was made from is the holder's.
    Holders over borrowed memory (PointCloud.borrowed) and foreign objects are uploaded every time."""
    if isinstance(cloud, _lib.Cloud):
        return cloud
    ctx = ctx or _lib.default_context()
    pts, nrm = points_of(cloud), normals_of(cloud)
    if isinstance(cloud, PointCloud) and len(pts) and not cloud._borrowed:
        key = (id(ctx), cloud._version)
        kept = getattr(cloud, "_device_copy", None)
        if kept is not None and kept[0] == key:
            return kept[1]
        handle = _lib.Cloud(ctx, pts, nrm)
        cloud._device_copy = (key, handle)
        return handle
    return _lib.Cloud(ctx, pts, nrm)


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


# ---------------------------------------------------------------- feature-based global registration

class Feature:
    """o3d.pipelines.registration.Feature: `data` is dimension x N (33 x N for FPFH), like Open3D's."""

    def __init__(self, rows=None):
        self._rows = np.zeros((0, 33)) if rows is None else np.ascontiguousarray(rows, np.float64)

    @property
    def data(self):
        return self._rows.T

    def dimension(self):
        return self._rows.shape[1]

    def num(self):
        return self._rows.shape[0]


def compute_fpfh_feature(input, search_param, ctx=None):
    """FPFH features of a cloud with normals (pose_estimation.py:132-137, :175-180, :255-260)."""
    from . import cloud_ops

    if normals_of(input) is None or len(normals_of(input)) != len(points_of(input)):
        raise RuntimeError("compute_fpfh_feature: the cloud has no normals")
    return Feature(cloud_ops.compute_fpfh(points_of(input), normals_of(input), search_param.radius, search_param.max_nn, ctx=ctx))


class CorrespondenceCheckerBasedOnEdgeLength:
    def __init__(self, similarity_threshold=0.9):
        self.similarity_threshold = float(similarity_threshold)


class CorrespondenceCheckerBasedOnDistance:
    def __init__(self, distance_threshold):
        self.distance_threshold = float(distance_threshold)


class CorrespondenceCheckerBasedOnNormal:
    def __init__(self, normal_angle_threshold):
        self.normal_angle_threshold = float(normal_angle_threshold)


class RANSACConvergenceCriteria:
    def __init__(self, max_iteration=100000, confidence=0.999):
        self.max_iteration, self.confidence = int(max_iteration), float(confidence)


_ransac_seed = [None]
RANSAC_CHUNK = 16384      # draws generated per launch
RANSAC_BATCH = 32         # accepted draws validated at once (they share the launches of pedp_icp_batched)


def set_ransac_seed(seed):
    """Seed of the RANSAC draws (o3d.utility.random.seed): the first call after this uses `seed`, the next
    `seed + 1`, ... (every registration draws afresh, a seeded program repeats); None: taken from numpy's
    global RNG per call."""
    _ransac_seed[0] = None if seed is None else int(seed)


def registration_ransac_based_on_feature_matching(source, target, source_feature, target_feature, mutual_filter,
                                                  max_correspondence_distance, estimation_method=None, ransac_n=3,
                                                  checkers=(), criteria=None, ctx=None):
    """RANSAC over nearest-feature correspondences (Open3D 0.18 Registration.cpp, restated; pose_estimation.py:482-501).

    Every source point is paired with the target point whose feature is nearest.  An iteration draws
    ransac_n = 3 pairs, fits Umeyama without scaling and applies the checkers; a draw that passes is
    scored ON THE PAIRS, as 0.18's RegistrationRANSACBasedOnCorrespondence does
    (EvaluateRANSACBasedOnCorrespondence): fitness = share of the pairs closer than
    max_correspondence_distance under the draw, inlier rmse over those pairs; the best draw wins
    (fitness, then rmse) and tightens the iteration budget, k = log(1 - confidence) / log(1 - fitness^3).
    (Round 2 validated a draw with a nearest-neighbour pass over the whole source -- what Open3D did
    before 0.13)  Open3D walks the iterations under OpenMP with random_device-seeded
    engines; here they are taken in order (the single-thread semantics), the draws are a counter-based
    function of (seed, iteration), generated RANSAC_CHUNK at a time on the GPU, and the accepted ones
    are scored RANSAC_BATCH at a time (results used in order, those behind a tightened budget dropped)."""
    from . import cloud_ops

    est = estimation_method if estimation_method is not None else TransformationEstimationPointToPoint()
    if est.code != _lib.POINT_TO_POINT:
        raise NotImplementedError("registration_ransac_based_on_feature_matching: the reference fits point to point")
    if ransac_n != 3:
        raise NotImplementedError("registration_ransac_based_on_feature_matching: ransac_n = 3 (the reference's)")
    if mutual_filter:
        raise NotImplementedError("registration_ransac_based_on_feature_matching: mutual_filter=False (the reference's)")
    crit = criteria if criteria is not None else RANSACConvergenceCriteria()
    best = RegistrationResult(np.eye(4))
    best.fitness, best.inlier_rmse = 0.0, 0.0
    n_src = len(points_of(source))
    if max_correspondence_distance <= 0.0 or n_src < ransac_n or len(points_of(target)) == 0:
        return best
    ctx = ctx or _lib.default_context()
    edge, dist, angle = 0.0, float("inf"), np.pi         # a missing checker never rejects
    for ch in checkers:
        if isinstance(ch, CorrespondenceCheckerBasedOnEdgeLength):
            edge = ch.similarity_threshold
        elif isinstance(ch, CorrespondenceCheckerBasedOnDistance):
            dist = ch.distance_threshold
        elif isinstance(ch, CorrespondenceCheckerBasedOnNormal):
            angle = ch.normal_angle_threshold
        else:
            raise NotImplementedError(f"correspondence checker {type(ch).__name__} is not built")
    corr = cloud_ops.match_features(source_feature.data.T, target_feature.data.T, ctx=ctx)
    d_src, d_tgt = upload(source, ctx), upload(target, ctx)
    src_pts, tgt_pts = points_of(source), points_of(target)
    if _ransac_seed[0] is not None:
        seed = _ransac_seed[0]
        _ransac_seed[0] += 1
    else:
        seed = int(np.random.randint(0, 2 ** 31 - 1))
    budget, itr, validated = crit.max_iteration, 0, 0
    paired = tgt_pts[corr]
    r2 = max_correspondence_distance ** 2
    while itr < budget:
        count = min(RANSAC_CHUNK, budget - itr)
        ok, T = _lib.ransac_hypotheses(ctx, d_src, d_tgt, corr, seed, itr, count, edge, dist, angle)
        passed = np.flatnonzero(ok)
        for b0 in range(0, len(passed), RANSAC_BATCH):
            batch = passed[b0:b0 + RANSAC_BATCH]
            batch = batch[itr + batch < budget]
            if len(batch) == 0:
                break
            # EvaluateRANSACBasedOnCorrespondence for the whole batch: the PAIRS under each accepted draw
            moved = np.einsum("bij,nj->bni", T[batch][:, :3, :3], src_pts) + T[batch][:, None, :3, 3]
            d2 = ((moved - paired[None]) ** 2).sum(2)
            good = d2 < r2
            for row, k in enumerate(batch):
                if itr + k >= budget:
                    break
                validated += 1
                n_good = int(good[row].sum())
                fitness = n_good / len(corr)
                rmse = float(np.sqrt(d2[row][good[row]].sum() / n_good)) if n_good else 0.0
                if fitness > best.fitness or (fitness == best.fitness and rmse < best.inlier_rmse):   # IsBetterRANSACThan
                    best = RegistrationResult(T[k].copy())
                    best.fitness, best.inlier_rmse = fitness, rmse
                    best.correspondence_set = np.column_stack([np.flatnonzero(good[row]), corr[good[row]]]).astype(np.int32)
                    ratio = fitness                       # corres_inlier_ratio = |inlier pairs| / |pairs|
                    if 0.0 < ratio < 1.0 and crit.confidence < 1.0:
                        k_est = np.log(1.0 - crit.confidence) / np.log(1.0 - ratio ** ransac_n)
                        if k_est < budget:
                            budget = int(np.ceil(k_est))
                    elif ratio >= 1.0:
                        budget = min(budget, itr + int(k) + 1)
        itr += count
    best.validated_draws = validated
    return best
