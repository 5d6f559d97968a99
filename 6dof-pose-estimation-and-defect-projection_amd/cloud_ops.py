"""Point-cloud operations behind the PointCloud holder's Open3D-style methods:
voxel_down_sample, segment_plane, cluster_dbscan, remove_statistical_outlier as used by
preprocess_source (src/pose_estimation.py:186-268).  The per-point / per-iteration work runs in
libpedp_hip.so; what is left on the host is O(N) bookkeeping in the oracle's order."""
import ctypes as C

import numpy as np

from . import _lib

_SEED = 0


def set_ransac_seed(seed):
    """Seed of segment_plane's sampler (Open3D: o3d.utility.random.seed).  The global numpy RNG is
    never touched, so improve_result's random stream stays the reference's."""
    global _SEED
    _SEED = int(seed)


def _pts(points):
    return np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)


def _is_device_tensor(x):
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)


def voxel_down_sample(points, voxel_size, normals=None, ctx=None):
    """Voxel-grid averages (PointCloud.voxel_down_sample).  `points` may be a float64 N x 3 torch tensor on the
    GPU (a scene that was back-projected there): the grid is then built from the device array and only the
    averages come to the host."""
    ctx = ctx or _lib.default_context()
    if _is_device_tensor(points):
        import torch

        if normals is not None and len(normals):
            raise NotImplementedError("voxel_down_sample: device points with normals")
        t = points.reshape(-1, 3)
        if t.device.index != ctx.device:   # a pointer of another GPU would be a fault inside the kernels, not an error
            raise _lib.PedpError(f"voxel_down_sample: the points are on cuda:{t.device.index}, the context on cuda:{ctx.device}")
        if t.dtype != torch.float64 or not t.is_contiguous():
            t = t.to(torch.float64).contiguous()              # (a float32 or strided tensor is copied on the device)
        torch.cuda.current_stream(t.device).synchronize()     # the library reads the array on its own stream
        out = np.empty((len(t), 3), np.float64)
        m = C.c_int64()
        _lib.check(_lib.load().pedp_voxel_down_sample_device_in(ctx._h, C.c_void_p(t.data_ptr()), len(t), float(voxel_size),
                                                                _lib._ptr(out), len(t), C.byref(m)),
                   "pedp_voxel_down_sample_device_in")
        return out[:m.value].copy(), None          # (a copy: the N x 3 output buffer is not kept alive by a small result)
    p = _pts(points)
    n = None if normals is None or len(normals) == 0 else _pts(normals)
    out, outn = np.empty_like(p), (np.empty_like(p) if n is not None else None)
    m = C.c_int64()
    _lib.check(_lib.load().pedp_voxel_down_sample(ctx._h, _lib._ptr(p), _lib._ptr(n), len(p), float(voxel_size), _lib._ptr(out),
                                                  _lib._ptr(outn), len(p), C.byref(m)), "pedp_voxel_down_sample")
    # copies: a down-sampled cloud must not keep the full-resolution output buffers alive
    return out[:m.value].copy(), (None if outn is None else outn[:m.value].copy())


def preprocess_source_fused(points, voxel_size, plane_distance, plane_iterations, first_frame, seed=None, normal_radius=2.0,
                            normal_max_nn=5, cluster_eps=10.0, cluster_min_points=10, outlier_neighbors=75, outlier_std_ratio=0.01,
                            ctx=None, box=False, report=False):
    """pedp_preprocess_source[_ex]: the reference's preprocess_source chain (pose_estimation.py:186-268; every branch but
    param['mesh']) with the scene staying on the device between the stages.  `points`: N x 3 float64 numpy array, or a
    float64 torch tensor on the GPU.  box: param['box'] (half-space cut instead of the plane removal).  report: also return
    what the reference's INFO lines print.  Returns (points, normals or None, stage_counts, status[, report]); status != 0:
    nothing was produced (no cluster / fewer than three points / an undecidable plane flip) and the caller takes the
    step-by-step path to reproduce the reference's behaviour.  report = {"plane_model": refit plane (4,), "mean_normal":
    (3,) un-normalised mean of the 10-unit voxel normals (1, 1, 1 on tracking frames), "flipped": bool (box only)}."""
    ctx = ctx or _lib.default_context()
    prm = _lib.PreprocessParams(float(voxel_size), float(plane_distance), int(plane_iterations), 1 if first_frame else 0,
                                int(_SEED if seed is None else seed), float(normal_radius), int(normal_max_nn), int(cluster_min_points),
                                float(cluster_eps), int(outlier_neighbors), 1 if box else 0, float(outlier_std_ratio), 0.0)
    if _is_device_tensor(points):
        import torch

        t = points.reshape(-1, 3)
        if t.device.index != ctx.device:
            raise _lib.PedpError(f"preprocess_source: the points are on cuda:{t.device.index}, the context on cuda:{ctx.device}")
        if t.dtype != torch.float64 or not t.is_contiguous():
            t = t.to(torch.float64).contiguous()
        torch.cuda.current_stream(t.device).synchronize()     # the library reads the array on its own stream
        n, ptr, on_dev, keep = len(t), C.c_void_p(t.data_ptr()), 1, t
    else:
        p = _pts(points)
        n, ptr, on_dev, keep = len(p), _lib._ptr(p), 0, p
    rep = np.zeros(12, np.float64) if report else None
    # the result is a small fraction of the frame: room for a tenth (at least 64 k points), the whole cloud if that is short
    for cap in (max(n // 10, 65536), n):
        cap = min(cap, max(n, 1))
        out = _lib.host_array((cap, 3), np.float64)            # page-locked, pooled: the download needs no staging copy
        outn = _lib.host_array((cap, 3), np.float64) if first_frame else None
        m, status = C.c_int64(), C.c_int()
        counts = (C.c_int64 * 4)()
        rc = _lib.load().pedp_preprocess_source_ex(ctx._h, ptr, n, on_dev, C.byref(prm), _lib._ptr(out), _lib._ptr(outn), cap,
                                                   C.byref(m), counts, C.byref(status), _lib._ptr(rep))
        if rc != 0 and m.value > cap and cap < n:
            continue                                           # (more points survived than a tenth: once more with full room)
        _lib.check(rc, "pedp_preprocess_source")
        break
    del keep
    res = (out[:m.value].copy(), (None if outn is None else outn[:m.value].copy()), list(counts), status.value)
    if report:
        res += ({"plane_model": rep[:4].copy(), "mean_normal": rep[4:7].copy(), "flipped": bool(rep[7]), "inliers": int(rep[8])},)
    return res


def cluster_dbscan(points, eps, min_points, ctx=None):
    ctx = ctx or _lib.default_context()
    p = _pts(points)
    labels = np.empty(len(p), np.int32)
    _lib.check(_lib.load().pedp_cluster_dbscan(ctx._h, _lib._ptr(p), len(p), float(eps), int(min_points), _lib._ptr(labels)),
               "pedp_cluster_dbscan")
    return labels


def knn_mean_distance(points, k, ctx=None):
    ctx = ctx or _lib.default_context()
    p = _pts(points)
    avg = np.empty(len(p), np.float64)
    _lib.check(_lib.load().pedp_knn_mean_distance(ctx._h, _lib._ptr(p), len(p), int(k), _lib._ptr(avg)), "pedp_knn_mean_distance")
    return avg


def statistical_outlier_indices(avg, std_ratio):
    """Open3D's global step on the per-point mean distances: mean over the valid ones, Bessel-
    corrected deviation, keep 0 < avg < mean + ratio * std; sums in index order (cumsum)."""
    avg = np.asarray(avg, dtype=np.float64)
    valid = int(np.count_nonzero(avg >= 0))
    if valid == 0:
        return np.zeros(0, np.int64)
    pos = avg > 0
    mean = (np.cumsum(np.where(pos, avg, 0.0))[-1] if len(avg) else 0.0) / valid
    dev = np.where(pos, (avg - mean) * (avg - mean), 0.0)
    sq = np.cumsum(dev)[-1] if len(avg) else 0.0
    std = np.sqrt(sq / (valid - 1)) if valid > 1 else float("nan")
    return np.nonzero(pos & (avg < mean + std_ratio * std))[0]


def remove_statistical_outlier(points, nb_neighbors, std_ratio, ctx=None):
    """Indices kept by PointCloud.remove_statistical_outlier."""
    return statistical_outlier_indices(knn_mean_distance(points, nb_neighbors, ctx), std_ratio)


def segment_plane(points, distance_threshold, ransac_n=3, num_iterations=100, seed=None, ctx=None):
    if ransac_n != 3:
        raise NotImplementedError("segment_plane: ransac_n = 3 is the only form the reference uses")
    ctx = ctx or _lib.default_context()
    p = _pts(points)
    plane = np.zeros(4)
    inl = np.empty(len(p), np.int32)
    n = C.c_int64()
    _lib.check(_lib.load().pedp_segment_plane(ctx._h, _lib._ptr(p), len(p), float(distance_threshold), int(num_iterations),
                                              C.c_uint64(_SEED if seed is None else int(seed)), _lib._ptr(plane), _lib._ptr(inl),
                                              C.byref(n)), "pedp_segment_plane")
    return plane, inl[:n.value].copy()


def estimate_normals(points, radius, max_nn, prior=None, ctx=None):
    """Normals by PCA over the hybrid neighbourhood (radius, max_nn); `prior` = existing normals to
    agree with (Open3D keeps the orientation of normals that are already there)."""
    ctx = ctx or _lib.default_context()
    p = _pts(points)
    pr = None if prior is None or len(prior) == 0 else _pts(prior)
    out = np.empty_like(p)
    _lib.check(_lib.load().pedp_estimate_normals(ctx._h, _lib._ptr(p), len(p), float(radius), int(max_nn), _lib._ptr(pr),
                                                 _lib._ptr(out)), "pedp_estimate_normals")
    return out


def compute_fpfh(points, normals, radius, max_nn, ctx=None):
    """FPFH features of a cloud with normals, N x 33 float64 (row i = column i of Open3D's
    Feature.data); hybrid neighbourhood (radius, max_nn) as compute_fpfh_feature takes it
    (pose_estimation.py:132-137)."""
    ctx = ctx or _lib.default_context()
    p, n = _pts(points), _pts(normals)
    if len(p) != len(n):
        raise RuntimeError("compute_fpfh_feature needs a normal for every point")
    out = np.empty((len(p), 33), np.float64)
    _lib.check(_lib.load().pedp_fpfh(ctx._h, _lib._ptr(p), _lib._ptr(n), len(p), float(radius), int(max_nn), _lib._ptr(out)),
               "pedp_fpfh")
    return out


def match_features(source_features, target_features, ctx=None):
    """idx[i] = the target feature (row) nearest to source feature i, squared L2 over 33 float64
    components, ties to the lower index -- the correspondences registration_ransac_based_on_feature_matching
    starts from."""
    ctx = ctx or _lib.default_context()
    fs = np.ascontiguousarray(source_features, np.float64).reshape(-1, 33)
    ft = np.ascontiguousarray(target_features, np.float64).reshape(-1, 33)
    idx = np.empty(len(fs), np.int32)
    _lib.check(_lib.load().pedp_feature_match(ctx._h, _lib._ptr(fs), len(fs), _lib._ptr(ft), len(ft), _lib._ptr(idx)),
               "pedp_feature_match")
    return idx
