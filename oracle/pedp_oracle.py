"""ORACLE -- TEST INFRASTRUCTURE ONLY (see pedp_oracle.h).

numpy/ctypes front end of the CPU restatement in this directory, plus a pure-Python
restatement of the reference's refinement control flow (improve_result,
predict_z_axis_adjustment) on top of the oracle's registration_icp.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY UNPINNED: the reference ships no tests or golden vectors for this path and its
arithmetic (open3d==0.18.0 / Embree / nanoflann) is not importable here; see the header.
"""
import copy
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libpedp_oracle.so")
P2PLANE, P2POINT = 0, 1
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("ray.c", "icp.c", "cluster.c", "pedp_oracle.h", "Makefile")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, stdout=subprocess.DEVNULL)
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.pedp_oracle_raycast.restype = C.c_int
        _lib.pedp_oracle_raycast_bvh.restype = C.c_int
        _lib.pedp_oracle_icp.restype = C.c_int
        _lib.pedp_oracle_solve6_ldlt.restype = C.c_int
        _lib.pedp_oracle_cluster_poses.restype = C.c_int
        _lib.pedp_oracle_mt_test.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


# ---------------------------------------------------------------- rays

def tri_setup(verts, tris):
    v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(tris, np.uint32).reshape(-1, 3)
    out = np.empty((len(t), 12), np.float32)   # v0, e1, e2, m = e2 x e1
    lib().pedp_oracle_tri_setup(_p(v), C.c_int64(len(v)), _p(t), C.c_int64(len(t)), _p(out))
    return out


def raycast(verts, tris, rays6, nthreads=0, bvh=False, timings=None):
    """Closest hit per ray: dict(t_hit f32, primitive_ids u32, primitive_uvs f32 Nx2)."""
    tri9 = tri_setup(verts, tris)
    r = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6)
    n = len(r)
    t = np.empty(n, np.float32)
    ids = np.empty(n, np.uint32)
    uv = np.empty((n, 2), np.float32)
    if bvh:
        tb, tc = C.c_double(0), C.c_double(0)
        rc = lib().pedp_oracle_raycast_bvh(_p(tri9), C.c_int64(len(tri9)), _p(r), C.c_int64(n), _p(t), _p(ids),
                                           _p(uv), C.c_int(nthreads), C.byref(tb), C.byref(tc))
        if timings is not None:
            timings["build_s"], timings["cast_s"] = tb.value, tc.value
    else:
        rc = lib().pedp_oracle_raycast(_p(tri9), C.c_int64(len(tri9)), _p(r), C.c_int64(n), _p(t), _p(ids),
                                       _p(uv), C.c_int(nthreads))
    if rc != 0:
        raise RuntimeError(f"oracle raycast failed ({rc})")
    return {"t_hit": t, "primitive_ids": ids, "primitive_uvs": uv}


def accepted_pairs(verts, tris, rays6):
    """Every (ray, triangle) pair the oracle's test accepts: int32 [n, 2] in (ray, triangle) order."""
    tri9 = tri_setup(verts, tris)
    r = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6)
    fn = lib().pedp_oracle_accepted_pairs
    fn.restype = C.c_int64
    cap = max(4 * len(r), 1024)
    while True:
        out = np.empty((cap, 2), np.int32)
        n = fn(_p(tri9), C.c_int64(len(tri9)), _p(r), C.c_int64(len(r)), _p(out), C.c_int64(cap))
        if n <= cap:
            return out[:n]
        cap = int(n)


def mt_test(o, d, tri9):
    o = np.ascontiguousarray(o, np.float32)
    d = np.ascontiguousarray(d, np.float32)
    tr = np.ascontiguousarray(tri9, np.float32)
    t, u, v = C.c_float(0), C.c_float(0), C.c_float(0)
    hit = lib().pedp_oracle_mt_test(_p(o), _p(d), _p(tr), C.byref(t), C.byref(u), C.byref(v))
    return (bool(hit), t.value, u.value, v.value)


# ---------------------------------------------------------------- ICP

def nn(src, tgt, kdtree=False, nthreads=0):
    s = np.ascontiguousarray(src, np.float64).reshape(-1, 3)
    t = np.ascontiguousarray(tgt, np.float64).reshape(-1, 3)
    idx = np.empty(len(s), np.int32)
    d2 = np.empty(len(s), np.float64)
    fn = lib().pedp_oracle_nn_kdtree if kdtree else lib().pedp_oracle_nn
    fn(_p(s), C.c_int64(len(s)), _p(t), C.c_int64(len(t)), _p(idx), _p(d2), C.c_int(nthreads))
    return idx, d2


def transform(T, pts):
    p = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
    M = np.ascontiguousarray(T, np.float64).reshape(16)
    out = np.empty_like(p)
    lib().pedp_oracle_transform(_p(M), _p(p), C.c_int64(len(p)), _p(out))
    return out


def pose_vertices(T, verts, dtype=np.float32):
    """TriangleMesh.transform (pose_estimation.py:406-409) + from_legacy's float32 cast
    (defect_projection.py:245): Open3D forms T * (x, y, z, 1) in float64 and divides by the
    fourth component.  Fixed order ((T0 x + T1 y) + T2 z) + T3, elementwise numpy (no FMA).
    dtype=np.float64: the vertices before the cast (what the transformed mesh itself holds)."""
    v = np.ascontiguousarray(verts, np.float64).reshape(-1, 3)
    M = np.asarray(T, np.float64).reshape(4, 4)
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    h = [((M[r, 0] * x + M[r, 1] * y) + M[r, 2] * z) + M[r, 3] for r in range(4)]
    return np.stack([h[0] / h[3], h[1] / h[3], h[2] / h[3]], axis=1).astype(dtype)


def project_heatmap(verts32, tris, heatmap, K, threshold=0.5, origin=(0, 0, 0), bvh=True):
    """heatmap_to_points + compute_rays + intersect_rays_with_mesh
    (defect_projection.py:165-179, :196-223, :225-266) on whole arrays: np.where order, float64
    directions with the norm formed as sqrt((xn*xn + yn*yn) + 1), [o | d] cast to float32 for the
    sweep, t != inf filter, o + d * t in float64."""
    h = np.asarray(heatmap, np.float64)
    K = np.asarray(K, np.float64)
    ys, xs = np.nonzero(h > threshold)
    xn = (xs.astype(np.float64) - K[0, 2]) / K[0, 0]
    yn = (ys.astype(np.float64) - K[1, 2]) / K[1, 1]
    ln = np.sqrt((xn * xn + yn * yn) + 1.0)
    d = np.stack([xn / ln, yn / ln, 1.0 / ln], axis=1)
    o = np.asarray(origin, np.float64).reshape(3)
    rays6 = np.hstack([np.tile(o, (len(d), 1)), d]).astype(np.float32)
    hit = raycast(verts32, tris, rays6, bvh=bvh) if len(d) else {"t_hit": np.zeros(0, np.float32),
                                                                  "primitive_ids": np.zeros(0, np.uint32)}
    valid = hit["t_hit"] != np.inf
    pts = o + d[valid] * hit["t_hit"][valid, None].astype(np.float64)
    return {"points": pts, "intensities": h[ys[valid], xs[valid]],
            "pixels": np.stack([xs[valid], ys[valid]], axis=1).astype(np.int32),
            "primitive_ids": hit["primitive_ids"][valid], "n_rays": len(d), "rays6": rays6}


def erode_depth(depth, radius=2, depth_diff_thres=0.001, ratio_thres=0.8, zfar=100, nthreads=0):
    d = np.ascontiguousarray(depth, np.float32)
    out = np.zeros_like(d)
    lib().pedp_oracle_erode_depth(_p(d), C.c_int(d.shape[0]), C.c_int(d.shape[1]), C.c_int(radius),
                                  C.c_float(depth_diff_thres), C.c_float(ratio_thres), C.c_float(zfar), _p(out),
                                  C.c_int(nthreads or os.cpu_count()))
    return out


def bilateral_filter_depth(depth, radius=2, zfar=100, sigmaD=2, sigmaR=100000, nthreads=0):
    d = np.ascontiguousarray(depth, np.float32)
    out = np.zeros_like(d)
    lib().pedp_oracle_bilateral_depth(_p(d), C.c_int(d.shape[0]), C.c_int(d.shape[1]), C.c_int(radius), C.c_float(zfar),
                                      C.c_float(sigmaD), C.c_float(sigmaR), _p(out), C.c_int(nthreads or os.cpu_count()))
    return out


def depth2xyzmap(depth, K):
    d = np.ascontiguousarray(depth, np.float32)
    Kd = np.ascontiguousarray(K, np.float64).reshape(9)
    out = np.zeros(d.shape + (3,), np.float32)
    lib().pedp_oracle_depth2xyzmap(_p(d), C.c_int(d.shape[0]), C.c_int(d.shape[1]), _p(Kd), _p(out))
    return out


def depth2xyzmap_batch(depths, Ks, zfar):
    d = np.ascontiguousarray(depths, np.float32)
    Kf = np.ascontiguousarray(Ks, np.float32).reshape(len(d), 9)
    out = np.zeros(d.shape + (3,), np.float32)
    lib().pedp_oracle_depth2xyzmap_batch(_p(d), C.c_int(d.shape[0]), C.c_int(d.shape[1]), C.c_int(d.shape[2]), _p(Kf),
                                         C.c_float(zfar), _p(out))
    return out


def voxel_down_sample(points, voxel_size, normals=None):
    """PointCloud.voxel_down_sample: (points, normals or None), voxels in ascending (ix, iy, iz)."""
    p = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    n = None if normals is None else np.ascontiguousarray(normals, np.float64).reshape(-1, 3)
    out = np.empty_like(p)
    outn = np.empty_like(p) if n is not None else None
    fn = lib().pedp_oracle_voxel_down_sample
    fn.restype = C.c_int64
    m = fn(_p(p), _p(n), C.c_int64(len(p)), C.c_double(voxel_size), _p(out), _p(outn))
    return out[:m].copy(), (None if outn is None else outn[:m].copy())


def cluster_dbscan(points, eps, min_points):
    p = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    labels = np.empty(len(p), np.int32)
    lib().pedp_oracle_dbscan(_p(p), C.c_int64(len(p)), C.c_double(eps), C.c_int(min_points), _p(labels))
    return labels


def knn_mean_distance(points, k):
    p = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    avg = np.empty(len(p), np.float64)
    lib().pedp_oracle_knn_mean_distance(_p(p), C.c_int64(len(p)), C.c_int(k), _p(avg))
    return avg


def statistical_outlier_indices(avg, std_ratio):
    """The global part of PointCloud.remove_statistical_outlier on the per-point mean distances:
    mean over the valid ones, Bessel-corrected deviation, keep 0 < avg < mean + ratio * std; all
    sums in index order."""
    valid = 0
    total = 0.0
    for a in avg:
        if a > 0:
            total += a
        if a >= 0:
            valid += 1
    if valid == 0:
        return np.zeros(0, np.int64)
    mean = total / valid
    sq = 0.0
    for a in avg:
        if a > 0:
            sq += (a - mean) * (a - mean)
    std = np.sqrt(sq / (valid - 1)) if valid > 1 else float("nan")
    thr = mean + std_ratio * std
    return np.nonzero((avg > 0) & (avg < thr))[0]


def remove_statistical_outlier(points, nb_neighbors, std_ratio):
    return statistical_outlier_indices(knn_mean_distance(points, nb_neighbors), std_ratio)


def estimate_normals(points, radius, max_nn, prior=None):
    p = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    pr = None if prior is None else np.ascontiguousarray(prior, np.float64).reshape(-1, 3)
    out = np.empty_like(p)
    lib().pedp_oracle_estimate_normals(_p(p), C.c_int64(len(p)), C.c_double(radius), C.c_int(max_nn), _p(pr), _p(out))
    return out


def fpfh(points, normals, radius, max_nn):
    """N x 33 FPFH features (the transpose of Open3D's Feature.data)."""
    p = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    n = np.ascontiguousarray(normals, np.float64).reshape(-1, 3)
    out = np.empty((len(p), 33), np.float64)
    lib().pedp_oracle_fpfh(_p(p), _p(n), C.c_int64(len(p)), C.c_double(radius), C.c_int(max_nn), _p(out))
    return out


def feature_match(fs, ft):
    fs = np.ascontiguousarray(fs, np.float64).reshape(-1, 33)
    ft = np.ascontiguousarray(ft, np.float64).reshape(-1, 33)
    idx = np.empty(len(fs), np.int32)
    lib().pedp_oracle_feature_match(_p(fs), C.c_int64(len(fs)), _p(ft), C.c_int64(len(ft)), _p(idx))
    return idx


def ransac_hypothesis(seed, itr, src, src_nrm, tgt, tgt_nrm, corr, edge, dist, angle):
    """(accepted, T) of RANSAC draw `itr`."""
    s = np.ascontiguousarray(src, np.float64).reshape(-1, 3)
    t = np.ascontiguousarray(tgt, np.float64).reshape(-1, 3)
    sn = None if src_nrm is None else np.ascontiguousarray(src_nrm, np.float64).reshape(-1, 3)
    tn = None if tgt_nrm is None else np.ascontiguousarray(tgt_nrm, np.float64).reshape(-1, 3)
    c = np.ascontiguousarray(corr, np.int32)
    T = np.empty(16, np.float64)
    fn = lib().pedp_oracle_ransac_hypothesis
    fn.restype = C.c_int
    ok = fn(C.c_uint64(seed), C.c_int64(itr), _p(s), _p(sn), C.c_int64(len(s)), _p(t), _p(tn), _p(c), C.c_double(edge),
            C.c_double(dist), C.c_double(angle), _p(T))
    return bool(ok), T.reshape(4, 4)


def corres_inlier_ratio(src, tgt, corr, T, max_dist):
    s = np.ascontiguousarray(src, np.float64).reshape(-1, 3)
    t = np.ascontiguousarray(tgt, np.float64).reshape(-1, 3)
    c = np.ascontiguousarray(corr, np.int32)
    Tm = np.ascontiguousarray(T, np.float64).reshape(16)
    fn = lib().pedp_oracle_corres_inlier_ratio
    fn.restype = C.c_double
    return fn(_p(s), C.c_int64(len(s)), _p(t), _p(c), _p(Tm), C.c_double(max_dist))


def smallest_eigenvector(cov):
    c = np.ascontiguousarray(cov, np.float64).reshape(9)
    out = np.zeros(3)
    lib().pedp_oracle_smallest_eigenvector(_p(c), _p(out))
    return out


def segment_plane(points, distance_threshold, num_iterations, seed=0):
    p = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    plane = np.zeros(4)
    inl = np.empty(len(p), np.int32)
    fn = lib().pedp_oracle_segment_plane
    fn.restype = C.c_int64
    n = fn(_p(p), C.c_int64(len(p)), C.c_double(distance_threshold), C.c_int(num_iterations), C.c_uint64(seed),
           _p(plane), _p(inl))
    return plane, inl[:n].copy()


def sample3(seed, t, n):
    out = (C.c_int64 * 3)()
    lib().pedp_oracle_sample3(C.c_uint64(seed), C.c_int64(t), C.c_int64(n), out)
    return list(out)


def icp(src, tgt, tgt_normals, max_corr_dist, init, estimator=P2PLANE, max_iter=30, rel_fitness=1e-6,
        rel_rmse=1e-6, kdtree=True, nthreads=0, want_trace=True):
    s = np.ascontiguousarray(src, np.float64).reshape(-1, 3)
    t = np.ascontiguousarray(tgt, np.float64).reshape(-1, 3)
    nrm = None if tgt_normals is None else np.ascontiguousarray(tgt_normals, np.float64).reshape(-1, 3)
    T0 = np.ascontiguousarray(init, np.float64).reshape(16)
    T = np.empty(16, np.float64)
    fit, rmse, it = C.c_double(0), C.c_double(0), C.c_int32(0)
    corr = np.empty(len(s), np.int32)
    trace = np.zeros((max_iter + 1, 18), np.float64) if want_trace else None
    rc = lib().pedp_oracle_icp(_p(s), C.c_int64(len(s)), _p(t), _p(nrm), C.c_int64(len(t)),
                               C.c_double(max_corr_dist), _p(T0), C.c_int(estimator), C.c_int(max_iter),
                               C.c_double(rel_fitness), C.c_double(rel_rmse), _p(T), C.byref(fit), C.byref(rmse),
                               C.byref(it), _p(corr), _p(trace), C.c_int(1 if kdtree else 0), C.c_int(nthreads))
    if rc == -2:
        raise RuntimeError("TransformationEstimationPointToPlane requires target normals")
    if rc != 0:
        raise RuntimeError(f"oracle icp failed ({rc})")
    out = {"T": T.reshape(4, 4), "fitness": fit.value, "inlier_rmse": rmse.value, "iters": it.value, "corr": corr}
    if want_trace:
        out["trace"] = trace[: it.value + 1]
    return out


def solve6(A, b):
    A = np.ascontiguousarray(A, np.float64).reshape(36)
    b = np.ascontiguousarray(b, np.float64).reshape(6)
    x = np.empty(6, np.float64)
    ok = lib().pedp_oracle_solve6_ldlt(_p(A), _p(b), _p(x))
    return bool(ok), x


def vec6_to_T(x):
    x = np.ascontiguousarray(x, np.float64).reshape(6)
    T = np.empty(16, np.float64)
    lib().pedp_oracle_vec6_to_T(_p(x), _p(T))
    return T.reshape(4, 4)


def kabsch(S, Tg):
    S = np.ascontiguousarray(S, np.float64).reshape(-1, 3)
    Tg = np.ascontiguousarray(Tg, np.float64).reshape(-1, 3)
    T = np.empty(16, np.float64)
    lib().pedp_oracle_kabsch(_p(S), _p(Tg), C.c_int64(len(S)), _p(T))
    return T.reshape(4, 4)


def rot_xyz(abc):
    a = np.ascontiguousarray(abc, np.float64).reshape(3)
    R = np.empty(9, np.float64)
    lib().pedp_oracle_rot_xyz(_p(a), _p(R))
    return R.reshape(3, 3)


def cluster_poses(angle_diff, dist_diff, poses, syms):
    p = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    s = np.ascontiguousarray(syms, np.float32).reshape(-1, 16)
    keep = np.empty(max(len(p), 1), np.int32)
    nk = C.c_int(0)
    rc = lib().pedp_oracle_cluster_poses(C.c_float(angle_diff), C.c_float(dist_diff), _p(p), C.c_int(len(p)),
                                         _p(s), C.c_int(len(s)), _p(keep), C.byref(nk))
    if rc != 0:
        raise RuntimeError("oracle cluster_poses failed")
    return keep[: nk.value].copy()


# ---------------------------------------------------------------- reference control flow
# Restated from the reference's Python (not imported: its first statement is
# `import open3d`, absent here).  Clouds are (points, normals) array pairs.

class Result:
    def __init__(self, transformation=None, fitness=0.0, inlier_rmse=0.0):
        self.transformation = np.eye(4) if transformation is None else np.array(transformation, dtype=np.float64)
        self.fitness = fitness
        self.inlier_rmse = inlier_rmse


def refine_registration(src, tgt, tgt_normals, transformation, param, **kw):
    """src/pose_estimation.py:505-522."""
    r = icp(src, tgt, tgt_normals, param["refine_registration"]["distance_threshold"], transformation, **kw)
    return Result(r["T"], r["fitness"], r["inlier_rmse"])


def predict_z_axis_adjustment(src, tgt, tgt_normals, initial_fp_transformation, param, max_adjustment=50,
                              initial_step=10):
    """src/pose_estimation.py:624-683 -- adaptive 1-D search on the camera-z offset, every
    probe a single-iteration ICP started from inv(T with T[2,3] -= adjustment)."""
    best_adj, best_fit, best_rmse = 0, 0, float("inf")
    cur, step, direction = 0, initial_step, 1
    while abs(step) >= 0.1:
        T = np.copy(initial_fp_transformation)
        T[2, 3] -= cur
        r = icp(src, tgt, tgt_normals, param["refine_registration"]["distance_threshold"], np.linalg.inv(T),
                max_iter=1)
        if r["fitness"] > best_fit or (r["fitness"] == best_fit and r["inlier_rmse"] < best_rmse):
            best_adj, best_fit, best_rmse = cur, r["fitness"], r["inlier_rmse"]
            cur += step * direction
        else:
            direction *= -1
            step /= 2
            cur += step * direction
        if abs(cur) > max_adjustment:
            cur = max_adjustment * np.sign(cur)
            step /= 1.25
            direction *= -1
        if best_fit > 0.95:
            break
    return best_adj, best_fit, best_rmse


def improve_result(src, tgt, tgt_normals, current_result, parameter, trace=None):
    """src/pose_estimation.py:547-622 -- up to 50 randomised restarts around the best
    transformation so far; global numpy RNG, consumption order of SURVEY Appendix C; the
    distance threshold compounds because `parameters.copy()` is shallow (:580-582)."""
    parameters = copy.deepcopy(parameter)
    if not hasattr(current_result, "fitness") or current_result.fitness is None:
        current_result = Result(current_result, 0.8, 3.0)
    best_fit, best_rmse = current_result.fitness, current_result.inlier_rmse
    best_T = np.linalg.inv(current_result.transformation)
    it, x = 0, 0.1
    while it < 50 and (best_fit < parameters["run_icp"]["fitness_threshold"]
                       or best_rmse > parameters["run_icp"]["rmse_threshold"]):
        cur = parameters.copy()
        cur["refine_registration"]["distance_threshold"] *= np.random.uniform(0.8, 1.2)
        R = rot_xyz([np.random.uniform(-0.01, 0.01) for _ in range(3)])
        tr = np.random.uniform(-x, x, 3)
        N = np.eye(4)
        N[:3, :3] = R
        N[:3, 3] = tr
        res = refine_registration(src, tgt, tgt_normals, N @ best_T, cur)
        if trace is not None:
            trace.append((cur["refine_registration"]["distance_threshold"], res.fitness, res.inlier_rmse,
                          res.transformation.copy()))
        if res.fitness > 0 and res.inlier_rmse > 0:
            if res.fitness > best_fit or (res.fitness == best_fit and res.inlier_rmse < best_rmse):
                best_fit, best_rmse, best_T = res.fitness, res.inlier_rmse, res.transformation
        else:
            x += 0.25
        it += 1
    return Result(best_T, best_fit, best_rmse)
