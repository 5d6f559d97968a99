"""CPU tests of the oracle: known answers, independent maths (numpy / scipy), and the
committed golden fixtures.  The oracle is "parity unpinned" against the reference (no fixture
exists upstream); these tests pin it to analytic results and to itself over time."""
import os

import numpy as np
import pytest
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bits(a):
    return np.asarray(a).view(np.uint32)


# ---------------------------------------------------------------- rays

def test_g1_triangle_known_answers(oracle):
    g = np.load(os.path.join(GOLD, "g1_triangle.npz"))
    r = oracle.raycast(g["verts"], g["tris"], g["rays"])
    assert r["t_hit"].tolist() == g["t_analytic"].tolist()
    assert np.array_equal(r["primitive_ids"], g["ids"])
    assert np.array_equal(_bits(r["primitive_uvs"]), _bits(g["uv"]))
    assert r["primitive_uvs"][0].tolist() == [0.25, 0.25]


def test_mt_single_test_edges_and_degenerate(oracle):
    tri = oracle.tri_setup(np.array([[0, 0, 1], [2, 0, 1], [0, 2, 1]], np.float32), np.array([[0, 1, 2]]))[0]
    assert oracle.mt_test([0.5, 0.5, 0], [0, 0, 1], tri) == (True, 1.0, 0.25, 0.25)
    assert oracle.mt_test([1.0, 1.0, 0], [0, 0, 1], tri)[0]          # on the hypotenuse
    assert oracle.mt_test([0.0, 0.0, 0], [0, 0, 1], tri)[0]          # on vertex v0
    assert not oracle.mt_test([1.01, 1.0, 0], [0, 0, 1], tri)[0]
    assert not oracle.mt_test([0.5, 0.5, 0], [1, 0, 0], tri)[0]      # parallel: det == 0
    assert not oracle.mt_test([0.5, 0.5, 2], [0, 0, 1], tri)[0]      # behind
    assert oracle.mt_test([0.5, 0.5, 1], [0, 0, 1], tri)[:2] == (True, 0.0)   # tnear = 0 inclusive
    flat = np.zeros(9, np.float32)                                   # zero-area triangle never hits
    assert not oracle.mt_test([0, 0, 0], [0, 0, 1], flat)[0]


def test_g2_torus_golden_and_bvh_equals_brute(oracle):
    g = np.load(os.path.join(GOLD, "g2_torus_rays.npz"))
    for bvh in (False, True):
        r = oracle.raycast(g["verts_posed"], g["tris"], g["rays6"], bvh=bvh)
        assert np.array_equal(r["primitive_ids"], g["ids"])
        assert np.array_equal(_bits(r["t_hit"]), _bits(g["t_hit"]))
        assert np.array_equal(_bits(r["primitive_uvs"]), _bits(g["uv"]))
    hit = np.isfinite(g["t_hit"])
    assert 300 < hit.sum() < len(hit)
    # hit points lie on their triangle: P = v0 + u e1 + v e2 (float64 check of fp32 results)
    tri9 = oracle.tri_setup(g["verts_posed"], g["tris"]).astype(np.float64)[g["ids"][hit]]
    P = g["rays6"][hit, :3].astype(np.float64) + g["rays6"][hit, 3:].astype(np.float64) * g["t_hit"][hit, None]
    u, v = g["uv"][hit, 0:1].astype(np.float64), g["uv"][hit, 1:2].astype(np.float64)
    Q = tri9[:, 0:3] + u * tri9[:, 3:6] + v * tri9[:, 6:9]
    assert np.abs(P - Q).max() < 2e-3   # fp32 rounding at ~400 mm range


def test_bvh_equals_brute_on_random_soup(oracle):
    rng = np.random.default_rng(1)
    v = rng.normal(0, 30, (900, 3)).astype(np.float32)
    t = rng.integers(0, 900, (2500, 3)).astype(np.uint32)
    o = rng.normal(0, 80, (4000, 3))
    d = rng.normal(0, 1, (4000, 3))
    rays = np.hstack([o, d]).astype(np.float32)
    a = oracle.raycast(v, t, rays)
    b = oracle.raycast(v, t, rays, bvh=True)
    assert np.array_equal(a["primitive_ids"], b["primitive_ids"]) and np.array_equal(_bits(a["t_hit"]), _bits(b["t_hit"]))


def test_tie_break_lowest_index(oracle):
    v = np.array([[0, 0, 5], [1, 0, 5], [0, 1, 5]] * 2, np.float32)
    t = np.array([[3, 4, 5], [0, 1, 2]], np.uint32)
    r = oracle.raycast(v, t, np.array([[0.2, 0.2, 0, 0, 0, 1]], np.float32))
    assert r["primitive_ids"][0] == 0 and r["t_hit"][0] == 5.0


# ---------------------------------------------------------------- ICP pieces

def test_nn_brute_kdtree_scipy_agree(oracle):
    rng = np.random.default_rng(2)
    tgt = rng.normal(0, 40, (3000, 3))
    src = np.vstack([rng.normal(0, 45, (2000, 3)), rng.normal(0, 40, (200, 3)) + [500, 0, 0]])
    i1, d1 = oracle.nn(src, tgt)
    i2, d2 = oracle.nn(src, tgt, kdtree=True)
    dd, ii = cKDTree(tgt).query(src)
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
    assert np.array_equal(i1, ii)
    assert np.allclose(np.sqrt(d1), dd, rtol=1e-13)


def test_nn_exact_ties_pick_lowest_index(oracle):
    tgt = np.array([[1.0, 0, 0], [-1.0, 0, 0], [0, 1.0, 0], [1.0, 0, 0]])
    src = np.zeros((1, 3))
    for kd in (False, True):
        i, d = oracle.nn(src, tgt, kdtree=kd)
        assert i[0] == 0 and d[0] == 1.0


def test_solve6_against_numpy(oracle):
    rng = np.random.default_rng(3)
    for _ in range(20):
        J = rng.normal(size=(40, 6))
        A = J.T @ J
        b = rng.normal(size=6)
        ok, x = oracle.solve6(A, b)
        assert ok and np.allclose(x, np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)
    ok, x = oracle.solve6(np.zeros((6, 6)), np.ones(6))   # Eigen's LDLT of a zero matrix solves to zero
    assert ok and np.all(x == 0)


def test_vec6_and_rot_xyz_conventions(oracle):
    x = np.array([0.3, -0.2, 0.5, 1.0, 2.0, 3.0])
    T = oracle.vec6_to_T(x)
    R = (Rotation.from_euler("z", x[2]) * Rotation.from_euler("y", x[1]) * Rotation.from_euler("x", x[0])).as_matrix()
    assert np.allclose(T[:3, :3], R, atol=1e-15) and np.array_equal(T[:3, 3], x[3:]) and np.array_equal(T[3], [0, 0, 0, 1])
    abc = [0.1, -0.4, 0.7]
    Rx = (Rotation.from_euler("x", abc[0]) * Rotation.from_euler("y", abc[1]) * Rotation.from_euler("z", abc[2])).as_matrix()
    assert np.allclose(oracle.rot_xyz(abc), Rx, atol=1e-15)


def test_kabsch_against_numpy_svd_including_planar_and_reflection(oracle):
    rng = np.random.default_rng(4)
    R = Rotation.from_rotvec([0.4, -0.3, 0.8]).as_matrix()
    t = np.array([3.0, -2.0, 1.0])
    for S in (rng.normal(0, 10, (50, 3)),                       # generic
              np.c_[rng.normal(0, 10, (50, 2)), np.zeros(50)],  # planar: rank-2 covariance
              np.c_[rng.normal(0, 10, 50), np.zeros((50, 2))]):  # collinear: rank 1 (rotation not unique)
        Tg = S @ R.T + t
        T = oracle.kabsch(S, Tg)
        assert abs(np.linalg.det(T[:3, :3]) - 1) < 1e-9
        assert np.abs(S @ T[:3, :3].T + T[:3, 3] - Tg).max() < 1e-8
    S = rng.normal(0, 10, (30, 3))
    Tm = S * [1, 1, -1]                                         # mirrored target: proper rotation must come back
    T = oracle.kabsch(S, Tm)
    assert abs(np.linalg.det(T[:3, :3]) - 1) < 1e-9


def test_transform_op_order(oracle):
    T = np.array([[0.1, 0.7, -0.3, 5.0], [0.2, -0.5, 0.9, -1.0], [0.6, 0.4, 0.8, 2.0], [0, 0, 0, 1.0]])
    p = np.array([[1e3, -2e3, 3.5e2]])
    out = oracle.transform(T, p)[0]
    exp = [((T[r, 0] * p[0, 0] + T[r, 1] * p[0, 1]) + T[r, 2] * p[0, 2]) + T[r, 3] for r in range(3)]
    assert out.tolist() == exp


# ---------------------------------------------------------------- registration_icp

def _numpy_icp_p2plane(src, tgt, nrm, r, init, iters):
    """Independent numpy/scipy statement of the same loop (cKDTree NN, numpy solve)."""
    T = init.copy()
    P = src @ T[:3, :3].T + T[:3, 3]
    tree = cKDTree(tgt)
    out = []
    for it in range(iters + 1):
        d, j = tree.query(P)
        m = d * d < r * r
        out.append((m.sum() / len(src), np.sqrt((d[m] ** 2).sum() / max(m.sum(), 1))))
        if it == iters:
            break
        s, t, n = P[m], tgt[j[m]], nrm[j[m]]
        res = ((s - t) * n).sum(1)
        J = np.hstack([np.cross(s, n), n])
        x = np.linalg.solve(J.T @ J, -(J.T @ res))
        U = np.eye(4)
        U[:3, :3] = (Rotation.from_euler("z", x[2]) * Rotation.from_euler("y", x[1]) * Rotation.from_euler("x", x[0])).as_matrix()
        U[:3, 3] = x[3:]
        T = U @ T
        P = P @ U[:3, :3].T + U[:3, 3]
    return T, out


def test_icp_matches_independent_numpy_statement(oracle):
    g = np.load(os.path.join(GOLD, "g3g4_icp_traces.npz"))
    for name in ("clean", "noisy"):
        scene = g[f"scene_{name}"]
        T, hist = _numpy_icp_p2plane(scene, g["model"], g["normals"], 10.0, g["init"], 20)
        o = oracle.icp(scene, g["model"], g["normals"], 10.0, g["init"], max_iter=20, rel_fitness=-1, rel_rmse=-1)
        assert np.abs(o["T"] - T).max() < 1e-8
        assert np.allclose(o["trace"][:, 0], [h[0] for h in hist], atol=0) and np.allclose(o["trace"][:, 1], [h[1] for h in hist], atol=1e-9)


@pytest.mark.parametrize("name", ["clean", "noisy"])
@pytest.mark.parametrize("est", [0, 1])
def test_g3g4_golden_traces(oracle, name, est):
    g = np.load(os.path.join(GOLD, "g3g4_icp_traces.npz"))
    for kd in (False, True):
        o = oracle.icp(g[f"scene_{name}"], g["model"], g["normals"], 10.0, g["init"], estimator=est, max_iter=20,
                       rel_fitness=-1, rel_rmse=-1, kdtree=kd)
        assert np.array_equal(o["corr"], g[f"corr_{name}_{est}"])
        assert np.abs(o["trace"] - g[f"trace_{name}_{est}"]).max() < 1e-12


def test_icp_known_answer_exact_correspondences(oracle):
    from pedp_hip import synth

    verts, tris, normals = synth.bumpy_torus(50, 40)
    model = verts.astype(np.float64)
    M = np.eye(4)
    M[:3, :3] = synth.axis_angle([0.3, -1.0, 0.5], np.deg2rad(1.5))
    M[:3, 3] = [0.4, -0.3, 0.5]
    scene = model[::3] @ M[:3, :3].T + M[:3, 3]
    for est in (0, 1):
        o = oracle.icp(scene, model, normals, 5.0, np.eye(4), estimator=est, max_iter=60, rel_fitness=1e-12, rel_rmse=1e-12)
        assert o["fitness"] == 1.0 and o["inlier_rmse"] < 1e-9
        assert np.abs(o["T"] - np.linalg.inv(M)).max() < 1e-9   # ground truth to 1e-9


def test_icp_conventions(oracle):
    from pedp_hip import synth

    verts, tris, normals = synth.bumpy_torus(16, 12)
    model = verts.astype(np.float64)
    scene = model[:40] + 0.05
    with pytest.raises(RuntimeError, match="normals"):
        oracle.icp(scene, model, None, 5.0, np.eye(4), estimator=0)
    far = np.eye(4)
    far[:3, 3] = 1e4
    o = oracle.icp(scene, model, normals, 1.0, far)
    assert o["fitness"] == 0 and o["inlier_rmse"] == 0 and o["iters"] == 1 and np.array_equal(o["T"], far)
    o = oracle.icp(scene, model, normals, 0.0, np.eye(4))     # radius <= 0: empty result
    assert o["fitness"] == 0 and (o["corr"] == -1).all()
    # strict inlier test: a point exactly at distance r is NOT a correspondence
    one = oracle.icp(model[:1] + [3.0, 0, 0], model[:1], normals[:1], 3.0, np.eye(4), max_iter=0)
    assert one["fitness"] == 0.0
    one = oracle.icp(model[:1] + [3.0, 0, 0], model[:1], normals[:1], 3.0000001, np.eye(4), max_iter=0)
    assert one["fitness"] == 1.0 and abs(one["inlier_rmse"] - 3.0) < 1e-12


# ---------------------------------------------------------------- cluster_poses / control flow

def test_g5_cluster_poses_golden_and_numpy_restatement(oracle):
    g = np.load(os.path.join(GOLD, "g5_cluster_poses.npz"))
    grid = g["grid"]
    assert len(grid) == 252
    assert np.array_equal(oracle.cluster_poses(30, 99999, grid, g["sym_id"]), g["keep_id"])
    assert np.array_equal(oracle.cluster_poses(30, 99999, grid, g["sym_z2"]), g["keep_z2"])
    assert len(g["keep_id"]) == 252 and len(g["keep_z2"]) == 126

    def restated(angle, dist, poses, syms):   # mycpp/src/app/pybind_api.cpp:24-68 in numpy float32
        keep = [0]
        thr = np.float32(angle / 180.0 * np.pi)
        for i in range(1, len(poses)):
            new = True
            for c in keep:
                if np.linalg.norm(poses[c][:3, 3] - poses[i][:3, 3]) >= dist:
                    continue
                for tf in syms:
                    R = (poses[i] @ tf)[:3, :3]
                    cos = np.float32((np.trace(R @ poses[c][:3, :3].T) - 1) / 2.0)
                    if np.arccos(np.clip(cos, -1, 1)) < thr:
                        new = False
                        break
                if not new:
                    break
            if new:
                keep.append(i)
        return np.array(keep, np.int32)

    sub = grid[::7]
    for ang in (30, 61, 95):
        assert np.array_equal(oracle.cluster_poses(ang, 99999, sub, g["sym_z2"]), restated(ang, 99999, sub, g["sym_z2"]))
    # translation gate: poses farther apart than dist_diff never merge
    shifted = sub.copy()
    shifted[1::2, :3, 3] += 10.0
    assert len(oracle.cluster_poses(180, 1.0, shifted, g["sym_id"])) == 2


def test_g6_improve_result_rng_order(oracle):
    g = np.load(os.path.join(GOLD, "g6_improve_result.npz"))
    f = np.load(os.path.join(GOLD, "g3g4_icp_traces.npz"))
    from pedp_hip import synth

    param = {"refine_registration": {"distance_threshold": 8.0}, "run_icp": {"fitness_threshold": 0.999, "rmse_threshold": 0.05}}
    np.random.seed(0)
    trace = []
    res = oracle.improve_result(f["scene_clean"], f["model"], f["normals"], synth.start_pose(), param, trace=trace)
    assert len(trace) == len(g["thresholds"]) == 50
    assert np.array_equal(np.array([t[0] for t in trace]), g["thresholds"])       # compounding threshold walk
    assert np.abs(np.array([t[3] for t in trace]) - g["T"]).max() < 1e-12
    assert res.fitness == float(g["best_fitness"]) and abs(np.random.uniform() - float(g["rng_after"])) == 0
    assert param["refine_registration"]["distance_threshold"] == 8.0             # caller's dict untouched (deepcopy)


# ------------------------------------------------------------------ depth pre-filters (G7)
def _py_erode(depth, radius, thr, ratio, zfar):
    """The warp kernel's loop nest (Utils.py:356-383) in plain Python on float32 scalars."""
    f = np.float32
    H, W = depth.shape
    out = np.zeros_like(depth)
    for h in range(H):
        for w in range(W):
            d_ori = depth[h, w]
            bad, total = f(0), f(0)
            for u in range(w - radius, w + radius + 1):
                if u < 0 or u >= W:
                    continue
                for v in range(h - radius, h + radius + 1):
                    if v < 0 or v >= H:
                        continue
                    cur = depth[v, u]
                    total += f(1)
                    if cur < f(0.001) or cur >= f(zfar) or abs(f(cur - d_ori)) > f(thr):
                        bad += f(1)
            out[h, w] = f(0) if f(bad / total) > f(ratio) else d_ori
    return out


def _py_bilateral(depth, radius, zfar, sD, sR):
    """Utils.py:304-345 in plain Python, float32 step by step."""
    f = np.float32
    H, W = depth.shape
    out = np.zeros_like(depth)
    for h in range(H):
        for w in range(W):
            mean, nv = f(0), 0
            win = [(u, v) for u in range(w - radius, w + radius + 1) if 0 <= u < W
                   for v in range(h - radius, h + radius + 1) if 0 <= v < H]
            for u, v in win:
                cur = depth[v, u]
                if cur >= f(0.001) and cur < f(zfar):
                    nv += 1
                    mean = f(mean + cur)
            if nv == 0:
                continue
            mean = f(mean / f(nv))
            c = depth[h, w]
            sw, sm = f(0), f(0)
            for u, v in win:
                cur = depth[v, u]
                if cur >= f(0.001) and cur < f(zfar) and abs(f(cur - mean)) < f(0.01):
                    a = f(-f((u - w) ** 2 + (h - v) ** 2) / f(f(f(2) * f(sD)) * f(sD)))
                    b = f(f(f(c - cur) * f(c - cur)) / f(f(f(2) * f(sR)) * f(sR)))
                    wgt = f(np.exp(f(a - b)))
                    sw = f(sw + wgt)
                    sm = f(sm + f(wgt * cur))
            if sw > 0:
                out[h, w] = f(sm / sw)
    return out


def test_depth_filters_golden_and_python_restatement(oracle):
    from pedp_hip import synth

    g = np.load(os.path.join(GOLD, "g7_depth_filters.npz"))
    same = lambda a, b: np.array_equal(a, b, equal_nan=True)  # noqa: E731
    assert same(oracle.erode_depth(g["depth"]), g["erode"])
    assert same(oracle.erode_depth(g["depth"], 3, 0.002, 0.5, 1.0), g["erode_r3"])
    assert same(oracle.bilateral_filter_depth(g["depth"]), g["bilateral"])
    assert same(oracle.bilateral_filter_depth(g["depth"], 1, 1.0, 1.5, 0.02), g["bilateral_r1"])
    assert same(oracle.depth2xyzmap(g["depth"], g["K"]), g["xyz"])
    assert same(oracle.depth2xyzmap_batch(g["depths"], g["Ks"], 0.9), g["xyz_batch"])
    # an independent statement of the same loops on a small image
    d = synth.depth_image(14, 19, seed=5)
    with np.errstate(invalid="ignore"):
        assert same(oracle.erode_depth(d, 2), _py_erode(d, 2, 0.001, 0.8, 100))
        assert same(oracle.erode_depth(d, 1, 0.003, 0.4, 0.9), _py_erode(d, 1, 0.003, 0.4, 0.9))
        pb = _py_bilateral(d, 2, 100, 2, 100000)
    ob = oracle.bilateral_filter_depth(d, 2)
    assert np.array_equal(np.isnan(ob), np.isnan(pb))
    m = ~np.isnan(ob)
    assert np.all(np.abs(ob[m] - pb[m]) <= 2e-6 * np.abs(pb[m]))  # numpy's float32 exp vs libm expf
    # numpy statement of depth2xyzmap (Utils.py:401-420)
    K = g["K"]
    dd = np.nan_to_num(d)
    vs, us = np.meshgrid(np.arange(14), np.arange(19), indexing="ij")
    pts = np.stack(((us - K[0, 2]) * dd / K[0, 0], (vs - K[1, 2]) * dd / K[1, 1], dd), -1).astype(np.float32)
    pts[dd < 0.001] = 0
    assert same(oracle.depth2xyzmap(dd, K), pts)


# ------------------------------------------------------------------ point-cloud operations (G8)
def test_cloud_ops_golden_and_independent_statements(oracle):
    g = np.load(os.path.join(GOLD, "g8_cloud_ops.npz"))
    pts = g["points"]
    # golden
    assert np.array_equal(oracle.voxel_down_sample(pts, 3.0)[0], g["voxel3"])
    assert np.array_equal(oracle.cluster_dbscan(pts, 6.0, 8), g["dbscan_6_8"])
    assert np.array_equal(oracle.knn_mean_distance(pts, 20), g["knn20"])
    assert np.array_equal(oracle.remove_statistical_outlier(pts, 20, 1.0), g["sor_20_1"])
    plane, inl = oracle.segment_plane(pts, 1.0, 100, seed=7)
    assert np.array_equal(plane, g["plane"]) and np.array_equal(inl, g["plane_inliers"])
    assert np.abs(oracle.estimate_normals(pts, 8.0, 12) - g["normals_8_12"]).max() < 1e-12
    # voxel grid: numpy statement (group by floor((p - (min - v/2)) / v), mean per group, lexicographic order)
    v = 3.0
    idx = np.floor((pts - (pts.min(axis=0) - v * 0.5)) / v).astype(np.int64)
    keys, inv = np.unique(idx, axis=0, return_inverse=True)
    mean = np.stack([np.bincount(inv.ravel(), pts[:, k]) for k in range(3)], 1) / np.bincount(inv.ravel())[:, None]
    assert g["voxel3"].shape == mean.shape and np.abs(g["voxel3"] - mean).max() < 1e-10
    # kNN mean distance: scipy's KD-tree (the point itself is its own first neighbour)
    d, _ = cKDTree(pts).query(pts, k=20)
    assert np.abs(d.mean(axis=1) - g["knn20"]).max() < 1e-10
    # DBSCAN: scikit-learn agrees on the noise set and on the partition of the core points
    sk = pytest.importorskip("sklearn.cluster")
    eps = 6.0
    model = sk.DBSCAN(eps=np.nextafter(eps, 0), min_samples=8).fit(pts)     # sklearn: d <= eps, Open3D: d < eps
    lab = g["dbscan_6_8"]
    core = np.zeros(len(pts), bool)
    core[model.core_sample_indices_] = True
    assert np.array_equal(lab == -1, model.labels_ == -1)
    pairs = set(zip(lab[core].tolist(), model.labels_[core].tolist()))
    assert len(pairs) == len(set(a for a, _ in pairs)) == len(set(b for _, b in pairs))   # one-to-one relabelling
    # plane: the table (z = 400), the inliers are those of a plane through three sampled points
    assert abs(abs(plane[2]) - 1) < 1e-3 and abs(abs(plane[3]) - 400) < 0.5 and len(inl) > 800
    assert len(set(oracle.sample3(7, 5, 1500))) == 3
    # normals: PCA over the same neighbourhoods with numpy's symmetric eigen solver
    tree = cKDTree(pts)
    est = g["normals_8_12"]
    for i in range(0, len(pts), 37):
        dd, jj = tree.query(pts[i], k=12, distance_upper_bound=8.0)
        jj = jj[np.isfinite(dd)]
        if len(jj) < 3:
            assert np.array_equal(est[i], [0, 0, 1])
            continue
        w, V = np.linalg.eigh(np.cov(pts[jj].T, bias=True))
        if w[1] - w[0] > 1e-6 * w[2]:
            assert abs(abs(est[i] @ V[:, 0]) - 1) < 1e-6


# ---------------------------------------------------------------- FPFH / feature matching (features.c)

def _fpfh_numpy(pts, nrm, radius, max_nn):
    """Independent restatement of Open3D's ComputeFPFHFeature (vector form, numpy's acos / arctan2)."""
    n = len(pts)
    d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    lists = []
    for i in range(n):
        cand = np.flatnonzero(d2[i] < radius * radius)
        cand = cand[np.lexsort((cand, d2[i, cand]))][:max_nn]
        lists.append(cand)
    spfh = np.zeros((n, 33))
    for i, cand in enumerate(lists):
        if len(cand) <= 1:
            continue
        incr = 100.0 / (len(cand) - 1)
        for j in cand[1:]:
            dp = pts[j] - pts[i]
            dist = np.sqrt(dp @ dp)
            f = np.zeros(3)
            if dist != 0.0:
                a, b = nrm[i], nrm[j]
                a1, a2 = a @ dp / dist, b @ dp / dist
                if np.arccos(abs(a1)) > np.arccos(abs(a2)):
                    a, b, dp, f[2] = nrm[j], nrm[i], -dp, -a2
                else:
                    f[2] = a1
                v = np.cross(dp, a)
                vn = np.sqrt(v @ v)
                if vn == 0.0:
                    f[:] = 0.0
                else:
                    v = v / vn
                    w = np.cross(a, v)
                    f[1] = v @ b
                    f[0] = np.arctan2(w @ b, a @ b)
            for k, x in enumerate((11 * (f[0] + np.pi) / (2 * np.pi), 11 * (f[1] + 1) * 0.5, 11 * (f[2] + 1) * 0.5)):
                spfh[i, 11 * k + min(max(int(np.floor(x)), 0), 10)] += incr
    out = np.zeros((n, 33))
    for i, cand in enumerate(lists):
        if len(cand) <= 1:
            continue
        s = np.zeros(3)
        for j in cand[1:]:
            if d2[i, j] == 0.0:
                continue
            val = spfh[j] / d2[i, j]
            out[i] += val
            s += val.reshape(3, 11).sum(1)
        s = np.where(s != 0.0, 100.0 / np.where(s != 0.0, s, 1.0), 0.0)
        out[i] = out[i] * np.repeat(s, 11) + spfh[i]
    return out


def test_fpfh_against_numpy_restatement_and_invariants(oracle):
    rng = np.random.default_rng(4)
    pts = rng.normal(0, 3.0, (90, 3)) * [1.0, 1.0, 0.15]          # a thick sheet
    pts[7] = pts[3]                                               # a duplicated point: zero distance is skipped
    nrm = rng.normal(size=pts.shape) * [0.3, 0.3, 1.0]
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    for radius, max_nn in ((2.5, 100), (2.5, 6), (0.4, 100)):
        got = oracle.fpfh(pts, nrm, radius, max_nn)
        ref = _fpfh_numpy(pts, nrm, radius, max_nn)
        assert got.shape == (90, 33)
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-9)
        thirds = got.reshape(90, 3, 11).sum(2)
        lonely = (thirds == 0).all(1)                             # no neighbour inside the radius: all zero
        assert np.allclose(thirds[~lonely], 200.0) or max_nn == 6 or radius == 0.4
        assert (got >= 0).all()
    # rigid motion of cloud and normals leaves the features unchanged (up to rounding in the bins)
    R = Rotation.from_euler("xyz", [0.3, -0.7, 1.1]).as_matrix()
    a = oracle.fpfh(pts, nrm, 2.5, 100)
    b = oracle.fpfh(pts @ R.T + [5.0, -2.0, 9.0], nrm @ R.T, 2.5, 100)
    assert np.mean(np.abs(a - b) > 1e-6) < 0.02


def test_feature_match_and_ransac_draw(oracle):
    rng = np.random.default_rng(6)
    ft = rng.uniform(0, 100, (400, 33))
    fs = ft[rng.integers(0, 400, 150)] + rng.normal(0, 1e-3, (150, 33))
    fs[10] = 0.5 * (ft[3] + ft[9])                                # equidistant from two targets: the lower index
    idx = oracle.feature_match(fs, ft)
    d = ((fs[:, None, :] - ft[None, :, :]) ** 2).sum(-1)
    assert np.array_equal(idx, d.argmin(1)) or np.allclose(d[np.arange(150), idx], d.min(1), rtol=1e-12)
    assert oracle.feature_match(fs, ft[:0]).tolist() == [-1] * 150
    # a RANSAC draw: exact correspondences under a rigid motion pass every checker and recover the motion
    src = rng.normal(0, 20, (300, 3))
    sn = rng.normal(size=(300, 3)); sn /= np.linalg.norm(sn, axis=1, keepdims=True)
    R = Rotation.from_euler("xyz", [0.4, 0.2, -0.9]).as_matrix()
    t = np.array([3.0, -8.0, 12.0])
    tgt, tn = src @ R.T + t, sn @ R.T
    corr = np.arange(300, dtype=np.int32)
    hits = 0
    for itr in range(40):
        ok, T = oracle.ransac_hypothesis(11, itr, src, sn, tgt, tn, corr, 0.9, 1.5, 0.5)
        if ok:
            hits += 1
            assert np.allclose(T[:3, :3], R, atol=1e-7) and np.allclose(T[:3, 3], t, atol=1e-6)
    assert hits >= 30                                            # (a draw with a repeated point may fail the fit)
    assert oracle.ransac_hypothesis(11, 5, src, sn, tgt, tn, corr, 0.9, 1.5, 0.5)[1].tolist() == \
        oracle.ransac_hypothesis(11, 5, src, sn, tgt, tn, corr, 0.9, 1.5, 0.5)[1].tolist()
    wrong = corr.copy(); wrong[::2] = rng.permutation(300)[:150]   # half the correspondences scrambled
    rejected = sum(not oracle.ransac_hypothesis(11, i, src, sn, tgt, tn, wrong, 0.9, 1.5, 0.5)[0] for i in range(200))
    assert rejected > 120
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    assert oracle.corres_inlier_ratio(src, tgt, corr, T, 0.01) == 1.0
    assert abs(oracle.corres_inlier_ratio(src, tgt, wrong, T, 0.01) - 0.5) < 0.02
