"""GPU parity of the feature-based global registration (src/pose_estimation.py:132-137, :467-503,
:524-545) against the oracle (oracle/features.c; Open3D 0.18 restated, parity unpinned): FPFH
features, nearest-feature correspondences, the RANSAC draws and the whole
registration_ransac_based_on_feature_matching / run_icp flow.  Everything through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _surface(n, seed, noise=0.0):
    """Points and outward normals on a bumpy ellipsoid (mm)."""
    rng = np.random.default_rng(seed)
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    r = 1.0 + 0.15 * np.sin(5 * u[:, 0]) * np.cos(4 * u[:, 1]) + 0.1 * np.sin(7 * u[:, 2])
    pts = u * r[:, None] * [40.0, 28.0, 18.0]
    nrm = u / [40.0, 28.0, 18.0]
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return pts + rng.normal(0, noise, pts.shape), nrm


@pytest.mark.parametrize("n,radius,max_nn", [(1500, 8.0, 100), (1500, 8.0, 12), (400, 3.0, 100), (3, 50.0, 100), (1, 1.0, 10)])
def test_fpfh_matches_oracle(ctx, oracle, n, radius, max_nn):
    from pedp_hip import cloud_ops

    pts, nrm = _surface(n, seed=n + max_nn)
    got = cloud_ops.compute_fpfh(pts, nrm, radius, max_nn, ctx=ctx)
    ref = oracle.fpfh(pts, nrm, radius, max_nn)
    assert got.shape == ref.shape == (n, 33)
    # the same float64 operations in the same order; acos / atan2 come from two libms, so a pair whose
    # angle sits within an ulp of a bin edge (or of the |angle1| = |angle2| swap) may land one bin over
    off = np.abs(got - ref) > 1e-9 * (1.0 + np.abs(ref))
    assert off.any(axis=1).mean() <= 0.002, f"{off.any(axis=1).sum()} of {n} points differ"
    if n == 1500:
        assert np.allclose(ref.reshape(n, 3, 11).sum(2), 200.0)      # every point has neighbours: each third sums to 200


def test_fpfh_edge_cases(ctx, oracle):
    from pedp_hip import _lib, cloud_ops

    pts, nrm = _surface(300, seed=2)
    pts[5] = pts[4]                                   # duplicate: zero distance skipped in the weighting
    flat_n = np.tile([0.0, 0.0, 1.0], (300, 1))       # the (0, 0, 1) normals Open3D gives sparse clouds
    for normals in (nrm, flat_n):
        got, ref = cloud_ops.compute_fpfh(pts, normals, 10.0, 50, ctx=ctx), oracle.fpfh(pts, normals, 10.0, 50)
        assert (np.abs(got - ref) > 1e-9 * (1 + np.abs(ref))).any(axis=1).mean() <= 0.01
    # more than 1,024 / 4,096 points inside one query's radius: the wave kernel's larger instantiation, then the
    # thread-per-query kernel, take the cloud
    rng = np.random.default_rng(9)
    for extra in (2300, 4400):
        clump = np.vstack([pts, pts[7] + rng.normal(0, 0.4, (extra, 3))])
        cn = np.vstack([nrm, rng.normal(size=(extra, 3))])
        cn /= np.linalg.norm(cn, axis=1, keepdims=True)
        got, ref = cloud_ops.compute_fpfh(clump, cn, 6.0, 100, ctx=ctx), oracle.fpfh(clump, cn, 6.0, 100)
        assert (np.abs(got - ref) > 1e-9 * (1 + np.abs(ref))).any(axis=1).mean() <= 0.01
    assert cloud_ops.compute_fpfh(np.zeros((0, 3)), np.zeros((0, 3)), 1.0, 10, ctx=ctx).shape == (0, 33)
    with pytest.raises(_lib.PedpError, match="max_nn"):
        cloud_ops.compute_fpfh(pts, nrm, 1.0, 500, ctx=ctx)
    with pytest.raises(RuntimeError):
        cloud_ops.compute_fpfh(pts, nrm[:10], 1.0, 10, ctx=ctx)


def test_feature_match_bit_exact(ctx, oracle):
    from pedp_hip import cloud_ops

    sp, sn = _surface(900, seed=3)
    tp, tn = _surface(2500, seed=4)
    fs, ft = oracle.fpfh(sp, sn, 9.0, 100), oracle.fpfh(tp, tn, 9.0, 100)
    fs[17] = ft[40]; ft[41] = ft[40]                  # an exact tie between two targets: the lower index
    got = cloud_ops.match_features(fs, ft, ctx=ctx)
    assert np.array_equal(got, oracle.feature_match(fs, ft)) and got[17] == 40
    assert cloud_ops.match_features(fs[:0], ft, ctx=ctx).shape == (0,)
    assert cloud_ops.match_features(fs[:5], ft[:0], ctx=ctx).tolist() == [-1] * 5
    assert np.array_equal(cloud_ops.match_features(fs[:65], ft[:1], ctx=ctx), np.zeros(65, np.int32))


def _pair(n_src, n_tgt, seed, noise=0.05):
    """A target surface with normals and a source = part of the same surface in another pose."""
    from scipy.spatial.transform import Rotation

    tp, tn = _surface(n_tgt, seed)
    rng = np.random.default_rng(seed + 100)
    keep = np.flatnonzero(tp[:, 2] > -6.0)                        # the camera sees the upper part
    keep = rng.choice(keep, min(n_src, len(keep)), replace=False)
    R = Rotation.from_euler("xyz", [0.5, -0.3, 0.8]).as_matrix()
    t = np.array([25.0, -40.0, 320.0])
    sp = (tp[keep] + rng.normal(0, noise, (len(keep), 3))) @ R.T + t      # model -> scene
    sn = tn[keep] @ R.T
    T_scene_to_model = np.eye(4)
    T_scene_to_model[:3, :3], T_scene_to_model[:3, 3] = R.T, -R.T @ t
    return sp, sn, tp, tn, T_scene_to_model


def test_ransac_draws_match_oracle(ctx, oracle):
    from pedp_hip import _lib

    sp, sn, tp, tn, _ = _pair(700, 2000, seed=5)
    rng = np.random.default_rng(0)
    fs, ft = oracle.fpfh(sp, sn, 9.0, 100), oracle.fpfh(tp, tn, 9.0, 100)
    corr = oracle.feature_match(fs, ft)
    d_src, d_tgt = _lib.Cloud(ctx, sp, sn), _lib.Cloud(ctx, tp, tn)
    for with_normals in (True, False):
        a, b = (d_src, d_tgt) if with_normals else (_lib.Cloud(ctx, sp, None), _lib.Cloud(ctx, tp, None))
        ok, T = _lib.ransac_hypotheses(ctx, a, b, corr, 99, 1000, 3000, 0.9, 4.0, 0.6)
        assert 0 < ok.sum() < 3000
        for k in list(range(0, 3000, 37)) + list(np.flatnonzero(ok)[:60]):
            ok_ref, T_ref = oracle.ransac_hypothesis(99, 1000 + k, sp, sn if with_normals else None, tp,
                                                     tn if with_normals else None, corr, 0.9, 4.0, 0.6)
            assert bool(ok[k]) == ok_ref, k
            assert np.allclose(T[k], T_ref, rtol=0, atol=1e-9 * (1 + np.abs(T_ref).max())), k
    with pytest.raises(_lib.PedpError, match="outside the target"):
        bad = corr.copy(); bad[3] = len(tp)
        _lib.ransac_hypotheses(ctx, d_src, d_tgt, bad, 1, 0, 10, 0.9, 4.0, 0.6)
    assert _lib.ransac_hypotheses(ctx, d_src, d_tgt, corr, 1, 0, 0, 0.9, 4.0, 0.6)[0].shape == (0,)
    del rng


def _oracle_global_registration(oracle, sp, sn, tp, tn, fs, ft, max_dist, edge, angle, iters, conf, seed):
    """The sequential statement of registration_ransac_based_on_feature_matching on the oracle's pieces."""
    corr = oracle.feature_match(fs, ft)
    best = (0.0, 0.0, np.eye(4))
    budget, itr = iters, 0
    while itr < budget:
        ok, T = oracle.ransac_hypothesis(seed, itr, sp, sn, tp, tn, corr, edge, max_dist, angle)
        if ok:
            # EvaluateRANSACBasedOnCorrespondence (open3d 0.18): the pairs, one after the other
            good, err2 = 0, 0.0
            for i, j in enumerate(corr):
                q = T[:3, :3] @ sp[i] + T[:3, 3] - tp[j]
                d2 = float(q @ q)
                if d2 < max_dist * max_dist:
                    good += 1
                    err2 += d2
            fit_k, rmse_k = (good / len(corr), np.sqrt(err2 / good)) if good else (0.0, 0.0)
            if fit_k > best[0] or (fit_k == best[0] and rmse_k < best[1]):
                best = (fit_k, rmse_k, T)
                ratio = fit_k
                if 0.0 < ratio < 1.0 and conf < 1.0:
                    k_est = np.log(1.0 - conf) / np.log(1.0 - ratio ** 3)
                    if k_est < budget:
                        budget = int(np.ceil(k_est))
                elif ratio >= 1.0:
                    budget = min(budget, itr + 1)
        itr += 1
    return best


def test_global_registration_flow_matches_oracle_and_finds_the_pose(ctx, oracle):
    from pedp_hip import compat
    from pedp_hip import registration as reg

    sp, sn, tp, tn, T_true = _pair(600, 1800, seed=8)
    src, tgt = compat.PointCloud(sp, normals=sn), compat.PointCloud(tp, normals=tn)
    search = compat.KDTreeSearchParamHybrid(radius=10.0, max_nn=100)
    fs, ft = compat.compute_fpfh_feature(src, search, ctx=ctx), compat.compute_fpfh_feature(tgt, search, ctx=ctx)
    assert fs.data.shape == (33, len(sp)) and fs.dimension() == 33 and ft.num() == len(tp)
    checkers = [compat.CorrespondenceCheckerBasedOnEdgeLength(0.9), compat.CorrespondenceCheckerBasedOnDistance(3.0),
                compat.CorrespondenceCheckerBasedOnNormal(0.5)]
    reg.set_ransac_seed(1234)
    try:
        res = compat.registration_ransac_based_on_feature_matching(
            src, tgt, fs, ft, False, 3.0, compat.TransformationEstimationPointToPoint(False), 3, checkers,
            compat.RANSACConvergenceCriteria(20000, 0.999), ctx=ctx)
        reg.set_ransac_seed(1234)
        again = compat.registration_ransac_based_on_feature_matching(
            src, tgt, fs, ft, False, 3.0, compat.TransformationEstimationPointToPoint(False), 3, checkers,
            compat.RANSACConvergenceCriteria(20000, 0.999), ctx=ctx)
    finally:
        reg.set_ransac_seed(None)
    assert np.array_equal(res.transformation, again.transformation) and res.fitness == again.fitness   # a seeded run repeats
    fit, rmse, T = _oracle_global_registration(oracle, sp, sn, tp, tn, fs.data.T, ft.data.T, 3.0, 0.9, 0.5, 20000, 0.999, 1234)
    assert res.fitness == fit and abs(res.inlier_rmse - rmse) < 1e-9 and np.allclose(res.transformation, T, atol=1e-8)
    # fitness is the share of the feature PAIRS that are inliers (open3d 0.18 scores the pairs); and it is the pose:
    # the bulk of the scene lands on the model (an independent KD-tree look), close to the truth
    from scipy.spatial import cKDTree
    assert 0.0 < res.fitness <= 1.0 and len(res.correspondence_set) == round(res.fitness * len(sp))
    moved = sp @ res.transformation[:3, :3].T + res.transformation[:3, 3]
    assert (cKDTree(tp).query(moved)[0] < 3.0).mean() > 0.8
    err = res.transformation @ np.linalg.inv(T_true)
    assert np.abs(err[:3, 3]).max() < 3.0 and np.abs(err[:3, :3] - np.eye(3)).max() < 0.08
    assert 0 < res.validated_draws < 20000
    # degenerate calls give the empty result like Open3D
    none = compat.registration_ransac_based_on_feature_matching(src, tgt, fs, ft, False, 0.0)
    assert none.fitness == 0.0 and np.array_equal(none.transformation, np.eye(4))


def test_determine_pose_with_global_registration(ctx, oracle):
    """determine_pose(icp=True) (pose_estimation.py:686-747) on a raw scene (object in front of a back
    plane) WITHOUT any start pose: preprocess with FPFH features in both sections, RANSAC global
    registration, one refinement, repeated until run_icp's thresholds hold, then the randomised
    restarts -- the model lands on the object.  (The scene cloud carries normals towards the camera:
    estimate_normals keeps the side of normals that are already there, so the re-estimated scene normals
    face outwards like the model's.)"""
    from scipy.spatial import cKDTree
    from scipy.spatial.transform import Rotation

    from pedp_hip import registration as reg
    from pedp_hip.compat import PointCloud, determine_pose

    rng = np.random.default_rng(12)
    mp, mn = _surface(26000, seed=21)                               # the model, ~1 point per mm^2
    R = Rotation.from_euler("xyz", [0.4, 2.9, -0.7]).as_matrix()
    t = np.array([12.0, -9.0, 350.0])
    op, on = mp @ R.T + t, mn @ R.T                                 # the object in the camera frame
    seen = np.einsum("ij,ij->i", on, -op / np.linalg.norm(op, axis=1, keepdims=True)) > 0.25
    gx, gy = np.meshgrid(np.arange(-110.0, 110.0, 1.0), np.arange(-90.0, 90.0, 1.0))
    plane = np.column_stack([gx.ravel(), gy.ravel(), np.full(gx.size, 420.0)])
    scene = np.vstack([op[seen], plane]) + rng.normal(0, 0.05, (int(seen.sum()) + len(plane), 3))
    scene = scene[rng.permutation(len(scene))]
    src = PointCloud(scene, normals=-scene / np.linalg.norm(scene, axis=1, keepdims=True))
    tgt = PointCloud(mp, normals=mn)
    params = {"preprocess_target": {"max_pcd": 100000, "keep_normals": True, "fpfh_radius": 8.0, "fpfh_max_nn": 100},
              "preprocess_source": {"down_sample": 1, "plane_removal": {"distance_threshold": 1.0, "num_iterations": 300},
                                    "fpfh_radius": 8.0, "fpfh_max_nn": 100},
              "box": False, "mesh": False,
              "execute_global_registration": {"distance_threshold": 2.0, "correspondence_checkers": [{"value": 0.9}],
                                              "angle_threshold": 0.6, "ransac_criteria": {"iterations": 100000, "confidence": 0.999}},
              "refine_registration": {"distance_threshold": 3.0}, "run_icp": {"fitness_threshold": 0.95, "rmse_threshold": 1.0}}
    np.random.seed(3)
    reg.set_ransac_seed(77)
    try:
        moved, best, z, tgt_proc = determine_pose(src, tgt, None, None, params, icp=True)
    finally:
        reg.set_ransac_seed(None)
    assert z == 0 and tgt_proc is tgt and len(moved.points) == len(mp)
    assert best.fitness >= 0.95 and best.inlier_rmse <= 1.0
    d, _ = cKDTree(np.asarray(moved.points)).query(op[seen])
    assert np.percentile(d, 95) < 1.5
    assert np.abs(np.linalg.inv(best.transformation)[:3, 3] - t).max() < 2.0
