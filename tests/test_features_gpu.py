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
