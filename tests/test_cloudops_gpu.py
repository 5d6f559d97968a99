"""GPU parity of the point-cloud operations of preprocess_source against
oracle/cloudops.c: voxel averages, DBSCAN labels, kept indices of the statistical outlier filter,
RANSAC inliers -- all bit-exact (float64 sums are taken in the oracle's order); the refitted plane
within 1e-12.  Through the C ABI (pedp_voxel_down_sample, pedp_cluster_dbscan,
pedp_knn_mean_distance, pedp_segment_plane)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(n_plane=6000, n_obj=2500, n_noise=150, seed=0):
    """A table plane, an object blob above it, a second smaller blob and scattered outliers (mm)."""
    rng = np.random.default_rng(seed)
    plane = np.column_stack([rng.uniform(-150, 150, n_plane), rng.uniform(-100, 100, n_plane), 400 + rng.normal(0, 0.4, n_plane)])
    obj = rng.normal([10, -5, 360], [18, 14, 9], (n_obj, 3))
    small = rng.normal([120, 70, 380], 4, (n_obj // 8, 3))
    noise = rng.uniform([-200, -150, 250], [200, 150, 450], (n_noise, 3))
    pts = np.vstack([plane, obj, small, noise])
    return pts[rng.permutation(len(pts))]


@pytest.mark.parametrize("voxel", [2.0, 5.0, 25.0])
def test_voxel_down_sample(ctx, oracle, voxel):
    from pedp_hip import cloud_ops

    pts = _scene()
    nrm = np.random.default_rng(1).normal(size=pts.shape)
    got, gotn = cloud_ops.voxel_down_sample(pts, voxel, nrm)
    ref, refn = oracle.voxel_down_sample(pts, voxel, nrm)
    assert got.shape == ref.shape and 10 < len(ref) < len(pts)
    assert np.array_equal(got, ref) and np.array_equal(gotn, refn)
    g2, n2 = cloud_ops.voxel_down_sample(pts, voxel)
    assert n2 is None and np.array_equal(g2, ref)
    # one point, all points in one voxel, empty, bad size
    one, _ = cloud_ops.voxel_down_sample(pts[:1], voxel)
    assert np.array_equal(one, pts[:1])
    big, _ = cloud_ops.voxel_down_sample(pts, 1e4)
    assert np.array_equal(big, oracle.voxel_down_sample(pts, 1e4)[0]) and len(big) <= 8
    assert len(cloud_ops.voxel_down_sample(np.zeros((0, 3)), voxel)[0]) == 0
    with pytest.raises(Exception, match="voxel_size"):
        cloud_ops.voxel_down_sample(pts, 0.0)
    with pytest.raises(Exception, match="too small"):
        cloud_ops.voxel_down_sample(pts, 1e-5)


@pytest.mark.parametrize("eps,min_points", [(10.0, 10), (4.0, 5), (2.0, 3), (60.0, 10)])
def test_cluster_dbscan_labels(ctx, oracle, eps, min_points):
    from pedp_hip import cloud_ops

    pts, _ = oracle.voxel_down_sample(_scene(seed=3), 3.0)
    got = cloud_ops.cluster_dbscan(pts, eps, min_points)
    ref = oracle.cluster_dbscan(pts, eps, min_points)
    assert np.array_equal(got, ref)
    assert ref.max() >= 0


def test_cluster_dbscan_edge_cases(ctx, oracle):
    from pedp_hip import cloud_ops

    rng = np.random.default_rng(5)
    # border points shared by two clusters, chains, duplicates, a single point, everything noise
    a = rng.normal([0, 0, 0], 1.0, (60, 3))
    b = rng.normal([9, 0, 0], 1.0, (60, 3))
    bridge = np.array([[4.5, 0, 0], [4.4, 0.1, 0], [4.6, -0.1, 0]])
    pts = np.vstack([a, bridge, b, a[:5]])
    for eps, mp in ((3.0, 8), (2.0, 4), (5.0, 20)):
        assert np.array_equal(cloud_ops.cluster_dbscan(pts, eps, mp), oracle.cluster_dbscan(pts, eps, mp))
    assert np.array_equal(cloud_ops.cluster_dbscan(pts[:1], 1.0, 1), [0])
    assert np.array_equal(cloud_ops.cluster_dbscan(pts[:1], 1.0, 2), [-1])
    far = rng.uniform(-1e3, 1e3, (200, 3))
    assert np.all(cloud_ops.cluster_dbscan(far, 1.0, 3) == -1)
    assert len(cloud_ops.cluster_dbscan(np.zeros((0, 3)), 1.0, 3)) == 0
    # huge extent against a small eps: the grid coarsens instead of overflowing
    wide = np.vstack([a, a + [5e6, 0, 0]])
    assert np.array_equal(cloud_ops.cluster_dbscan(wide, 3.0, 8), oracle.cluster_dbscan(wide, 3.0, 8))


@pytest.mark.parametrize("k,ratio", [(75, 0.01), (20, 1.0), (1, 0.5), (5000, 2.0)])
def test_statistical_outlier_removal(ctx, oracle, k, ratio):
    from pedp_hip import cloud_ops

    pts, _ = oracle.voxel_down_sample(_scene(seed=7), 4.0)
    if k > 300:
        with pytest.raises(Exception, match="1..300"):
            cloud_ops.knn_mean_distance(pts, k)
        return
    avg = cloud_ops.knn_mean_distance(pts, k)
    assert np.array_equal(avg, oracle.knn_mean_distance(pts, k))
    keep = cloud_ops.remove_statistical_outlier(pts, k, ratio)
    assert np.array_equal(keep, oracle.remove_statistical_outlier(pts, k, ratio))
    few = pts[:40]                       # fewer points than neighbours: all of them are "the k nearest"
    assert np.array_equal(cloud_ops.knn_mean_distance(few, 75), oracle.knn_mean_distance(few, 75))
    # all three kernels: a wave per query over ALL points (N <= 4096), a wave per query over the grid's
    # shells (larger clouds), and a thread per query when a query's shells overflow the wave's LDS share
    big, _ = oracle.voxel_down_sample(_scene(n_plane=14000, n_obj=5000, seed=8), 2.5)
    assert len(big) > 8193
    for sub in (pts[:3000], big[:4096], big[:4097], big[:8193]):
        assert np.array_equal(cloud_ops.knn_mean_distance(sub, k), oracle.knn_mean_distance(sub, k))
    rng = np.random.default_rng(3)
    clump = np.concatenate([big[:6000], big[100] + rng.normal(0, 1e-3, (2600, 3))])   # 2,600 points in one grid cell
    assert np.array_equal(cloud_ops.knn_mean_distance(clump, k), oracle.knn_mean_distance(clump, k))
    # queries whose cube grows to the whole grid without k points within its reach (a handful of stragglers far along
    # the diagonal from everything else), equal distances (a lattice: the selection's ties), duplicates of the query
    far = np.concatenate([rng.normal([400.0, 400.0, 400.0], 6.0, (4200, 3)), rng.uniform(0.0, 3.0, (30, 3))])
    assert np.array_equal(cloud_ops.knn_mean_distance(far, k), oracle.knn_mean_distance(far, k))
    g = np.arange(17, dtype=np.float64)
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)           # 4,913 points, many equal d^2
    lattice = np.concatenate([lattice, lattice[:200]])                                          # and 200 exact duplicates
    assert np.array_equal(cloud_ops.knn_mean_distance(lattice, k), oracle.knn_mean_distance(lattice, k))


@pytest.mark.parametrize("seed", [0, 1, 12345])
def test_segment_plane(ctx, oracle, seed):
    from pedp_hip import cloud_ops

    pts, _ = oracle.voxel_down_sample(_scene(seed=9), 3.0)
    plane, inl = cloud_ops.segment_plane(pts, 1.0, 3, 300, seed=seed)
    rplane, rinl = oracle.segment_plane(pts, 1.0, 300, seed=seed)
    assert np.array_equal(inl, rinl) and len(inl) > 1000
    assert np.abs(plane - rplane).max() < 1e-12
    assert abs(abs(plane[2]) - 1.0) < 1e-3 and abs(abs(plane[3]) - 400.0) < 1.0     # the table: z = 400
    # degenerate: fewer than three points, collinear points
    p0, i0 = cloud_ops.segment_plane(pts[:2], 1.0, 3, 10)
    assert not p0.any() and len(i0) == 0
    line = np.outer(np.arange(50.0), [1.0, 2.0, 3.0])
    p1, i1 = cloud_ops.segment_plane(line, 1.0, 3, 20)
    r1, ri1 = oracle.segment_plane(line, 1.0, 20)
    assert np.array_equal(i1, ri1) and np.array_equal(p1, r1)


def test_pointcloud_methods_follow_open3d_shapes(ctx, oracle):
    """The holder's Open3D-style methods as preprocess_source calls them."""
    from pedp_hip.compat import PointCloud

    pts = _scene(seed=11)
    pcd = PointCloud(pts)
    down = pcd.voxel_down_sample(voxel_size=5)
    assert isinstance(down, PointCloud) and np.array_equal(down.points, oracle.voxel_down_sample(pts, 5)[0])
    plane, inliers = down.segment_plane(distance_threshold=1.0, ransac_n=3, num_iterations=200)
    assert len(plane) == 4 and len(inliers) > 0 and int(inliers[0]) == inliers[0]
    rest = down.select_by_index(inliers, invert=True)
    assert len(rest.points) == len(down.points) - len(inliers)
    labels = np.array(rest.cluster_dbscan(eps=10, min_points=10, print_progress=True))
    assert np.array_equal(labels, oracle.cluster_dbscan(rest.points, 10, 10))
    clean, ind = rest.remove_statistical_outlier(nb_neighbors=75, std_ratio=0.01)
    assert np.array_equal(ind, oracle.remove_statistical_outlier(rest.points, 75, 0.01)) and len(clean.points) == len(ind)


def test_preprocess_source_flow_matches_oracle_chain(ctx, oracle):
    """preprocess_source (pose_estimation.py:186-268) on a rendered frame (object in front of a back
    plane): the same chain put together from the oracle's operations gives the same points."""
    from pedp_hip import synth
    from pedp_hip.compat import PointCloud, preprocess_source

    f = synth.Frame("parity")
    depth = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)["t_hit"]
    scene = f.scene(depth)                                     # object + back plane at z = 600, 0.5 mm noise
    param = {"preprocess_source": {"down_sample": 4, "plane_removal": {"distance_threshold": 2.0, "num_iterations": 200}},
             "box": False, "mesh": False}
    out, filtered, fpfh = preprocess_source(PointCloud(scene), None, param, i=0)
    assert filtered is out and fpfh is None
    # the oracle's chain
    down, _ = oracle.voxel_down_sample(scene, 4)
    plane, inl = oracle.segment_plane(down, 2.0, 200, seed=0)
    # first frame: the average normal (estimate_normals radius 2 / max_nn 5, 10-unit voxel mean) orients the plane
    nrm = oracle.estimate_normals(down, 2.0, 5)
    _, coarse = oracle.voxel_down_sample(down, 10, nrm)
    avg = coarse.mean(axis=0); avg /= np.linalg.norm(avg)
    assert abs(avg[2]) > 0.9
    rest = np.delete(down, inl, axis=0)                        # remove_plane (box = False)
    labels = oracle.cluster_dbscan(rest, 10, 10)
    ids, counts = np.unique(labels[labels >= 0], return_counts=True)
    big = rest[labels == ids[np.argmax(counts)]]
    ref = big[oracle.remove_statistical_outlier(big, 75, 0.01)]
    assert np.array_equal(out.points, ref) and 200 < len(ref) < len(down)
    assert out.has_normals() and np.abs(out.normals - oracle.estimate_normals(ref, 2.0, 5)).max() < 1e-9
    assert abs(abs(plane[2]) - 1) < 1e-3                       # the back plane was the segmented plane
    # box = True keeps the half space in front of the plane instead (background_removal returns its input)
    param["box"] = True
    boxed, _, _ = preprocess_source(PointCloud(scene), PointCloud(scene[::50]), param, i=1)
    assert param["preprocess_source"]["down_sample"] == 5      # tracking-frame mutation (:202-203)
    down5, _ = oracle.voxel_down_sample(scene, 5)
    pl5, _ = oracle.segment_plane(down5, 2.0, 200, seed=0)
    if np.dot(pl5[:3] / np.linalg.norm(pl5[:3]), [1, 1, 1]) < 0:
        pl5 = -pl5
    dist = (down5 @ pl5[:3] + pl5[3]) / np.sqrt((pl5[:3] ** 2).sum())
    half = down5[dist <= 0]
    lab = oracle.cluster_dbscan(half, 10, 10)
    ids, counts = np.unique(lab[lab >= 0], return_counts=True)
    big = half[lab == ids[np.argmax(counts)]]
    assert np.array_equal(boxed.points, big[oracle.remove_statistical_outlier(big, 75, 0.01)])
    with pytest.raises(NotImplementedError):
        preprocess_source(PointCloud(scene), None, dict(param, mesh=True), i=1)


@pytest.mark.parametrize("radius,max_nn", [(2.0, 5), (6.0, 30), (15.0, 100), (0.01, 5)])
def test_estimate_normals(ctx, oracle, radius, max_nn):
    from pedp_hip import cloud_ops

    pts, _ = oracle.voxel_down_sample(_scene(seed=13), 1.5)
    pts = pts[:6000]
    got = cloud_ops.estimate_normals(pts, radius, max_nn)
    ref = oracle.estimate_normals(pts, radius, max_nn)
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-9
    if radius >= 6.0:                                            # the table points get the table's normal
        table = np.abs(pts[:, 2] - 400) < 0.8
        assert np.mean(np.abs(got[table][:, 2]) > 0.95) > 0.8
    if radius == 0.01:                                           # nobody has three neighbours: Open3D's default
        assert np.array_equal(got, np.tile([0.0, 0.0, 1.0], (len(pts), 1)))
    # existing normals fix the orientation
    prior = np.tile([0.0, 0.0, -1.0], (len(pts), 1))
    flipped = cloud_ops.estimate_normals(pts, radius, max_nn, prior)
    assert np.abs(flipped - oracle.estimate_normals(pts, radius, max_nn, prior)).max() < 1e-9
    assert np.all((flipped * prior).sum(axis=1) >= 0)


def test_preprocess_target_estimates_normals_like_the_reference(ctx, oracle):
    """preprocess_target re-estimates the model's normals (pose_estimation.py:174) unless told to keep
    them; on a dense model they agree with the analytic normals up to sign."""
    from pedp_hip import synth
    from pedp_hip.compat import PointCloud, preprocess_target

    verts, tris, normals = synth.bumpy_torus(700, 300)           # 210,000 vertices, spacing ~0.5 mm
    sel = np.random.default_rng(0).choice(len(verts), 60000, replace=False)
    pts = verts[sel].astype(np.float64)
    out, feat = preprocess_target(PointCloud(pts), {"preprocess_target": {"max_pcd": 100000}})
    assert feat is None and out.has_normals()
    est = np.asarray(out.normals)
    cosang = np.abs((est * normals[sel]).sum(axis=1))
    assert np.mean(cosang > 0.9) > 0.9
    kept, _ = preprocess_target(PointCloud(pts, normals=normals[sel]), {"preprocess_target": {"max_pcd": 100000, "keep_normals": True}})
    assert np.array_equal(kept.normals, normals[sel])


def test_golden_g8(ctx):
    """The committed fixture (tests/golden/g8_cloud_ops.npz, generated by the oracle)."""
    import os
    from pedp_hip import cloud_ops

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_cloud_ops.npz"))
    pts = g["points"]
    assert np.array_equal(cloud_ops.voxel_down_sample(pts, 3.0)[0], g["voxel3"])
    assert np.array_equal(cloud_ops.cluster_dbscan(pts, 6.0, 8), g["dbscan_6_8"])
    assert np.array_equal(cloud_ops.knn_mean_distance(pts, 20), g["knn20"])
    assert np.array_equal(cloud_ops.remove_statistical_outlier(pts, 20, 1.0), g["sor_20_1"])
    plane, inl = cloud_ops.segment_plane(pts, 1.0, 3, 100, seed=7)
    assert np.array_equal(inl, g["plane_inliers"]) and np.abs(plane - g["plane"]).max() < 1e-12
    assert np.abs(cloud_ops.estimate_normals(pts, 8.0, 12) - g["normals_8_12"]).max() < 1e-9


def test_foreign_cloud_objects_take_the_gpu_methods(ctx, oracle):
    """An Open3D-style cloud (own CPU methods of the same names) handed to preprocess_source is
    wrapped, so the chain runs on the GPU methods and gives the holder's result."""
    from pedp_hip import synth
    from pedp_hip.compat import PointCloud, preprocess_source

    class Foreign:  # stands for o3d.geometry.PointCloud: same attribute surface, methods must not be called
        def __init__(self, pts):
            self.points = pts
            self.normals = np.zeros((0, 3))

        def has_normals(self):
            return False

        def has_colors(self):
            return False

        def voxel_down_sample(self, voxel_size):
            raise AssertionError("the foreign object's CPU method was called")

    f = synth.Frame("parity")
    depth = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)["t_hit"]
    scene = f.scene(depth)
    param = {"preprocess_source": {"down_sample": 4, "plane_removal": {"distance_threshold": 2.0, "num_iterations": 200}},
             "box": False, "mesh": False}
    a, _, _ = preprocess_source(Foreign(scene), None, param, i=0)
    b, _, _ = preprocess_source(PointCloud(scene), None, param, i=0)
    assert isinstance(a, PointCloud) and np.array_equal(a.points, b.points)


def test_device_resident_scene_through_the_voxel_grid(ctx, oracle):
    """A scene cloud that lives on the GPU (float64 N x 3 torch tensor, what the depth back-projection
    leaves there): `voxel_down_sample` works from the device array -- same averages bit for bit -- and
    a PointCloud holding the tensor goes through preprocess_source like the host copy, without the full
    cloud visiting the host unless `points` is read."""
    torch = pytest.importorskip("torch")
    from pedp_hip import cloud_ops
    from pedp_hip.compat import PointCloud, preprocess_source

    pts = _scene(n_plane=60000, n_obj=9000, seed=5)
    dev = torch.from_numpy(pts).cuda()
    got, none = cloud_ops.voxel_down_sample(dev, 2.0, ctx=ctx)
    ref, _ = cloud_ops.voxel_down_sample(pts, 2.0, ctx=ctx)
    assert none is None and np.array_equal(got, ref) and np.array_equal(got, oracle.voxel_down_sample(pts, 2.0)[0])
    assert np.array_equal(cloud_ops.voxel_down_sample(dev.float(), 2.0, ctx=ctx)[0],                 # other dtypes are converted
                          cloud_ops.voxel_down_sample(pts.astype(np.float32).astype(np.float64), 2.0, ctx=ctx)[0])
    holder = PointCloud(dev)
    assert len(holder) == len(pts) and holder.has_points() and not holder.has_normals() and holder._points is None
    holder.paint_uniform_color([1, 0, 0])
    assert holder.has_colors() and holder._points is None                                           # still only on the device
    params = {"preprocess_source": {"down_sample": 2, "plane_removal": {"distance_threshold": 2.0, "num_iterations": 200}},
              "box": False, "mesh": False}
    cloud_ops.set_ransac_seed(4)
    a, _, _ = preprocess_source(holder, None, params, i=0)
    assert holder._points is None
    cloud_ops.set_ransac_seed(4)
    b, _, _ = preprocess_source(PointCloud(pts), None, params, i=0)
    assert np.array_equal(a.points, b.points) and np.array_equal(a.normals, b.normals)
    assert np.array_equal(holder.points, pts) and holder.colors.shape == pts.shape                   # read: now it is made


def test_preprocess_source_in_one_call_equals_the_steps(ctx, oracle):
    """pedp_preprocess_source (the scene on the device between the stages) against preprocess_source through the
    single operations: the same points and normals in every bit, for the first frame and a tracking frame, from a host
    array and from a device tensor; a frame that leaves no cluster takes the step path (the reference's behaviour)."""
    torch = pytest.importorskip("torch")
    from pedp_hip import cloud_ops, icp_refine, synth
    from pedp_hip.compat import PointCloud, preprocess_source

    f = synth.Frame("parity")
    depth = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)["t_hit"]
    scene = f.scene(depth)
    for i in (0, 1):
        param = {"preprocess_source": {"down_sample": 4, "plane_removal": {"distance_threshold": 2.0, "num_iterations": 200}},
                 "box": False, "mesh": False}
        icp_refine._FORCE_STEPS = True
        try:
            steps, _, _ = preprocess_source(PointCloud(scene), None, {k: (dict(v) if isinstance(v, dict) else v) for k, v in param.items()}, i=i)
        finally:
            icp_refine._FORCE_STEPS = False
        one, _, fp = preprocess_source(PointCloud(scene), None, param, i=i)
        assert np.array_equal(one.points, steps.points) and len(one.points) > 200
        assert one.has_normals() == steps.has_normals() == (i == 0)
        if i == 0:
            assert np.array_equal(one.normals, steps.normals) and fp is None
        dev, _, _ = preprocess_source(PointCloud(torch.from_numpy(scene).cuda()), None, dict(param), i=i)
        assert np.array_equal(dev.points, steps.points) and (i != 0 or np.array_equal(dev.normals, steps.normals))
    # ---- the arguments run.py really passes (run.py:99-101, :154-156, :252-260): a background cloud, the root logger at
    # INFO; and param['box'].  One call again, the same cloud bit for bit, and the reference's two INFO lines with the
    # same text as the steps print them
    import logging

    background = PointCloud(scene[::7] + np.array([0.0, 0.0, 40.0]))
    root = logging.getLogger()
    level = root.level
    lines = []

    class Tap(logging.Handler):
        def emit(self, record):
            lines.append(record.getMessage())

    tap = Tap()
    root.addHandler(tap)
    try:
        for box in (False, True):
            for i in (0, 1):
                got = {}
                for name, force, lvl in (("steps", True, logging.INFO), ("one", False, logging.INFO), ("quiet", False, logging.WARNING)):
                    param = {"preprocess_source": {"down_sample": 4, "plane_removal": {"distance_threshold": 2.0, "num_iterations": 200}},
                             "box": box, "mesh": False}
                    root.setLevel(lvl)
                    del lines[:]
                    icp_refine._FORCE_STEPS = force
                    calls = []
                    fused = cloud_ops.preprocess_source_fused
                    cloud_ops.preprocess_source_fused = lambda *a, **k: (calls.append(k), fused(*a, **k))[1]
                    try:
                        cloud, _, _ = preprocess_source(PointCloud(scene), background, param, i=i)
                    finally:
                        icp_refine._FORCE_STEPS = False
                        cloud_ops.preprocess_source_fused = fused
                    assert len(calls) == (0 if force else 1)                # the one-call path really ran (and only once)
                    got[name] = (cloud, [ln for ln in lines if ln.startswith(":: ")])
                    assert param["preprocess_source"]["down_sample"] == (5 if i else 4)
                steps, one, quiet = got["steps"], got["one"], got["quiet"]
                assert np.array_equal(one[0].points, steps[0].points) and np.array_equal(quiet[0].points, steps[0].points)
                assert len(steps[0].points) > 200 and one[0].has_normals() == steps[0].has_normals() == (i == 0)
                if i == 0:
                    assert np.array_equal(one[0].normals, steps[0].normals) and np.array_equal(quiet[0].normals, steps[0].normals)
                assert one[1] == steps[1] and quiet[1] == []
                assert (i != 0) or any(ln.startswith(":: Average Normal for Source = [") for ln in one[1])
            if not box:     # without the box nothing of the background or the logging reaches the result
                assert np.array_equal(steps[0].points, preprocess_source(PointCloud(scene), None, dict(param), i=1)[0].points)
    finally:
        root.removeHandler(tap)
        root.setLevel(level)
    # the report itself: the plane segment_plane returns and compute_average_normal's mean, against the oracle
    p, n, counts, status, rep = cloud_ops.preprocess_source_fused(scene, 4, 2.0, 200, first_frame=True, ctx=ctx, report=True)
    down, _ = oracle.voxel_down_sample(scene, 4)
    plane_ref, inl_ref = oracle.segment_plane(down, 2.0, 200, seed=0)
    assert status == 0 and rep["inliers"] == len(inl_ref) and np.abs(rep["plane_model"] - plane_ref).max() < 1e-12
    _, coarse = oracle.voxel_down_sample(down, 10, oracle.estimate_normals(down, 2.0, 5))
    assert np.abs(rep["mean_normal"] - coarse.mean(axis=0)).max() < 1e-9
    # the average normal's other ways: a far outlier makes the input's box too large for dense cells (the sort-based grid
    # answers), a clump of more than 512 down-sampled points in one 10-unit voxel overflows a cell's list (likewise)
    far = np.vstack([scene, [[4.0e4, -3.0e4, 9.0e4]]])
    _, _, _, st_far, rep_far = cloud_ops.preprocess_source_fused(far, 4, 2.0, 200, first_frame=True, ctx=ctx, report=True)
    down_far, _ = oracle.voxel_down_sample(far, 4)
    _, coarse_far = oracle.voxel_down_sample(down_far, 10, oracle.estimate_normals(down_far, 2.0, 5))
    assert st_far == 0 and np.abs(rep_far["mean_normal"] - coarse_far.mean(axis=0)).max() < 1e-9
    clump = np.vstack([scene, np.random.default_rng(3).uniform([20, 20, 300], [29, 29, 309], (4000, 3))])
    _, _, _, st_c, rep_c = cloud_ops.preprocess_source_fused(clump, 0.5, 2.0, 200, first_frame=True, ctx=ctx, report=True)
    down_c, _ = oracle.voxel_down_sample(clump, 0.5)
    _, coarse_c = oracle.voxel_down_sample(down_c, 10, oracle.estimate_normals(down_c, 2.0, 5))
    assert st_c == 0 and np.abs(rep_c["mean_normal"] - coarse_c.mean(axis=0)).max() < 1e-9
    assert np.unique(np.floor((down_c - (down_c.min(0) - 5.0)) / 10.0), axis=0, return_counts=True)[1].max() > 512
    # the counters of the stages, and the statuses
    voxel = 4 if True else 0
    p, n, counts, status = cloud_ops.preprocess_source_fused(scene, voxel, 2.0, 200, first_frame=True, ctx=ctx)
    down, _ = oracle.voxel_down_sample(scene, voxel)
    assert status == 0 and counts[0] == len(down) and counts[1] > counts[2] >= counts[3] == len(p) and n.shape == p.shape
    flat = np.c_[np.random.default_rng(0).uniform(0, 100, (4000, 2)), np.zeros(4000)]      # nothing but the plane
    p, n, counts, status = cloud_ops.preprocess_source_fused(flat, 2.0, 1.0, 50, first_frame=False, ctx=ctx)
    assert status == 1 and len(p) == 0 and n is None and counts[1] == 0
    p, n, counts, status = cloud_ops.preprocess_source_fused(flat[:2], 2.0, 1.0, 50, first_frame=False, ctx=ctx)
    assert status == 2 and len(p) == 0
