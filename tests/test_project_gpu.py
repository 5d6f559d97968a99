"""GPU parity of the fused defect projection against the oracle's restatement of
heatmap_to_points + compute_rays + intersect_rays_with_mesh: selected pixels, hit count, pixel
list and triangle indices bit-exact; hit points within 1e-9 mm (float64 o + d * t).  Posable
meshes: records rebuilt on the device from the float64 model vertices and a pose give bit-exact
t_hit / primitive_ids against the oracle on pose_vertices()."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POINT_TOL = 1e-9


def _heatmap(h, w, seed, fill=0.3):
    rng = np.random.default_rng(seed)
    hm = rng.uniform(0.0, 1.0, size=(h, w))
    hm[rng.uniform(size=(h, w)) > fill] = 0.0  # sparse blobs like an anomaly map
    hm[0, 0] = np.nan                          # NaN never passes `>`
    return hm


def _check(out, ref):
    assert out["n_rays"] == ref["n_rays"]
    assert np.array_equal(out["pixels"], ref["pixels"])
    assert np.array_equal(out["primitive_ids"], ref["primitive_ids"])
    assert np.array_equal(out["intensities"], ref["intensities"])
    assert out["points"].shape == ref["points"].shape
    if len(ref["points"]):
        assert np.abs(out["points"] - ref["points"]).max() < POINT_TOL


@pytest.mark.parametrize("config,threshold", [("tiny", 0.5), ("tiny", -1.0), ("parity", 0.75), ("parity", 0.0)])
def test_project_heatmap_matches_oracle(ctx, oracle, config, threshold):
    from pedp_hip import _lib, synth

    f = synth.Frame(config)
    hm = _heatmap(f.height, f.width, 3)
    mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
    out = mesh.project_heatmap(hm, f.K, threshold)
    ref = oracle.project_heatmap(f.verts_posed, f.tris, hm, f.K, threshold)
    assert len(ref["points"]) > 5
    _check(out, ref)


def test_project_heatmap_edge_cases(ctx, oracle):
    from pedp_hip import _lib, synth

    f = synth.Frame("tiny")
    mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
    # nothing selected
    out = mesh.project_heatmap(np.zeros((f.height, f.width)), f.K, 0.5)
    assert out["n_rays"] == 0 and len(out["points"]) == 0
    # selected, but every ray misses (camera looks away: flip the mesh behind the camera)
    behind = _lib.Mesh(ctx, f.verts_posed * np.array([1, 1, -1], np.float32), f.tris)
    hm = np.ones((f.height, f.width))
    out = behind.project_heatmap(hm, f.K, 0.5)
    assert out["n_rays"] == f.height * f.width and len(out["points"]) == 0
    # a single hot pixel, a non-zero origin, a 1 x 1 image, an empty image
    hm = np.zeros((f.height, f.width)); hm[f.height // 2, f.width // 2] = 0.9
    _check(mesh.project_heatmap(hm, f.K, 0.5), oracle.project_heatmap(f.verts_posed, f.tris, hm, f.K, 0.5))
    o = (1.5, -2.0, 3.25)
    hm = _heatmap(f.height, f.width, 9)
    _check(mesh.project_heatmap(hm, f.K, 0.2, o), oracle.project_heatmap(f.verts_posed, f.tris, hm, f.K, 0.2, o))
    K1 = np.array([[500.0, 0, 0.0], [0, 500.0, 0.0], [0, 0, 1]])
    _check(mesh.project_heatmap(np.ones((1, 1)), K1, 0.5), oracle.project_heatmap(f.verts_posed, f.tris, np.ones((1, 1)), K1, 0.5))
    out = mesh.project_heatmap(np.zeros((0, 0)), f.K, 0.5)
    assert out["n_rays"] == 0
    # capacity too small is an error, not a truncation
    import ctypes as C
    hm = np.ones((f.height, f.width))
    cam = _lib.Pinhole(f.K[0, 0], f.K[1, 1], f.K[0, 2], f.K[1, 2], f.width, f.height)
    pts = np.empty((4, 3)); it = np.empty(4); nr, nh = C.c_int64(), C.c_int64()
    org = np.zeros(3)
    rc = _lib.load().pedp_project_heatmap(ctx._h, mesh._h, C.byref(cam), _lib._ptr(hm), 0.5, _lib._ptr(org), _lib.HOST, 4,
                                          _lib._ptr(pts), _lib._ptr(it), None, None, C.byref(nr), C.byref(nh))
    assert rc == -1 and nh.value > 4 and b"capacity" in _lib.load().pedp_last_error()


def test_posable_mesh_matches_host_transform(ctx, oracle):
    from pedp_hip import _lib, synth

    f = synth.Frame("parity")
    model = f.model_points
    posable = _lib.Mesh(ctx, model, f.tris, posable=True)
    # identity pose = plain float32 cast
    ref = oracle.raycast(model.astype(np.float32), f.tris, f.rays6, bvh=True)
    res = posable.cast_rays(f.rays6)
    assert np.array_equal(res["primitive_ids"], ref["primitive_ids"])
    assert np.array_equal(res["t_hit"].view(np.uint32), ref["t_hit"].view(np.uint32))
    rng = np.random.default_rng(11)
    for k in range(3):
        T = f.T_gt.copy()
        T[:3, 3] += rng.normal(0, 4.0, 3)
        T[:3, :3] = T[:3, :3] @ oracle.rot_xyz(rng.normal(0, 0.05, 3))
        posable.set_pose(T)
        v32 = oracle.pose_vertices(T, model)
        # the BLAS-ordered host transform may differ from the fixed order by float64 ulps only
        host = (model @ T[:3, :3].T + T[:3, 3])
        assert np.abs(host - v32).max() < 1e-4
        ref = oracle.raycast(v32, f.tris, f.rays6, bvh=True)
        res = posable.cast_rays(f.rays6)
        assert np.isfinite(ref["t_hit"]).sum() > 50
        assert np.array_equal(res["primitive_ids"], ref["primitive_ids"])
        assert np.array_equal(res["t_hit"].view(np.uint32), ref["t_hit"].view(np.uint32))
    with pytest.raises(_lib.PedpError):
        _lib.Mesh(ctx, f.verts_posed, f.tris).set_pose(np.eye(4))


def test_frame_projector_and_ray_tracing(ctx, oracle, tmp_path):
    """ray_tracing (reference signature) and the resident-model FrameProjector give the
    oracle's cloud for the same frame."""
    import json
    from pedp_hip import compat, synth
    from pedp_hip.ray_projection import FrameProjector

    f = synth.Frame("tiny")
    c2d = np.eye(4); c2d[:3, 3] = (32.0, -1.5, 2.0)
    c2d[:3, :3] = oracle.rot_xyz([0.01, -0.02, 0.005])
    (tmp_path / "configs").mkdir()
    d2c = np.linalg.inv(c2d)
    (tmp_path / "configs" / "camera_extrinsics.json").write_text(json.dumps({
        "color_to_depth": {"rotation_matrix": c2d[:3, :3].tolist(), "translation_vector": [c2d[:3, 3].tolist()]},
        "depth_to_color": {"rotation_matrix": d2c[:3, :3].tolist(), "translation_vector": [d2c[:3, 3].tolist()]}}))
    hm = _heatmap(f.height, f.width, 21, fill=0.6)
    intr = compat.PinholeCameraIntrinsic(f.width, f.height, intrinsic_matrix=f.K)
    model = compat.TriangleMesh(f.model_points, f.tris)
    posed = compat.transform_object(model, f.T_gt)
    cloud, moved = compat.ray_tracing(str(tmp_path), posed, hm, intr, 0.6)
    v32 = np.asarray(moved.vertices, np.float64).astype(np.float32)
    ref = oracle.project_heatmap(v32, f.tris, hm, f.K, 0.6)
    assert len(ref["points"]) > 5
    assert np.abs(np.asarray(cloud.points) - ref["points"]).max() < POINT_TOL
    det = {}
    cloud2 = FrameProjector(model, intr, c2d).project(f.T_gt, hm, 0.6, det)
    ref2 = oracle.project_heatmap(oracle.pose_vertices(np.linalg.inv(c2d) @ f.T_gt, f.model_points), f.tris, hm, f.K, 0.6)
    _check(det, ref2)
    assert np.array_equal(np.asarray(cloud2.colors), np.asarray(compat.create_intersection_pcd(ref2["points"], ref2["intensities"]).colors))


def test_projection_with_colours_move_and_resident_heat_map(ctx, oracle):
    """pedp_project_heatmap_ex: the hits' jet colours (create_intersection_pcd's arithmetic, bit for bit), the trailing
    `.transform(color_to_depth)` on the device (the oracle's operation order, bit for bit), the heat map as float32 and
    as a CUDA tensor (taken where and as it is: same selection, same hits), and the viewer's float64 posed vertices."""
    torch = pytest.importorskip("torch")
    from pedp_hip import _lib, compat, synth
    from pedp_hip.ray_projection import FrameProjector, _jet_lut

    f = synth.Frame("parity")
    hm = _heatmap(f.height, f.width, 8, fill=0.5)
    c2d = np.eye(4)
    c2d[:3, :3] = oracle.rot_xyz([0.02, -0.01, 0.03])
    c2d[:3, 3] = (31.0, -2.5, 1.75)
    mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
    ref = oracle.project_heatmap(f.verts_posed, f.tris, hm, f.K, 0.6)
    assert len(ref["points"]) > 50
    plain = mesh.project_heatmap(hm, f.K, 0.6)
    out = mesh.project_heatmap(hm, f.K, 0.6, jet_lut=_jet_lut(), post=c2d)
    for k in ("pixels", "primitive_ids", "intensities"):
        assert np.array_equal(out[k], ref[k])
    assert np.array_equal(out["points"], oracle.transform(c2d, plain["points"]))
    assert np.array_equal(out["colors"], np.asarray(compat.create_intersection_pcd(ref["points"], ref["intensities"]).colors))
    # equal intensities divide by zero like the reference: NaN -> black
    flat = np.where(hm > 0.6, 0.8, 0.0)
    black = mesh.project_heatmap(flat, f.K, 0.6, jet_lut=_jet_lut())
    assert len(black["colors"]) == len(ref["points"]) and not black["colors"].any()
    # float32 heat map: upcast exactly -- the selection and the hits of its float64 copy
    hm32 = np.nan_to_num(hm, nan=0.0).astype(np.float32)
    ref32 = oracle.project_heatmap(f.verts_posed, f.tris, hm32.astype(np.float64), f.K, 0.6)
    out32 = mesh.project_heatmap(hm32, f.K, 0.6, jet_lut=_jet_lut())
    _check(out32, ref32)
    assert np.array_equal(out32["colors"], np.asarray(compat.create_intersection_pcd(ref32["points"], ref32["intensities"]).colors))
    # resident heat maps (CUDA tensors, both types)
    for dev in (torch.from_numpy(hm32).cuda(), torch.from_numpy(np.nan_to_num(hm, nan=0.0)).cuda()):
        got = mesh.project_heatmap(dev, f.K, 0.6, jet_lut=_jet_lut())
        want = ref32 if dev.dtype == torch.float32 else oracle.project_heatmap(f.verts_posed, f.tris, np.nan_to_num(hm, nan=0.0), f.K, 0.6)
        _check(got, want)
    # fewer, then more hits than the room kept from the last call
    few = np.zeros_like(hm); few[ref["pixels"][0][1], ref["pixels"][0][0]] = 1.0
    assert len(mesh.project_heatmap(few, f.K, 0.5)["points"]) == 1
    _check(mesh.project_heatmap(np.ones_like(hm), f.K, 0.5), oracle.project_heatmap(f.verts_posed, f.tris, np.ones_like(hm), f.K, 0.5))
    # the viewer's copy of the posed mesh
    intr = compat.PinholeCameraIntrinsic(f.width, f.height, intrinsic_matrix=f.K)
    model = compat.TriangleMesh(f.model_points, f.tris)
    proj = FrameProjector(model, intr, c2d)
    posed = proj.posed_mesh(f.T_gt, model)
    assert np.array_equal(posed.vertices, oracle.pose_vertices(f.T_gt, f.model_points, dtype=np.float64))
    assert np.array_equal(posed.triangles, f.tris) and np.abs(posed.vertices - compat.transform_object(model, f.T_gt).vertices).max() < 1e-9
    cloud = proj.project(f.T_gt, hm, 0.6, into=c2d)
    direct = proj.project(f.T_gt, hm, 0.6)
    assert np.array_equal(cloud.points, oracle.transform(c2d, direct.points)) and np.array_equal(cloud.colors, direct.colors)


def test_align_to_surface_matches_oracle_nn(ctx, oracle):
    """align_to_surface (defect_projection.py:413-460): nearest model point by the exact NN pass,
    offset along its normal; indices equal the oracle's KD-tree search."""
    from pedp_hip import compat, synth

    f = synth.Frame("parity")
    rng = np.random.default_rng(2)
    pick = rng.integers(0, len(f.model_points), 700)
    defects = np.hstack([f.model_points[pick] + rng.normal(0, 1.5, (700, 3)), rng.uniform(0, 1, (700, 1))])
    model = compat.PointCloud(f.model_points, normals=f.normals)
    off, aligned = compat.align_to_surface(defects, model, offset=0.25)
    idx, _ = oracle.nn(defects[:, :3], f.model_points, kdtree=True)
    assert np.array_equal(aligned, f.model_points[idx])
    assert np.array_equal(off, f.model_points[idx] + f.normals[idx] * 0.25)
    e_off, e_al = compat.align_to_surface(np.zeros((0, 4)), model)
    assert e_off.size == 0 and e_al.size == 0
    # a model without normals gets them estimated first, in place, with radius 0.1 / max_nn 30 (:428-433) -- on this
    # model (points ~1 mm apart) every neighbourhood holds the point alone, so Open3D's rule gives (0, 0, 1)
    bare = compat.PointCloud(f.model_points)
    off2, al2 = compat.align_to_surface(defects, bare, offset=0.25)
    ref_n = oracle.estimate_normals(f.model_points, 0.1, 30)
    assert bare.has_normals() and np.array_equal(bare.normals, ref_n)
    assert np.array_equal(al2, aligned) and np.array_equal(off2, f.model_points[idx] + ref_n[idx] * 0.25)
    # ... and with a radius that reaches the neighbours the estimated normals are real ones
    dense = compat.PointCloud(f.model_points * 0.02)                        # the model in a unit where 0.1 spans neighbours
    compat.align_to_surface(defects * 0.02, dense, offset=0.01)
    ref_d = oracle.estimate_normals(f.model_points * 0.02, 0.1, 30)
    assert np.abs(dense.normals - ref_d).max() < 1e-9 and (np.abs(dense.normals[:, 2]) < 0.999).mean() > 0.5   # real normals, not the (0, 0, 1) default


def test_project_heatmap_device_memory(ctx, oracle):
    """PEDP_DEVICE mode: heat map and outputs are device pointers (torch tensors); only the two
    counts come back to the host."""
    import ctypes as C
    import torch
    from pedp_hip import _lib, synth

    f = synth.Frame("parity")
    hm = _heatmap(f.height, f.width, 17)
    ref = oracle.project_heatmap(f.verts_posed, f.tris, hm, f.K, 0.4)
    mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
    d_hm = torch.from_numpy(hm).cuda()
    cap = hm.size
    d_pts = torch.empty((cap, 3), dtype=torch.float64, device="cuda")
    d_int = torch.empty(cap, dtype=torch.float64, device="cuda")
    d_pix = torch.empty((cap, 2), dtype=torch.int32, device="cuda")
    d_prim = torch.empty(cap, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    cam = _lib.Pinhole(f.K[0, 0], f.K[1, 1], f.K[0, 2], f.K[1, 2], f.width, f.height)
    org = np.zeros(3)
    nr, nh = C.c_int64(), C.c_int64()
    rc = _lib.load().pedp_project_heatmap(ctx._h, mesh._h, C.byref(cam), C.c_void_p(d_hm.data_ptr()), 0.4, _lib._ptr(org),
                                          _lib.DEVICE, cap, C.c_void_p(d_pts.data_ptr()), C.c_void_p(d_int.data_ptr()),
                                          C.c_void_p(d_pix.data_ptr()), C.c_void_p(d_prim.data_ptr()), C.byref(nr), C.byref(nh))
    assert rc == 0, _lib.load().pedp_last_error()
    ctx.synchronize()
    m = nh.value
    assert nr.value == ref["n_rays"] and m == len(ref["points"])
    assert np.array_equal(d_pix[:m].cpu().numpy(), ref["pixels"])
    assert np.array_equal(d_prim[:m].cpu().numpy().view(np.uint32), ref["primitive_ids"])
    assert np.array_equal(d_int[:m].cpu().numpy(), ref["intensities"])
    assert np.abs(d_pts[:m].cpu().numpy() - ref["points"]).max() < POINT_TOL


def test_new_entry_points_reject_bad_arguments(ctx):
    """Status codes and messages of the entry points added beside the hot path."""
    import ctypes as C
    from pedp_hip import _lib, synth

    lib = _lib.load()
    f = synth.Frame("tiny")
    mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
    err = lambda: lib.pedp_last_error().decode()  # noqa: E731
    # set_pose on a fixed mesh, bad mem flag, negative capacity
    assert lib.pedp_mesh_set_pose(mesh._h, None) == -1 and "posable" in err()
    cam = _lib.Pinhole(100.0, 100.0, 10.0, 10.0, 20, 20)
    hm = np.ones((20, 20)); org = np.zeros(3); nr, nh = C.c_int64(), C.c_int64()
    args = lambda mem, cap: (ctx._h, mesh._h, C.byref(cam), _lib._ptr(hm), 0.5, _lib._ptr(org), mem, cap, None, None, None, None,  # noqa: E731
                             C.byref(nr), C.byref(nh))
    assert lib.pedp_project_heatmap(*args(7, 10)) == -1 and "mem flag" in err()
    assert lib.pedp_project_heatmap(*args(_lib.HOST, -1)) == -1 and "capacity" in err()
    cam0 = _lib.Pinhole(0.0, 100.0, 10.0, 10.0, 20, 20)
    assert lib.pedp_project_heatmap(ctx._h, mesh._h, C.byref(cam0), _lib._ptr(hm), 0.5, _lib._ptr(org), _lib.HOST, 400, None,
                                    None, None, None, C.byref(nr), C.byref(nh)) == -1 and "focal" in err()
    # depth filters: radius, null image, bad mem
    d = np.ones((4, 4), np.float32); out = np.empty_like(d)
    assert lib.pedp_erode_depth(ctx._h, _lib._ptr(d), 4, 4, 65, 0.001, 0.8, 100.0, _lib.HOST, _lib._ptr(out)) == -1 and "radius" in err()
    assert lib.pedp_erode_depth(ctx._h, None, 4, 4, 2, 0.001, 0.8, 100.0, _lib.HOST, _lib._ptr(out)) == -1 and "null" in err()
    assert lib.pedp_bilateral_filter_depth(ctx._h, _lib._ptr(d), 4, 4, 2, 100.0, 2.0, 1e5, 3, _lib._ptr(out)) == -1
    assert lib.pedp_depth2xyzmap(ctx._h, _lib._ptr(d), 4, 4, None, _lib.HOST, _lib._ptr(out)) == -1 and "intrinsics" in err()
    # cloud operations
    p = np.zeros((5, 3)); lab = np.empty(5, np.int32); avg = np.empty(5)
    assert lib.pedp_cluster_dbscan(ctx._h, _lib._ptr(p), 5, 0.0, 3, _lib._ptr(lab)) == -1 and "eps" in err()
    assert lib.pedp_knn_mean_distance(ctx._h, _lib._ptr(p), 5, 0, _lib._ptr(avg)) == -1
    assert lib.pedp_estimate_normals(ctx._h, _lib._ptr(p), 5, -1.0, 5, None, _lib._ptr(p)) == -1 and "radius" in err()
    assert lib.pedp_estimate_normals(ctx._h, _lib._ptr(p), 5, 1.0, 500, None, _lib._ptr(p)) == -1 and "max_nn" in err()
    bad = p.copy(); bad[2, 1] = np.inf
    assert lib.pedp_cluster_dbscan(ctx._h, _lib._ptr(bad), 5, 1.0, 3, _lib._ptr(lab)) == -1 and "non-finite" in err()
