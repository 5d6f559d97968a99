"""GPU tests of the drop-in surface (pedp_hip.compat): the reference's function names run on
the HIP library and reproduce the oracle's restatement of the same control flow."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _frame():
    from pedp_hip import synth

    return synth.Frame("parity")


def test_intersect_rays_with_mesh_points_and_details(oracle):
    from pedp_hip.compat import TriangleMesh, intersect_rays_with_mesh

    f = _frame()
    mesh = TriangleMesh(f.verts_posed.astype(np.float64), f.tris)
    intens = np.linspace(0.0, 1.0, f.n_rays)
    details = {}
    pts, it = intersect_rays_with_mesh(mesh, f.dirs, np.array([0, 0, 0]), intens, details=details)
    ref = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)
    valid = np.isfinite(ref["t_hit"])
    assert np.array_equal(details["primitive_ids"], ref["primitive_ids"])          # bit-exact triangle ids
    assert np.array_equal(details["valid"], valid) and np.array_equal(it, intens[valid])
    expect = f.dirs[valid] * ref["t_hit"][valid, None].astype(np.float64)          # float64 from float32 t, :261-263
    assert np.abs(pts - expect).max() <= 1e-5 and pts.dtype == np.float64


def test_ray_tracing_end_to_end(oracle, tmp_path):
    from pedp_hip.compat import PinholeCameraIntrinsic, PointCloud, LineSet, TriangleMesh, ray_tracing

    f = _frame()
    c2d = np.eye(4)
    c2d[:3, :3] = [[0.9998, -0.02, 0.0], [0.02, 0.9998, 0.0], [0, 0, 1]]
    c2d[:3, 3] = [-32.0, 1.5, 3.0]
    (tmp_path / "configs").mkdir()
    (tmp_path / "configs" / "camera_extrinsics.json").write_text(json.dumps({
        "color_to_depth": {"rotation_matrix": c2d[:3, :3].tolist(), "translation_vector": [c2d[:3, 3].tolist()]},
        "depth_to_color": {"rotation_matrix": c2d[:3, :3].T.tolist(), "translation_vector": [(-c2d[:3, :3].T @ c2d[:3, 3]).tolist()]}}))
    K = PinholeCameraIntrinsic(f.width, f.height, f.f, f.f, f.cx, f.cy)
    # mesh handed over in the depth-camera frame, as run.py:109 does
    mesh_depth = TriangleMesh(f.verts_posed.astype(np.float64) @ c2d[:3, :3].T + c2d[:3, 3], f.tris)
    yy, xx = np.mgrid[0:f.height, 0:f.width]
    heat = np.exp(-(((xx - 80) / 30.0) ** 2 + ((yy - 70) / 25.0) ** 2))
    pcd, moved = ray_tracing(str(tmp_path), mesh_depth, heat, K, heatmap_threshold=0.75)
    assert isinstance(pcd, PointCloud) and moved is not mesh_depth
    # the same thing by hand through the oracle
    vcol = (np.asarray(mesh_depth.vertices) @ np.linalg.inv(c2d)[:3, :3].T + np.linalg.inv(c2d)[:3, 3]).astype(np.float32)
    mask = heat > 0.75
    dirs = f.dirs.reshape(f.height, f.width, 3)[mask]
    rays6 = np.hstack([np.zeros_like(dirs), dirs]).astype(np.float32)
    ref = oracle.raycast(vcol, f.tris, rays6)
    ok = np.isfinite(ref["t_hit"])
    assert ok.sum() > 100 and len(pcd.points) == ok.sum()
    assert np.abs(pcd.points - dirs[ok] * ref["t_hit"][ok, None].astype(np.float64)).max() <= 1e-5
    assert pcd.colors.shape == (ok.sum(), 3) and pcd.colors.min() >= 0 and pcd.colors.max() <= 1
    # nothing hot enough to hit the object -> red debug rays
    heat2 = np.zeros_like(heat)
    heat2[0, 0] = 1.0
    dbg, _ = ray_tracing(str(tmp_path), mesh_depth, heat2, K, heatmap_threshold=0.75)
    assert isinstance(dbg, LineSet) and len(dbg.lines) == 1


def test_registration_icp_open3d_style_call(oracle):
    from pedp_hip.compat import (ICPConvergenceCriteria, PointCloud, TransformationEstimationPointToPlane,
                                 registration_icp)

    g = np.load(os.path.join(GOLD, "g3g4_icp_traces.npz"))
    src, tgt = PointCloud(g["scene_noisy"]), PointCloud(g["model"], normals=g["normals"])
    res = registration_icp(src, tgt, 10.0, g["init"], TransformationEstimationPointToPlane(),
                           ICPConvergenceCriteria(max_iteration=1), want_correspondences=True)
    ref = oracle.icp(g["scene_noisy"], g["model"], g["normals"], 10.0, g["init"], max_iter=1)
    assert res.fitness == ref["fitness"] and np.abs(res.transformation - ref["T"]).max() < 1e-5
    keep = ref["corr"] >= 0
    assert np.array_equal(res.correspondence_set[:, 0], np.nonzero(keep)[0])
    assert np.array_equal(res.correspondence_set[:, 1], ref["corr"][keep])
    with pytest.raises(RuntimeError, match="normals"):
        registration_icp(src, PointCloud(g["model"]), 10.0, g["init"], TransformationEstimationPointToPlane())


def test_improve_result_and_z_search_match_oracle_flow(oracle):
    from pedp_hip import synth
    from pedp_hip.compat import PointCloud, improve_result, predict_z_axis_adjustment

    g = np.load(os.path.join(GOLD, "g3g4_icp_traces.npz"))
    src, tgt = PointCloud(g["scene_clean"]), PointCloud(g["model"], normals=g["normals"])
    param = {"refine_registration": {"distance_threshold": 8.0}, "run_icp": {"fitness_threshold": 0.999, "rmse_threshold": 0.05}}
    np.random.seed(0)
    res = improve_result(src, tgt, synth.start_pose(), param)
    gold = np.load(os.path.join(GOLD, "g6_improve_result.npz"))
    assert res.fitness == float(gold["best_fitness"])
    assert np.abs(res.transformation - gold["best_T"]).max() < 1e-5 and abs(res.inlier_rmse - float(gold["best_rmse"])) < 1e-9
    assert np.random.uniform() == float(gold["rng_after"])        # same number of restarts, same RNG draws
    shifted = synth.start_pose()
    shifted[2, 3] += 12.0
    got = predict_z_axis_adjustment(src, tgt, shifted, param)
    ref = oracle.predict_z_axis_adjustment(g["scene_clean"], g["model"], g["normals"], shifted, param)
    assert got[0] == ref[0] and got[1] == ref[1] and abs(got[2] - ref[2]) < 1e-9


def test_refine_pose_with_icp_full_flow(oracle):
    from pedp_hip import synth
    from pedp_hip.compat import PointCloud, refine_pose_with_icp

    f = _frame()
    depth = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)["t_hit"]
    hit = np.isfinite(depth)
    scene = synth.scene_from_depth(depth[hit], f.dirs[hit], noise_sigma=0.2)
    src, tgt = PointCloud(scene), PointCloud(f.model_points, normals=f.normals)
    params = {"preprocess_target": {"max_pcd": 100000, "keep_normals": True}, "refine_registration": {"distance_threshold": 6.0},
              "run_icp": {"fitness_threshold": 0.98, "rmse_threshold": 1.0}}
    init = synth.start_pose()
    init[2, 3] += 4.0
    init_ref = init.copy()
    np.random.seed(7)
    moved, best, z, _ = refine_pose_with_icp(src, tgt, None, init, params)
    # oracle flow, same seed
    np.random.seed(7)
    z_ref, fit, rmse = oracle.predict_z_axis_adjustment(scene, f.model_points, f.normals, init_ref, params)
    init_ref[2, 3] += z_ref
    start = oracle.Result(init_ref, fit, rmse)
    best_ref = oracle.improve_result(scene, f.model_points, f.normals, start, params)
    assert z == z_ref and best.fitness == best_ref.fitness
    assert np.abs(best.transformation - best_ref.transformation).max() < 1e-5
    assert np.array_equal(init, init_ref)                       # caller's matrix mutated the same way
    err = np.abs(np.linalg.inv(best.transformation) - f.T_gt).max()
    assert err < 0.5                                            # and it actually refines towards the truth (mm)


def test_refine_pose_with_icp_from_raw_frame(oracle):
    """The reference's whole per-frame chain on a raw rendered frame (object + back plane): the
    preprocess_source section switches on voxel grid, plane removal, clustering and the outlier
    filter on the GPU, then z search and randomised ICP restarts; the refined pose lands on the
    ground truth."""
    from pedp_hip import synth
    from pedp_hip.compat import PointCloud, refine_pose_with_icp

    f = synth.Frame("parity")
    depth = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)["t_hit"]
    scene = f.scene(depth)                                      # 0.5 mm noise, misses on the z = 600 plane
    src, tgt = PointCloud(scene), PointCloud(f.model_points, normals=f.normals)
    params = {"preprocess_target": {"max_pcd": 100000, "keep_normals": True},
              "preprocess_source": {"down_sample": 2, "plane_removal": {"distance_threshold": 2.0, "num_iterations": 300}},
              "box": False, "mesh": False,
              "refine_registration": {"distance_threshold": 6.0}, "run_icp": {"fitness_threshold": 0.97, "rmse_threshold": 3.5}}
    init = synth.start_pose()
    init[2, 3] += 5.0
    np.random.seed(3)
    moved, best, z, tgt_proc = refine_pose_with_icp(src, tgt, PointCloud(scene[::40]), init, params)
    # the 2,000-vertex model is sampled every ~5 mm, so the point-to-point rmse at the true pose is ~2.8 mm
    assert best.fitness > 0.97 and best.inlier_rmse < 3.5
    assert np.abs(np.linalg.inv(best.transformation) - f.T_gt).max() < 1.0
    assert len(moved.points) == len(f.model_points)


def test_a_holder_keeps_its_device_copy_between_calls():
    """reg.upload: the model cloud of a camera loop is uploaded, ordered and packed once -- the same handle comes back
    exactly as long as the holder's points and normals are what the copy was made from: the holder owns immutable arrays
    (an in-place edit raises, the caller's source array is copied like Vector3dVector does), and every setter and
    transform bumps its version (the sampled fingerprint missed edits between the sampled rows)."""
    from pedp_hip import registration as reg, synth
    from pedp_hip.compat import PointCloud

    f = synth.Frame("parity")
    mine = f.model_points.copy()
    cloud = PointCloud(mine, normals=f.normals)
    a = reg.upload(cloud)
    assert reg.upload(cloud) is a and reg.upload(cloud) is a          # nothing changed: the same handle
    with pytest.raises(ValueError):
        np.asarray(cloud.points)[3] += 1.0                               # in place: refused, never silently stale
    with pytest.raises(ValueError):
        cloud.normals[5:9] *= -1.0
    mine[3] += 1.0                                                       # the caller's own array is not the holder's
    assert reg.upload(cloud) is a and np.array_equal(cloud.points[3], f.model_points[3])
    edited = np.array(cloud.points)
    edited[3] += 1.0                                                     # one row, none of the old fingerprint's samples
    cloud.points = edited
    b = reg.upload(cloud)
    assert b is not a and reg.upload(cloud) is b
    flipped = np.array(cloud.normals)
    flipped[::7] *= -1.0
    cloud.normals = flipped
    c = reg.upload(cloud)
    assert c is not b
    cloud.transform(np.eye(4))
    d = reg.upload(cloud)
    assert d is not c
    moved = PointCloud.moved_copy(cloud, np.eye(4))                      # aliases the source's arrays until read: immutable ones
    cloud.points = np.array(cloud.points) + 1.0
    assert np.array_equal(moved.points, edited)
    reg.forget_device_copy(cloud)
    assert reg.upload(cloud) is not d
    assert reg.upload(a) is a                                            # a handle passes through
    borrowed = PointCloud.borrowed(mine)                                 # memory the caller keeps rewriting: never kept
    assert reg.upload(borrowed) is not reg.upload(borrowed)


def test_dist_hip_backend_single_rank(oracle):
    """pedp_hip.dist with the product backend on one GPU (no process group): same answers as
    the plain calls; exercises HipBackend (torch stream hand-over, device packet view)."""
    torch = pytest.importorskip("torch")
    from pedp_hip import dist as pdist
    from pedp_hip import synth

    f = _frame()
    be = pdist.HipBackend(0)
    t, ids = pdist.sharded_cast_rays(be, f.verts_posed, f.tris, f.rays6)
    ref = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)
    assert np.array_equal(ids, ref["primitive_ids"]) and np.array_equal(t.view(np.uint32), ref["t_hit"].view(np.uint32))
    scene = f.scene(ref["t_hit"])
    res = pdist.sharded_registration_icp(be, scene, f.model_points, f.normals, 10.0, f.icp_init(), max_iteration=5,
                                         rel_fitness=-1, rel_rmse=-1)
    ro = oracle.icp(scene, f.model_points, f.normals, 10.0, f.icp_init(), max_iter=5, rel_fitness=-1, rel_rmse=-1)
    assert res["fitness"] == ro["fitness"] and np.abs(res["T"] - ro["T"]).max() < 1e-5
    # the all-reduce hook path (icp_reduce kernel + host callback) with an identity "collective"
    seen = []

    def fake_allreduce(ptr, n, stream):
        with be.ordered():                      # the library's stream: item() waits for the reduce kernel
            pkt = be.packet_tensor(ptr, n)
            seen.append(float(pkt[28].item()))  # correspondence count of this pass

    from pedp_hip import _lib
    r2 = _lib.icp(be.ctx, be.make_cloud(scene), be.make_cloud(f.model_points, f.normals), 10.0, f.icp_init(),
                  max_iteration=5, relative_fitness=-1, relative_rmse=-1, allreduce=fake_allreduce,
                  n_source_global=len(scene))
    assert len(seen) == 6 and seen[-1] == round(ro["fitness"] * len(scene))
    assert np.abs(r2["T"] - ro["T"]).max() < 1e-5


def test_dist_hip_backend_rccl_single_rank_group(oracle):
    """The real collectives on one GPU: a world-size-1 RCCL group with always_collective=True
    sends the hit records through all_gather_into_tensor and every ICP pass through all_reduce,
    on the stream the library's kernels run on (ordering without host synchronisation)."""
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    from pedp_hip import dist as pdist

    f = _frame()
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1)
        created = True
    try:
        be = pdist.HipBackend(0)
        ref = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)
        for _ in range(3):  # repeated: an unordered collective would read a half-written buffer sooner or later
            t, ids = pdist.sharded_cast_rays(be, f.verts_posed, f.tris, f.rays6, always_collective=True)
            assert np.array_equal(ids, ref["primitive_ids"]) and np.array_equal(t.view(np.uint32), ref["t_hit"].view(np.uint32))
        scene = f.scene(ref["t_hit"])
        ro = oracle.icp(scene, f.model_points, f.normals, 10.0, f.icp_init(), max_iter=8, rel_fitness=-1, rel_rmse=-1)
        for _ in range(3):
            res = pdist.sharded_registration_icp(be, scene, f.model_points, f.normals, 10.0, f.icp_init(), max_iteration=8,
                                                 rel_fitness=-1, rel_rmse=-1, always_collective=True)
            assert res["fitness"] == ro["fitness"] and np.abs(res["T"] - ro["T"]).max() < 1e-5
    finally:
        if created:
            dist.destroy_process_group()


def test_two_ranks_on_one_gpu_product_backend(oracle, tmp_path):
    """Two processes, product backend (HipBackend) on GPU 0, gloo as the transport: the sharded ray
    cast with its all-gather and the sharded ICP with one packet all-reduce per pass, on the
    library's streams.  Both ranks must hold the single-process oracle result."""
    import socket
    import subprocess
    import sys

    from pedp_hip import synth

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "_dist_worker.py"), str(r), "2", str(port),
                               str(tmp_path), "hip"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    f = synth.Frame("tiny")
    rays = f.rays6[: 48 * 40 - 3]
    ref = oracle.raycast(f.verts_posed, f.tris, rays)
    g = np.load(os.path.join(GOLD, "g3g4_icp_traces.npz"))
    ricp = oracle.icp(g["scene_noisy"][:-5], g["model"], g["normals"], 10.0, g["init"], max_iter=6, rel_fitness=-1, rel_rmse=-1)
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(2))
    for r in (r0, r1):
        assert np.array_equal(r["ids"], ref["primitive_ids"]) and np.array_equal(r["t"].view(np.uint32), ref["t_hit"].view(np.uint32))
        assert float(r["fitness"]) == ricp["fitness"] and int(r["iters"]) == 6
        assert np.abs(r["T"] - ricp["T"]).max() < 1e-5
        # the device-resident sharded frame (bench.py's form) gives the same frame and pose
        assert np.array_equal(r["sid"], ref["primitive_ids"]) and np.array_equal(r["st"].view(np.uint32), ref["t_hit"].view(np.uint32))
        assert float(r["sfit"]) == ricp["fitness"] and np.abs(r["sT"] - ricp["T"]).max() < 1e-5
        assert not bool(r["native"])             # two ranks on one GPU: RCCL refuses, torch collectives stay
    assert np.array_equal(r0["T"], r1["T"])      # every rank holds the identical pose
    assert np.array_equal(r0["sT"], r1["sT"])
    # batched-pose sharding: both ranks hold all 5 results, equal to single calls BIT FOR BIT: the scene's
    # spatial order -- and with it the order of every float64 sum -- is a function of the cloud alone
    from pedp_hip import _lib
    ctx = _lib.default_context()
    src, tgt = _lib.Cloud(ctx, g["scene_noisy"]), _lib.Cloud(ctx, g["model"], g["normals"])
    assert np.array_equal(r0["bT"], r1["bT"]) and np.array_equal(r0["bfit"], r1["bfit"])
    for b in range(5):
        one = _lib.icp(ctx, src, tgt, 10.0, r0["binits"][b], max_iteration=4, relative_fitness=-1, relative_rmse=-1)
        assert np.array_equal(r0["bT"][b], one["T"]) and r0["bfit"][b] == one["fitness"]
        assert r0["brmse"][b] == one["inlier_rmse"]


def test_native_rccl_communicator_single_rank(oracle):
    """The library's own RCCL communicator (pedp_comm_create, dlopen'ed librccl) on one GPU: a
    one-rank communicator carries the all-gather of the hit records and the per-pass all-reduce
    of the ICP packet (pedp_icp_params.use_comm) on the library's stream, no Python in between."""
    torch = pytest.importorskip("torch")
    from pedp_hip import _lib
    from pedp_hip import dist as pdist

    f = _frame()
    be = pdist.HipBackend(0)
    assert be.init_comm(force=True) and be.native
    assert _lib.comm_size(be.ctx) == (1, 0)
    ref = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)
    for _ in range(3):
        t, ids = pdist.sharded_cast_rays(be, f.verts_posed, f.tris, f.rays6, always_collective=True)
        assert np.array_equal(ids, ref["primitive_ids"]) and np.array_equal(t.view(np.uint32), ref["t_hit"].view(np.uint32))
    scene = f.scene(ref["t_hit"])
    ro = oracle.icp(scene, f.model_points, f.normals, 10.0, f.icp_init(), max_iter=8, rel_fitness=-1, rel_rmse=-1)
    for _ in range(3):
        res = pdist.sharded_registration_icp(be, scene, f.model_points, f.normals, 10.0, f.icp_init(), max_iteration=8,
                                             rel_fitness=-1, rel_rmse=-1, always_collective=True)
        assert res["fitness"] == ro["fitness"] and np.abs(res["T"] - ro["T"]).max() < 1e-5
    # raw entry points on device buffers
    x = torch.arange(29, dtype=torch.float64, device="cuda:0")
    with be.ordered():
        y = x.clone()
    _lib.comm_allreduce_f64(be.ctx, y.data_ptr(), 29)
    be.ctx.synchronize()
    assert torch.equal(x, y)                         # sum over one rank
    # use_comm without a communicator is an error, not a silent single-rank run
    plain = _lib.default_context()
    with pytest.raises(_lib.PedpError):
        _lib.icp(plain, _lib.Cloud(plain, scene), _lib.Cloud(plain, f.model_points, f.normals), 10.0, f.icp_init(),
                 max_iteration=1, use_comm=True)
    _lib.comm_destroy(be.ctx)


def test_sharded_icp_batched_single_rank(oracle):
    """sharded_icp_batched on one GPU equals pedp_icp_batched (and the gather path keeps the bits)."""
    pytest.importorskip("torch")
    from pedp_hip import _lib, synth
    from pedp_hip import dist as pdist

    f = _frame()
    depth = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)["t_hit"]
    scene = f.scene(depth)
    inits = np.stack([np.linalg.inv(T) for T in synth.batched_start_poses(6)])
    be = pdist.HipBackend(0)
    T, fit, rmse = pdist.sharded_icp_batched(be, scene, f.model_points, f.normals, 10.0, inits, max_iteration=5)
    T2, fit2, rmse2 = _lib.icp_batched(be.ctx, be.make_cloud(scene), be.make_cloud(f.model_points, f.normals), 10.0, inits,
                                       max_iteration=5)
    assert np.array_equal(T, T2) and np.array_equal(fit, fit2) and np.array_equal(rmse, rmse2)
    ro = oracle.icp(scene, f.model_points, f.normals, 10.0, inits[4], max_iter=5, rel_fitness=-1, rel_rmse=-1)
    assert fit[4] == ro["fitness"] and np.abs(T[4] - ro["T"]).max() < 1e-5


def test_icp_with_shuffled_subsampled_target(oracle):
    """preprocess_target subsamples the model with np.random.choice: a target in random order
    (no spatial coherence in index order) must give the same exact correspondences."""
    from pedp_hip import _lib, synth

    f = synth.Frame("parity")
    depth = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)["t_hit"]
    scene = f.scene(depth)
    keep = np.random.default_rng(5).choice(len(f.model_points), 1500, replace=False)
    model, normals = f.model_points[keep], f.normals[keep]
    ctx = _lib.default_context()
    res = _lib.icp(ctx, _lib.Cloud(ctx, scene), _lib.Cloud(ctx, model, normals), 12.0, f.icp_init(), max_iteration=8,
                   relative_fitness=-1, relative_rmse=-1, want_corr=True)
    ref = oracle.icp(scene, model, normals, 12.0, f.icp_init(), max_iter=8, rel_fitness=-1, rel_rmse=-1)
    assert np.array_equal(res["corr"], ref["corr"]) and np.abs(res["T"] - ref["T"]).max() < 1e-5
