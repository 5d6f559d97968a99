"""BASELINE config 5 (geometry part): the whole per-frame chain of the reference at the bench
frame size, every stage on the GPU, every stage checked against the oracle's chain:

    depth image -> erode_depth -> bilateral_filter_depth -> depth2xyzmap_batch   (estimater.py:255-259)
    -> scene cloud (mm) -> preprocess_source -> z search + randomised ICP restarts (run.py:95-99)
    -> posed mesh -> ray projection of the heat map (run.py:109-119)
    -> update_dash_data message                                                   (run.py:131)

and then run.py's tracking branch with a defect detection (run.py:132-207): preprocess_source(i > 0) ->
improve_result from the bare 4x4 -> delta_pose -> posed mesh -> ray projection -> relative_transformation
on the earlier hit clouds -> update_dash_data.  Both with the arguments run.py passes: the reader's
background cloud and the root logger at INFO (run.py:99-101, :252-260).

The FoundationPose networks that supply the start pose are out of scope: the start pose is the
ground truth perturbed like the bench's ICP start.  Prints frames/s of the chain."""
import logging
import queue
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config5_frame_chain_at_camera_resolution(oracle, tmp_path):
    pytest.importorskip("torch")
    from pedp_hip import compat, synth, viewer_wire
    from pedp_hip.compat import PointCloud
    from pedp_hip.frame_chain import bench_frame_setup

    f = synth.Frame("bench_100k")                                         # 640 x 576, 100k triangles
    t_hit = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)["t_hit"]
    chain, depth_m, heat, init_pose = bench_frame_setup(f, t_hit)          # the chain bench.py's frame_chain region times
    assert chain.background is not None and len(chain.background.points) == f.width * f.height   # reader.background
    root = logging.getLogger()
    level = root.level
    root.setLevel(logging.INFO)                                             # run.py:252, :260
    try:
        _frame0_and_tracking(oracle, f, chain, depth_m, heat, init_pose, t_hit)
    finally:
        root.setLevel(level)


def _frame0_and_tracking(oracle, f, chain, depth_m, heat, init_pose, t_hit):
    from pedp_hip import cloud_ops, compat, icp_refine, synth, viewer_wire
    from pedp_hip.compat import PointCloud
    from pedp_hip.frame_chain import bench_frame_setup

    K32 = f.K.astype(np.float32)
    params, color_to_depth = chain.params, chain.color_to_depth
    depth_to_color = np.linalg.inv(color_to_depth)
    q = queue.Queue()
    viewer_wire.attach_queues(q)

    one_call = []
    fused = cloud_ops.preprocess_source_fused
    cloud_ops.preprocess_source_fused = lambda *a, **k: (one_call.append(k), fused(*a, **k))[1]
    chain.process(depth_m, init_pose(), heat, seed=0)                          # warm-up: buffers, graphs
    cloud_ops.preprocess_source_fused = fused
    assert len(one_call) == 1 and one_call[0]["report"]      # under run.py's arguments the ONE-CALL preprocess_source ran
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        out = chain.process(depth_m, init_pose(), heat, seed=0)
    dt = (time.perf_counter() - t0) / n
    d, xyz, pts, init, icp, z, cloud, mesh_copy, msg = (out[k] for k in ("depth", "xyz", "points", "init", "icp", "z", "cloud",
                                                                           "mesh", "message"))
    print(f"config 5 geometry chain: {1e3 * dt:.1f} ms per 640x576 frame = {1.0 / dt:.1f} frames/s "
          f"({len(pts)} scene points, {len(cloud.points)} projected hits)")

    # ---- every stage against the oracle's chain
    e_ref = oracle.erode_depth(depth_m, radius=2)
    b_ref = oracle.bilateral_filter_depth(e_ref, radius=2)
    d_host = d.cpu().numpy()
    assert np.abs(d_host - b_ref).max() <= 2e-6 * np.abs(b_ref).max()          # bilateral: two libm exps (DESIGN s2)
    xyz_ref = oracle.depth2xyzmap_batch(d_host[None], K32[None], np.inf)[0]
    assert np.array_equal(xyz.cpu().numpy(), xyz_ref, equal_nan=True)
    assert len(pts) == f.width * f.height == 368640 or len(pts) > 300000
    # refinement: the oracle's flow on the same scene cloud, same seed
    from pedp_hip.compat import preprocess_source
    root = logging.getLogger()
    root.setLevel(logging.WARNING)
    sp, _, _ = preprocess_source(PointCloud(pts), None, params, i=0)         # no background, nobody listening:
    plain = bench_frame_setup(f, t_hit, with_background=False)[0]
    out_plain = plain.process(depth_m, init_pose(), heat, seed=0)           # the same frame, bit for bit
    root.setLevel(logging.INFO)
    assert np.array_equal(out_plain["icp"].transformation, icp.transformation) and out_plain["z"] == z
    assert np.array_equal(out_plain["cloud"].points, cloud.points)
    init_ref = synth.start_pose()
    init_ref[2, 3] += 5.0
    np.random.seed(0)
    z_ref, fit, rmse = oracle.predict_z_axis_adjustment(sp.points, f.model_points, f.normals, init_ref, params)
    init_ref[2, 3] += z_ref
    best_ref = oracle.improve_result(sp.points, f.model_points, f.normals, oracle.Result(init_ref, fit, rmse), params)
    assert z == z_ref and icp.fitness == best_ref.fitness and np.abs(icp.transformation - best_ref.transformation).max() < 1e-5
    assert np.array_equal(init, init_ref)
    model_in_scene = np.linalg.inv(icp.transformation)
    assert np.abs(model_in_scene - f.T_gt).max() < 0.05                        # lands on the ground truth (mm)
    # projection: the oracle on the vertices posed by the same chain
    v32 = oracle.pose_vertices(depth_to_color @ model_in_scene, f.model_points)
    rp = oracle.project_heatmap(v32, f.tris, heat, f.K, 0.75)
    hits_depth_frame = rp["points"] @ color_to_depth[:3, :3].T + color_to_depth[:3, 3]
    assert len(cloud.points) == len(rp["points"]) > 10000 and np.abs(cloud.points - hits_depth_frame).max() < 1e-9
    # the message to the viewer thread (web_vis.py:203-217)
    got = q.get_nowait()
    assert got is not None and set(got) == {"pcds", "vertices", "faces"} and set(got["pcds"][0]) == {"points", "colors"}
    assert np.array_equal(got["pcds"][0]["points"], cloud.points) and got["pcds"][0]["colors"].shape == cloud.points.shape
    assert np.array_equal(got["vertices"], mesh_copy.vertices) and np.array_equal(got["faces"], f.tris)
    assert msg["vertices"] is not None and (got["pcds"][0]["colors"] >= 0).all() and (got["pcds"][0]["colors"] <= 1).all()
    # ---- a tracking frame with a defect detection pending (run.py:132-207), same arguments
    from pedp_hip.compat import preprocess_source as pre
    hits0 = cloud.points.copy()                                                # frame 0's hits, depth-camera frame
    previous = icp.transformation
    depth_1 = (depth_m + np.random.default_rng(11).normal(0.0, 2e-4, depth_m.shape)).astype(np.float32)   # the next image
    nudge = np.eye(4)
    nudge[:3, :3] = oracle.rot_xyz([0.004, -0.006, 0.005])
    nudge[:3, 3] = (0.8, -0.5, 1.0)
    init_1 = nudge @ model_in_scene                                            # the tracker's pose (est.track_one)
    heat_1 = np.zeros_like(heat)
    heat_1[150:330, 260:460] = np.linspace(0.76, 1.0, 200)[None, :]            # a new detection's heat map
    assert params["preprocess_source"]["down_sample"] == 2                     # frame 0 works on a copy (:766)
    one_call = []
    cloud_ops.preprocess_source_fused = lambda *a, **k: (one_call.append(k), fused(*a, **k))[1]
    t0 = time.perf_counter()
    out1 = chain.process_tracking(depth_1, init_1.copy(), heat_1, i=1, seed=3)
    dt1 = time.perf_counter() - t0
    cloud_ops.preprocess_source_fused = fused
    print(f"tracking frame with a detection: {1e3 * dt1:.1f} ms (first call of its kind: graphs and buffers included)")
    assert len(one_call) == 1 and one_call[0]["report"] and not one_call[0]["first_frame"]
    assert params["preprocess_source"]["down_sample"] == 5                     # run.py:154-156 mutates the reader's dict (:202-203)
    sp1 = out1["source_processed"]
    icp_refine._FORCE_STEPS = True
    try:
        steps1, _, _ = pre(PointCloud(out1["points"]), chain.background, params, i=1)   # through the single operations
    finally:
        icp_refine._FORCE_STEPS = False
    assert np.array_equal(sp1.points, steps1.points) and not sp1.has_normals() and 500 < len(sp1.points) < len(sp.points)
    np.random.seed(3)                                                          # improve_result from the bare 4x4 (:564-569)
    ref1 = oracle.improve_result(sp1.points, f.model_points, f.normals, init_1, params)
    rng_after = np.random.get_state()[1].copy()
    cur = out1["icp"].transformation
    assert out1["icp"].fitness == ref1.fitness and abs(out1["icp"].inlier_rmse - ref1.inlier_rmse) < 1e-12
    assert np.abs(cur - ref1.transformation).max() < 1e-5
    np.random.seed(3)
    chain_again = compat.improve_result(sp1, chain.target_processed, init_1, params)
    assert np.array_equal(chain_again.transformation, cur) and np.array_equal(np.random.get_state()[1], rng_after)
    mis1 = np.linalg.inv(cur)
    assert np.abs(mis1 - f.T_gt).max() < 0.1                                   # (a 5-unit grid: 876 scene points)
    assert np.array_equal(out1["delta_pose"], np.linalg.inv(init_1) @ mis1)                        # run.py:176-178
    assert np.array_equal(out1["relative"], mis1 @ previous)                                       # run.py:183-184
    assert np.array_equal(chain.previous_transformation, cur) and len(chain.intersection_pcds) == 2
    moved0 = hits0 @ out1["relative"][:3, :3].T + out1["relative"][:3, 3]                          # run.py:196-197
    assert np.abs(chain.intersection_pcds[0].points - moved0).max() < 1e-9
    v32 = oracle.pose_vertices(depth_to_color @ mis1, f.model_points)
    rp1 = oracle.project_heatmap(v32, f.tris, heat_1, f.K, 0.75)
    hits1 = rp1["points"] @ color_to_depth[:3, :3].T + color_to_depth[:3, 3]
    assert len(out1["cloud"].points) == len(hits1) > 10000 and np.abs(out1["cloud"].points - hits1).max() < 1e-9
    while q.qsize() > 1:
        q.get_nowait()
    got = q.get_nowait()                                                       # run.py:206: both clouds, the new mesh
    assert len(got["pcds"]) == 2 and np.array_equal(got["pcds"][0]["points"], chain.intersection_pcds[0].points)
    assert np.array_equal(got["pcds"][1]["points"], out1["cloud"].points) and np.array_equal(got["vertices"], out1["mesh"].vertices)
    assert np.abs(got["vertices"] - (f.model_points @ mis1[:3, :3].T + mis1[:3, 3])).max() < 1e-9
    # ---- a tracking frame without a detection (run.py:208-210)
    init_2 = nudge @ init_1
    assert np.array_equal(chain.track_only(init_2), np.linalg.inv(init_2 @ out1["delta_pose"]))
    # steady rate of the tracking branch (what bench.py's tracking_frame region times)
    for _ in range(2):
        chain.process_tracking(depth_1, init_1.copy(), heat_1, i=2, seed=3)
    t0 = time.perf_counter()
    for k in range(3):
        chain.process_tracking(depth_1, init_1.copy(), heat_1, i=3 + k, seed=3)
    print(f"tracking frames: {1e3 * (time.perf_counter() - t0) / 3:.1f} ms per 640x576 frame with a detection")
    viewer_wire.attach_queues(None)
    with pytest.raises(RuntimeError):
        compat.update_dash_data([cloud], mesh_copy)
