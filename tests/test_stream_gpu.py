"""BASELINE config 5 (geometry part): the whole per-frame chain of the reference at the bench
frame size, every stage on the GPU, every stage checked against the oracle's chain:

    depth image -> erode_depth -> bilateral_filter_depth -> depth2xyzmap_batch   (estimater.py:255-259)
    -> scene cloud (mm) -> preprocess_source -> z search + randomised ICP restarts (run.py:95-99)
    -> posed mesh -> ray projection of the heat map (run.py:109-119)
    -> update_dash_data message                                                   (run.py:131)

The FoundationPose networks that supply the start pose are out of scope: the start pose is the
ground truth perturbed like the bench's ICP start.  Prints frames/s of the chain."""
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
    K32 = f.K.astype(np.float32)
    params, color_to_depth = chain.params, chain.color_to_depth
    depth_to_color = np.linalg.inv(color_to_depth)
    q = queue.Queue()
    viewer_wire.attach_queues(q)

    chain.process(depth_m, init_pose(), heat, seed=0)                          # warm-up: buffers, graphs
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
    sp, _, _ = preprocess_source(PointCloud(pts), None, params, i=0)
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
    viewer_wire.attach_queues(None)
    with pytest.raises(RuntimeError):
        compat.update_dash_data([cloud], mesh_copy)
