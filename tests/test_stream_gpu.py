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
    torch = pytest.importorskip("torch")
    from pedp_hip import compat, synth, viewer_wire
    from pedp_hip.compat import PinholeCameraIntrinsic, PointCloud, TriangleMesh
    from pedp_hip.ray_projection import FrameProjector

    f = synth.Frame("bench_100k")                                         # 640 x 576, 100k triangles
    t_hit = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)["t_hit"]
    rng = np.random.default_rng(0)
    z_mm = np.where(np.isfinite(t_hit), t_hit * f.dirs[:, 2], 600.0) + rng.normal(0.0, 0.5, t_hit.shape)
    depth_m = (z_mm / 1000.0).reshape(f.height, f.width).astype(np.float32)   # the filters work in metres
    K32 = f.K.astype(np.float32)
    model = PointCloud(f.model_points, normals=f.normals)
    mesh = TriangleMesh(f.model_points, f.tris)
    intr = PinholeCameraIntrinsic(f.width, f.height, intrinsic_matrix=f.K)
    color_to_depth = np.eye(4)
    color_to_depth[:3, 3] = (2.0, -1.0, 0.5)
    depth_to_color = np.linalg.inv(color_to_depth)
    heat = np.zeros((f.height, f.width))
    heat[200:380, 220:420] = np.linspace(0.76, 1.0, 200)[None, :]
    params = {"preprocess_target": {"max_pcd": 100000, "keep_normals": True},
              "preprocess_source": {"down_sample": 2, "plane_removal": {"distance_threshold": 2.0, "num_iterations": 500}},
              "box": False, "mesh": False,
              "refine_registration": {"distance_threshold": 6.0}, "run_icp": {"fitness_threshold": 0.97, "rmse_threshold": 0.8}}
    proj = FrameProjector(mesh, intr, color_to_depth)
    q = queue.Queue()
    viewer_wire.attach_queues(q)
    # the scene cloud comes back to the host (the reference's chain works on host clouds): into a PINNED
    # buffer that lives across frames.  A pageable destination makes the runtime pin and unpin 9 MB per
    # frame, which holds up the next submissions by 20-30 ms (DESIGN s6).
    host_pts = torch.empty((f.width * f.height, 3), dtype=torch.float64, pin_memory=True)

    def frame(seed):
        # ---- depth pre-filters and back-projection, device tensors throughout (estimater.py:255-259)
        d = torch.from_numpy(depth_m).cuda()
        d = compat.erode_depth(d, radius=2, device="cuda")
        d = compat.bilateral_filter_depth(d, radius=2, device="cuda")
        xyz = compat.depth2xyzmap_batch(d[None], torch.as_tensor(K32, device="cuda")[None], zfar=np.inf)[0]
        dev_pts = xyz[xyz[..., 2] >= 0.001].double() * 1000.0                  # scene cloud in mm (run.py works in mm)
        host_pts[: len(dev_pts)].copy_(dev_pts)
        pts = host_pts[: len(dev_pts)].numpy().copy()
        source = PointCloud(pts)
        # ---- run.py:95-99: start pose (depth-camera frame), refinement
        init = synth.start_pose()
        init[2, 3] += 5.0
        np.random.seed(seed)
        _, icp, z, _ = compat.refine_pose_with_icp(source, model, None, init, params)
        # ---- run.py:109-119: posed mesh, projection in the colour-camera frame, back into the depth frame
        model_in_scene = np.linalg.inv(icp.transformation)
        mesh_copy = compat.transform_object(mesh, model_in_scene)
        cloud = proj.project(model_in_scene, heat, 0.75)
        cloud.transform(color_to_depth)
        msg = compat.update_dash_data([cloud], mesh_copy)                      # run.py:131
        return d, xyz, pts, init, icp, z, cloud, mesh_copy, msg

    frame(0)                                                                   # warm-up: buffers, graphs
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        out = frame(0)
    dt = (time.perf_counter() - t0) / n
    d, xyz, pts, init, icp, z, cloud, mesh_copy, msg = out
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
