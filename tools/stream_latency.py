"""Stage times of BASELINE config 5's geometry chain on the bench frame (the chain of
tests/test_stream_gpu.py): python tools/stream_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pedp_hip import _lib, compat, synth
from pedp_hip.compat import PinholeCameraIntrinsic, PointCloud, TriangleMesh
from pedp_hip.ray_projection import FrameProjector

ctx = _lib.default_context()
f = synth.Frame("bench_100k")
m = _lib.Mesh(ctx, f.verts_posed, f.tris)
t_hit = m.cast_rays(f.rays6, want_uv=False)["t_hit"]
rng = np.random.default_rng(0)
z_mm = np.where(np.isfinite(t_hit), t_hit * f.dirs[:, 2], 600.0) + rng.normal(0.0, 0.5, t_hit.shape)
depth_m = (z_mm / 1000.0).reshape(f.height, f.width).astype(np.float32)
K32 = f.K.astype(np.float32)
model = PointCloud(f.model_points, normals=f.normals)
mesh = TriangleMesh(f.model_points, f.tris)
intr = PinholeCameraIntrinsic(f.width, f.height, intrinsic_matrix=f.K)
c2d = np.eye(4); c2d[:3, 3] = (2.0, -1.0, 0.5)
heat = np.zeros((f.height, f.width)); heat[200:380, 220:420] = np.linspace(0.76, 1.0, 200)[None, :]
params = {"preprocess_target": {"max_pcd": 100000, "keep_normals": True},
          "preprocess_source": {"down_sample": 2, "plane_removal": {"distance_threshold": 2.0, "num_iterations": 500}},
          "box": False, "mesh": False,
          "refine_registration": {"distance_threshold": 6.0}, "run_icp": {"fitness_threshold": 0.97, "rmse_threshold": 0.8}}
proj = FrameProjector(mesh, intr, c2d)
Kt = torch.as_tensor(K32, device="cuda")[None]
for rep in range(4):
    T = [time.perf_counter()]
    def lap(): torch.cuda.synchronize(); T.append(time.perf_counter())
    d = torch.from_numpy(depth_m).cuda(); lap()
    one = ctx if "--one-ctx" in sys.argv else None
    d = compat.bilateral_filter_depth(compat.erode_depth(d, radius=2, ctx=one), radius=2, ctx=one)
    if one is not None: one.synchronize()
    lap()
    xyz = compat.depth2xyzmap_batch(d[None], Kt, zfar=np.inf, ctx=one)[0]
    if one is not None: one.synchronize()
    lap()
    if "--device-scene" in sys.argv:     # the scene stays on the GPU: the voxel grid is built from the device array
        pts = xyz[xyz[..., 2] >= 0.001].double() * 1000.0
        torch.cuda.synchronize()
    elif "--pinned" in sys.argv:
        dev_pts = xyz[xyz[..., 2] >= 0.001].double() * 1000.0
        if rep == 0: host_pts = torch.empty((f.width * f.height, 3), dtype=torch.float64, pin_memory=True)
        host_pts[: len(dev_pts)].copy_(dev_pts)
        pts = host_pts[: len(dev_pts)].numpy()
    else:
        pts = (xyz[xyz[..., 2] >= 0.001].double() * 1000.0).cpu().numpy()
    lap()
    source = PointCloud(pts); lap()
    if "--stages" in sys.argv:
        from pedp_hip import icp_refine as R
        t = [time.perf_counter()]
        pp = params["preprocess_source"]
        down = source.voxel_down_sample(voxel_size=2); t.append(time.perf_counter())
        plane, inl = R.perform_plane_segmentation(down, pp["plane_removal"]); t.append(time.perf_counter())
        R.estimate_normals(down, pp); t.append(time.perf_counter())
        avg = R.compute_average_normal(down); t.append(time.perf_counter())
        rest = R.remove_plane(down, inl); t.append(time.perf_counter())
        big = R.filter_largest_cluster(rest); t.append(time.perf_counter())
        clean = R.remove_statistical_outliers(big, nb_neighbors=75, std_ratio=0.01); t.append(time.perf_counter())
        R.estimate_normals(clean, pp); t.append(time.perf_counter())
        print("   stages ms:", np.round(np.diff(t) * 1e3, 2), len(down.points), len(rest.points), len(big.points), len(clean.points))
    if "--profile" in sys.argv and rep == 2:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        sp, _, _ = compat.preprocess_source(source, None, params, i=0)
        pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
        lap()
    else:
        sp, _, _ = compat.preprocess_source(source, None, params, i=0); lap()
    init = synth.start_pose(); init[2, 3] += 5.0
    np.random.seed(0)
    z = compat.predict_z_axis_adjustment(sp, model, init, params); lap()
    init[2, 3] += z[0]
    start = compat.RegistrationResult(init); start.fitness, start.inlier_rmse = z[1], z[2]
    best = compat.improve_result(sp, model, start, params); lap()
    mis = np.linalg.inv(best.transformation)
    mesh_copy = compat.transform_object(mesh, mis); lap()
    cloud = proj.project(mis, heat, 0.75); lap()
    cloud.transform(c2d); lap()
    names = ["upload depth", "erode+bilateral", "xyzmap", "valid points -> host mm", "PointCloud()", "preprocess_source",
             "z search", "improve_result", "transform_object(mesh)", "project", "cloud.transform"]
    dt = np.diff(T) * 1e3
    print(f"rep {rep}: total {dt.sum():.1f} ms | " + " | ".join(f"{n} {v:.2f}" for n, v in zip(names, dt)), flush=True)
