"""Latency of the reference's whole per-frame chain on a raw rendered frame (run.py:95-119):
preprocess_source (GPU cloud operations) -> z search -> randomised ICP restarts -> fused defect
projection.  python tools/frame_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth
from pedp_hip.compat import PinholeCameraIntrinsic, PointCloud, TriangleMesh, preprocess_source, preprocess_target, refine_pose_with_icp
from pedp_hip.ray_projection import FrameProjector

ctx = _lib.default_context()
f = synth.Frame("bench_100k")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
scene = f.scene(mesh.cast_rays(f.rays6, want_uv=False)["t_hit"])
src, tgt = PointCloud(scene), PointCloud(f.model_points, normals=f.normals)
params = {"preprocess_target": {"max_pcd": 100000, "keep_normals": True},
          "preprocess_source": {"down_sample": 2, "plane_removal": {"distance_threshold": 2.0, "num_iterations": 1000}},
          "box": False, "mesh": False,
          "refine_registration": {"distance_threshold": 6.0}, "run_icp": {"fitness_threshold": 0.97, "rmse_threshold": 0.8}}
proj = FrameProjector(TriangleMesh(f.model_points, f.tris), PinholeCameraIntrinsic(f.width, f.height, intrinsic_matrix=f.K))
heat = np.zeros((f.height, f.width)); heat[200:380, 220:420] = 1.0
for rep in range(3):
    init = synth.start_pose(); init[2, 3] += 5.0
    np.random.seed(rep)
    t0 = time.perf_counter()
    sp, _, _ = preprocess_source(src, None, params, i=0)
    t1 = time.perf_counter()
    moved, best, z, _ = refine_pose_with_icp(src, tgt, None, init, params)
    t2 = time.perf_counter()
    cloud = proj.project(np.linalg.inv(best.transformation), heat, 0.75)
    t3 = time.perf_counter()
    err = np.abs(np.linalg.inv(best.transformation) - f.T_gt).max()
    print(f"preprocess_source {1e3*(t1-t0):.1f} ms ({len(sp.points)} pts) | refine_pose_with_icp (incl. its own preprocess) "
          f"{1e3*(t2-t1):.1f} ms fitness {best.fitness:.3f} rmse {best.inlier_rmse:.3f} |pose - gt| {err:.3f} | "
          f"projection {1e3*(t3-t2):.2f} ms ({0 if cloud is None else len(cloud.points)} hits)", flush=True)
