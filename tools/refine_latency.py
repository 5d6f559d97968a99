"""Latency of the reference-sized refinement flow (run.py:99): ~8k-point scene view, 5k-point
model, z search + 50 randomised ICP restarts, on the GPU through pedp_hip.compat."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth
from pedp_hip.compat import PointCloud, improve_result, predict_z_axis_adjustment, registration_icp, \
    TransformationEstimationPointToPlane
f = synth.Frame("parity")
ctx = _lib.default_context()
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
depth = mesh.cast_rays(f.rays6, want_uv=False)["t_hit"]
hit = np.isfinite(depth)
verts, tris, normals = synth.bumpy_torus(100, 50)            # 5,000-point model
big = synth.Frame("bench_100k")
m2 = _lib.Mesh(ctx, big.verts_posed, big.tris)
d2 = m2.cast_rays(big.rays6, want_uv=False)["t_hit"]
h2 = np.isfinite(d2)
scene = synth.scene_from_depth(d2[h2], big.dirs[h2], noise_sigma=0.3)[::4]   # ~9k points
src, tgt = PointCloud(scene), PointCloud(verts.astype(np.float64), normals=normals)
print("scene", len(scene), "model", len(verts))
param = {"refine_registration": {"distance_threshold": 6.0}, "run_icp": {"fitness_threshold": 2.0, "rmse_threshold": 0.0}}
T0 = synth.start_pose()
for rep in range(3):
    t0 = time.perf_counter()
    r = registration_icp(src, tgt, 6.0, np.linalg.inv(T0), TransformationEstimationPointToPlane())
    t1 = time.perf_counter()
    print(f"single registration_icp (host clouds, {r.iterations} its): {1e3*(t1-t0):.2f} ms  fitness {r.fitness:.3f}")
for rep in range(3):      # the first call of each kind captures its graphs and sizes the workspace
    np.random.seed(0)
    t0 = time.perf_counter(); z = predict_z_axis_adjustment(src, tgt, T0.copy(), param); t1 = time.perf_counter()
    print(f"predict_z_axis_adjustment: {1e3*(t1-t0):.1f} ms -> {z}")
    t0 = time.perf_counter(); res = improve_result(src, tgt, T0, param); t1 = time.perf_counter()
    print(f"improve_result (50 restarts): {1e3*(t1-t0):.1f} ms  fitness {res.fitness:.4f} rmse {res.inlier_rmse:.4f}")
# the whole refine_pose_with_icp of a frame (run.py:99): raw scene with its back plane, preprocess_source included
from pedp_hip.compat import refine_pose_with_icp
raw = big.scene(d2)
params = {"preprocess_target": {"max_pcd": 100000, "keep_normals": True},
          "preprocess_source": {"down_sample": 2, "plane_removal": {"distance_threshold": 2.0, "num_iterations": 500}},
          "box": False, "mesh": False,
          "refine_registration": {"distance_threshold": 6.0}, "run_icp": {"fitness_threshold": 0.97, "rmse_threshold": 0.8}}
model = PointCloud(big.model_points, normals=big.normals)
for rep in range(4):
    init = synth.start_pose(); init[2, 3] += 5.0
    np.random.seed(0)
    t0 = time.perf_counter()
    moved, best, z, _ = refine_pose_with_icp(PointCloud(raw), model, None, init, params)
    t1 = time.perf_counter()
    print(f"refine_pose_with_icp ({len(raw)} scene points x {len(big.model_points)} model points): {1e3*(t1-t0):.1f} ms  "
          f"fitness {best.fitness:.4f} rmse {best.inlier_rmse:.3f}")
