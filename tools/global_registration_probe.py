import sys, time, numpy as np, logging
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import os
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
from pedp_hip import synth, _lib, icp_refine, registration as reg
from pedp_hip.compat import PointCloud
cfg = sys.argv[1] if len(sys.argv) > 1 else "bench_100k"
f = synth.Frame(cfg)
ctx = _lib.default_context()
m = _lib.Mesh(ctx, f.verts_posed, f.tris)
depth = m.cast_rays(f.rays6, want_uv=False)["t_hit"]
scene = f.scene(depth)
toward_camera = -scene / np.linalg.norm(scene, axis=1, keepdims=True)
src = PointCloud(scene, normals=toward_camera)          # estimate_normals keeps the side of normals that are there
tgt = PointCloud(f.model_points, normals=f.normals).voxel_down_sample(2.0) if "--voxel-model" in sys.argv else PointCloud(f.model_points, normals=f.normals)
params = {"preprocess_target": {"max_pcd": 100000, "keep_normals": True, "fpfh_radius": 15.0, "fpfh_max_nn": 100},
          "preprocess_source": {"down_sample": 2, "plane_removal": {"distance_threshold": 2.0, "num_iterations": 300},
                                "fpfh_radius": 15.0, "fpfh_max_nn": 100},
          "box": False, "mesh": False,
          "execute_global_registration": {"distance_threshold": 5.0, "correspondence_checkers": [{"value": 0.9}],
                                          "angle_threshold": 0.7, "ransac_criteria": {"iterations": 100000, "confidence": 0.999}},
          "refine_registration": {"distance_threshold": 6.0}, "run_icp": {"fitness_threshold": 0.9, "rmse_threshold": 3.5}}
t0 = time.perf_counter()
tp, tf = icp_refine.preprocess_target(tgt, params)
sp, _, sf = icp_refine.preprocess_source(src, None, params)
from pedp_hip import cloud_ops
corr = cloud_ops.match_features(sf.data.T, tf.data.T)
T_true = np.linalg.inv(f.T_gt)
moved = np.asarray(sp.points) @ T_true[:3, :3].T + T_true[:3, 3]
dd = np.linalg.norm(moved - np.asarray(tp.points)[corr], axis=1)
print("share of feature matches within 5 mm of the truth: %.4f, within 10 mm: %.4f" % ((dd < 5).mean(), (dd < 10).mean()))
print("preprocess", time.perf_counter() - t0, len(sp.points), len(tp.points), "src normals z-only:", float(np.mean(np.abs(np.asarray(sp.normals)[:, 2]) == 1.0)))
reg.set_ransac_seed(5)
for k in range(4):
    t0 = time.perf_counter()
    r = icp_refine.execute_global_registration(sp, tp, sf, tf, params)
    t1 = time.perf_counter()
    ri = icp_refine.refine_registration(sp, tp, r.transformation, params)
    print(k, "ransac fit %.3f rmse %.3f validated %d  %.1f ms | refined fit %.3f rmse %.3f | err %.2f" % (r.fitness, r.inlier_rmse, r.validated_draws, 1e3 * (t1 - t0), ri.fitness, ri.inlier_rmse, np.abs(np.linalg.inv(ri.transformation) - f.T_gt).max()))
