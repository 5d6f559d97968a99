import os, sys, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import numpy as np
from pedp_hip import _lib, synth
from pedp_hip.compat import PointCloud, refine_pose_with_icp
big = synth.Frame("bench_100k")
ctx = _lib.default_context()
m2 = _lib.Mesh(ctx, big.verts_posed, big.tris)
d2 = m2.cast_rays(big.rays6, want_uv=False)["t_hit"]
raw = big.scene(d2)
params = {"preprocess_target": {"max_pcd": 100000, "keep_normals": True},
          "preprocess_source": {"down_sample": 2, "plane_removal": {"distance_threshold": 2.0, "num_iterations": 500}},
          "box": False, "mesh": False,
          "refine_registration": {"distance_threshold": 6.0}, "run_icp": {"fitness_threshold": 0.97, "rmse_threshold": 0.8}}
model = PointCloud(big.model_points, normals=big.normals)
for rep in range(4):
    init = synth.start_pose(); init[2, 3] += 5.0
    np.random.seed(0)
    src = PointCloud(raw)
    if rep == 3:
        pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    refine_pose_with_icp(src, model, None, init, params)
    t1 = time.perf_counter()
    if rep == 3:
        pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(16)
    print("%.1f ms" % (1e3 * (t1 - t0)))
