"""Wall time of one pedp_icp_batched call against its GPU time: 8 single-iteration probes (the z search's batch)
on a 6.8 k x 50 k problem."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth
from pedp_hip import registration as reg
from pedp_hip.compat import PointCloud

f = synth.Frame("bench_100k")
ctx = _lib.default_context()
m = _lib.Mesh(ctx, f.verts_posed, f.tris)
d = m.cast_rays(f.rays6, want_uv=False)["t_hit"]
h = np.isfinite(d)
scene = synth.scene_from_depth(d[h], f.dirs[h], noise_sigma=0.3)[::5]
src, tgt = reg.upload(PointCloud(scene)), reg.upload(PointCloud(f.model_points, normals=f.normals))
T0 = np.linalg.inv(synth.start_pose())
plane = reg.TransformationEstimationPointToPlane()
for iters, n in ((1, 7), (1, 1), (30, 8), (30, 1), (0, 32)):
    crit = reg.ICPConvergenceCriteria(max_iteration=iters)
    inits = [T0] * n
    for rep in range(3):
        reg.registration_icp_batch(src, tgt, [6.0] * n, inits, plane, crit)
    t0 = time.perf_counter()
    for rep in range(50):
        reg.registration_icp_batch(src, tgt, [6.0] * n, inits, plane, crit)
    dt = (time.perf_counter() - t0) / 50
    print(f"{n:2d} poses x max_iteration {iters:2d}: {1e6 * dt:7.1f} us per call", flush=True)
