"""One 20-iteration registration of the bench frame: pedp_icp (direct launches) against pedp_icp_batched with one pose
(a captured graph per stretch of passes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth
from pedp_hip import registration as reg
from pedp_hip.compat import PointCloud

f = synth.Frame("bench_100k")
ctx = _lib.default_context()
m = _lib.Mesh(ctx, f.verts_posed, f.tris)
scene = f.scene(m.cast_rays(f.rays6, want_uv=False)["t_hit"])
src, tgt = reg.upload(PointCloud(scene)), reg.upload(PointCloud(f.model_points, normals=f.normals))
T0 = f.icp_init()
plane = reg.TransformationEstimationPointToPlane()
crit = reg.ICPConvergenceCriteria(relative_fitness=-1.0, relative_rmse=-1.0, max_iteration=20)
for name, fn in (("pedp_icp", lambda: reg.registration_icp(src, tgt, 10.0, T0, plane, crit)),
                 ("pedp_icp_batched x1", lambda: reg.registration_icp_batch(src, tgt, [10.0], [T0], plane, crit)[0])):
    for rep in range(5):
        r = fn()
    t0 = time.perf_counter()
    for rep in range(30):
        r = fn()
    dt = (time.perf_counter() - t0) / 30
    print(f"{name:22s} {1e3 * dt:.3f} ms  fitness {r.fitness:.6f} iterations {r.iterations}", flush=True)
