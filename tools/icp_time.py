"""Median wall time of the bench-size 20-iteration registration (and of a 1-iteration one): python tools/icp_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth
ctx = _lib.Context(0)
f = synth.Frame("bench_100k")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
depth = mesh.cast_rays(f.rays6, want_uv=False)["t_hit"]
src = _lib.Cloud(ctx, f.scene(depth)); tgt = _lib.Cloud(ctx, f.model_points, f.normals)
def med(n_it, reps=40):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), max_iteration=n_it, relative_fitness=-1, relative_rmse=-1)
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts[5:])), r
t20, r = med(20)
t1, _ = med(1)
print(f"{os.path.basename(os.environ.get('PEDP_LIB', 'libpedp_hip.so')):28s} 20 iterations {t20:.3f} ms ({20 / t20:.1f} k its/s)   1 iteration {t1:.3f} ms   fitness {r['fitness']:.12f} rmse {r['inlier_rmse']:.12f}")
