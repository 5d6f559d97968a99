"""Latency of pedp_icp_batched_ex on a reference-sized problem (~9k scene points, 5k model points):
B registrations with their own radius each, Open3D's default criteria.  python tools/batch_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth

ctx = _lib.default_context()
verts, tris, normals = synth.bumpy_torus(100, 50)
big = synth.Frame("bench_100k")
m2 = _lib.Mesh(ctx, big.verts_posed, big.tris)
d2 = m2.cast_rays(big.rays6, want_uv=False)["t_hit"]
h2 = np.isfinite(d2)
scene = synth.scene_from_depth(d2[h2], big.dirs[h2], noise_sigma=0.3)[::4]
src, tgt = _lib.Cloud(ctx, scene), _lib.Cloud(ctx, verts.astype(np.float64), normals)
T0 = np.linalg.inv(synth.start_pose())
rng = np.random.default_rng(0)
for B in (1, 2, 4, 8, 16):
    inits = np.repeat(T0[None], B, 0).copy()
    inits[:, :3, 3] += rng.normal(0, 0.1, (B, 3))
    radii = 6.0 * rng.uniform(0.8, 1.2, B)
    ts = []
    for rep in range(6):
        t0 = time.perf_counter()
        T, fit, rmse, its = _lib.icp_batched_ex(ctx, src, tgt, radii, inits, max_iteration=30)
        ts.append(time.perf_counter() - t0)
    print(f"B={B:2d}: {1e3*min(ts):7.3f} ms per call, {1e3*min(ts)/B:6.3f} ms per registration; iterations {its.tolist()}", flush=True)
ts = []
for rep in range(6):
    t0 = time.perf_counter()
    r = _lib.icp(ctx, src, tgt, 6.0, T0, max_iteration=30)
    ts.append(time.perf_counter() - t0)
print(f"single pedp_icp: {1e3*min(ts):.3f} ms ({r['iters']} iterations)")
