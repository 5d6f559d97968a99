"""Batched registrations at bench size: iterations/s for B start poses sharing the launches (BASELINE config 3 names 256):
python tools/batch_poses.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth
ctx = _lib.Context(0)
f = synth.Frame("bench_100k")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
depth = mesh.cast_rays(f.rays6, want_uv=False)["t_hit"]
src = _lib.Cloud(ctx, f.scene(depth)); tgt = _lib.Cloud(ctx, f.model_points, f.normals)
for B in [int(a) for a in sys.argv[1:]] or [1, 8, 32, 64, 128, 256]:
    inits = np.stack([np.linalg.inv(T) for T in synth.batched_start_poses(B)])
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        out = _lib.icp_batched(ctx, src, tgt, 10.0, inits, max_iteration=20)
        ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    fit = np.array([o["fitness"] for o in out]) if isinstance(out, list) else np.asarray(out[1])
    print(f"B = {B:4d}: {1e3 * t:8.2f} ms = {1e3 * t / B:6.3f} ms per registration, {20 * B / t / 1e3:7.1f} k iterations/s, fitness {fit.min():.4f}..{fit.max():.4f}", flush=True)
