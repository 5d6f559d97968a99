"""BASELINE config 3 (256 start poses, 20 iterations each, one launch per pass and group of 32): python tools/batch_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth
ctx = _lib.Context(0)
f = synth.Frame("bench_100k")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
depth = mesh.cast_rays(f.rays6, want_uv=False)["t_hit"]
src = _lib.Cloud(ctx, f.scene(depth)); tgt = _lib.Cloud(ctx, f.model_points, f.normals)
inits = np.stack([np.linalg.inv(T) for T in synth.batched_start_poses(256)])
ts = []
for _ in range(4):
    t0 = time.perf_counter(); out = _lib.icp_batched(ctx, src, tgt, f.max_correspondence_distance, inits, max_iteration=20); ts.append(time.perf_counter() - t0)
t = min(ts[1:])
print(f"256 poses x 20 iterations: {1e3 * t:.2f} ms = {256 * 20 / t / 1e3:.1f} k its/s; checksum {float(np.sum(out[0])):.12f}")
