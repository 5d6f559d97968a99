"""Timing of one 20-iteration registration at bench size and of pedp_icp_batched (32 poses): python tools/icp_only.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth
ctx = _lib.Context(0)
f = synth.Frame("bench_100k")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
depth = mesh.cast_rays(f.rays6, want_uv=False)["t_hit"]
src = _lib.Cloud(ctx, f.scene(depth)); tgt = _lib.Cloud(ctx, f.model_points, f.normals)
for rep in range(3):
    t0 = time.perf_counter()
    r = _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), max_iteration=20, relative_fitness=-1, relative_rmse=-1)
    t1 = time.perf_counter()
    print(f"icp {1e3*(t1-t0):.2f} ms", _lib.icp_last_stats(ctx), r["fitness"], flush=True)
# batched: B perturbed start poses, sequential loop vs pedp_icp_batched (8 streams)
rng = np.random.default_rng(0)
B = 32
inits = np.repeat(f.icp_init()[None], B, 0).copy()
inits[:, :3, 3] += rng.normal(0, 0.5, (B, 3))
t0 = time.perf_counter()
seq = [_lib.icp(ctx, src, tgt, 10.0, inits[b], max_iteration=20, relative_fitness=-1, relative_rmse=-1) for b in range(B)]
t1 = time.perf_counter()
for rep in range(3):
    t2 = time.perf_counter()
    bat = _lib.icp_batched(ctx, src, tgt, 10.0, inits, max_iteration=20)
    t3 = time.perf_counter()
    print(f"batched B={B}: {1e3*(t3-t2):.2f} ms  (sequential {1e3*(t1-t0):.2f} ms)", flush=True)
Tb = bat[0]
print("max |T_batched - T_seq|", max(np.abs(Tb[b] - seq[b]["T"]).max() for b in range(B)))
