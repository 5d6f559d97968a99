"""The all-pairs NN sweep (every scene point x every model point of the bench frame), bf16 matrix pipe against PEDP_NN_F32=1:
python tools/nn_sweep_ab.py   (run once per setting; prints the sweep kernel's time, the fallback count and checks the result against the other
setting's file when it exists)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth
ctx = _lib.Context(0)
f = synth.Frame("bench_100k")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
depth = mesh.cast_rays(f.rays6, want_uv=False)["t_hit"]
scene = f.scene(depth)
src = _lib.Cloud(ctx, scene); tgt = _lib.Cloud(ctx, f.model_points, f.normals)
tag = os.environ.get("PEDP_NN_BF16_FORM", "32") + "bf" if os.environ.get("PEDP_NN_F32") != "1" else "f32"
ts, ks = [], []
for _ in range(6):
    t0 = time.perf_counter(); idx, d2 = _lib.nn(ctx, src, tgt, f.icp_init()); ts.append(1e3 * (time.perf_counter() - t0)); ks.append(_lib.nn_last_sweep_ms(ctx))
passes, pairs, fb = _lib.icp_last_stats(ctx)
print(f"{tag}: pedp_nn {np.median(ts):.3f} ms, sweep kernel {np.median(ks):.3f} ms, pairs {pairs:.3e}, fallback points {fb}")
out = f"/tmp/nn_{tag}.npz"
np.savez(out, idx=idx, d2=d2)
import glob
for other in sorted(glob.glob("/tmp/nn_*.npz")):
    if other != out:
        o = np.load(other)
        print(f"   equal to {os.path.basename(other)}:", bool(np.array_equal(o["idx"], idx) and np.array_equal(o["d2"], d2)))
# exhaustive registration (every pass all pairs)
_lib.icp_configure(ctx, exhaustive=True, timed_pass=1)
r = _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), max_iteration=20, relative_fitness=-1, relative_rmse=-1)
t0 = time.perf_counter(); r = _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), max_iteration=20, relative_fitness=-1, relative_rmse=-1); dt = time.perf_counter() - t0
print(f"   exhaustive 20-iteration registration {1e3 * dt:.2f} ms, timed sweep {_lib.nn_last_sweep_ms(ctx):.3f} ms, fitness {r['fitness']:.12f} rmse {r['inlier_rmse']:.12f}, stats {_lib.icp_last_stats(ctx)}")
