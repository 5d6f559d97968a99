"""Scratch performance probe (GPU box): sweep timings of the ray and NN kernels at the
bench_100k configuration for a few launch configurations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "bench_100k"
ctx = _lib.Context(0)
f = synth.Frame(cfg)
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
print("config", cfg, "rays", f.n_rays, "tris", f.n_tris, flush=True)
res = None
for variant, chunks in [(1, 0), (1, 32), (3, 0), (0, 0)]:
    _lib.raycast_configure(ctx, chunks, variant)
    ts = []
    for rep in range(4):
        res = mesh.cast_rays(f.rays6, want_uv=False)
        ts.append(_lib.raycast_last_sweep_ms(ctx))
    ms = min(ts)
    tests = f.n_rays * f.n_tris
    print(f"ray variant {variant} chunks {chunks:3d}: sweep {ms:8.3f} ms  {f.n_rays/ms/1e3:7.2f} Mrays/s  "
          f"{46*tests/ms/1e9:7.1f} TFLOP/s(46/test)  hits {np.isfinite(res['t_hit']).sum()}", flush=True)
# general-origin path of the packed kernel: perturb one origin by an ulp
rays_g = f.rays6.copy(); rays_g[0, 0] = np.float32(1e-30)
_lib.raycast_configure(ctx, 0, 1)
ts = []
for rep in range(4):
    mesh.cast_rays(rays_g, want_uv=False); ts.append(_lib.raycast_last_sweep_ms(ctx))
print(f"ray variant 1 general-origin: sweep {min(ts):8.3f} ms  {f.n_rays/min(ts)/1e3:7.2f} Mrays/s  {46*f.n_rays*f.n_tris/min(ts)/1e9:7.1f} TFLOP/s(46/test)", flush=True)
_lib.raycast_configure(ctx, 0, 0)
scene = f.scene(res["t_hit"])
src = _lib.Cloud(ctx, scene)
tgt = _lib.Cloud(ctx, f.model_points, f.normals)
for rep in range(3):
    t0 = time.perf_counter()
    idx, d2 = _lib.nn(ctx, src, tgt, f.icp_init())
    t1 = time.perf_counter()
    ms = _lib.nn_last_sweep_ms(ctx)
    pairs = len(scene) * len(f.model_points)
    print(f"nn sweep {ms:8.3f} ms  ({8*pairs/ms/1e9:6.1f} TFLOP/s MFMA-accounting)  wall {1e3*(t1-t0):.1f} ms", flush=True)
for est in (0, 1):
    for rep in range(2):
        t0 = time.perf_counter()
        r = _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), estimator=est, max_iteration=20, relative_fitness=-1, relative_rmse=-1)
        t1 = time.perf_counter()
        print(f"icp est {est}: 20 iters wall {1e3*(t1-t0):8.2f} ms -> {20/(t1-t0):7.1f} iters/s  fitness {r['fitness']:.5f} rmse {r['inlier_rmse']:.4f}", flush=True)
    print(np.abs(np.linalg.inv(r["T"]) - f.T_gt).max())
