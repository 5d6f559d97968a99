"""Timing of the fused defect projection (pedp_project_heatmap) and of a pose update of a posable
mesh at benchmark size: python tools/project_only.py [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "bench_100k"
ctx = _lib.Context(0)
f = synth.Frame(cfg)
mesh = _lib.Mesh(ctx, f.model_points, f.tris, posable=True)
hm = np.ones((f.height, f.width))
for rep in range(4):
    t0 = time.perf_counter()
    mesh.set_pose(f.T_gt); ctx.synchronize()
    t1 = time.perf_counter()
    out = mesh.project_heatmap(hm, f.K, 0.5)
    t2 = time.perf_counter()
    print(f"{cfg}: set_pose {1e3*(t1-t0):.3f} ms, project_heatmap (host heat map in, hits out) {1e3*(t2-t1):.3f} ms, "
          f"rays {out['n_rays']}, hits {len(out['points'])}", flush=True)
# stepwise path for comparison: host rays -> pedp_raycast (host mode) -> numpy filter
fixed = _lib.Mesh(ctx, f.verts_posed, f.tris)
for rep in range(2):
    t0 = time.perf_counter()
    r = fixed.cast_rays(f.rays6, want_uv=False)
    v = r["t_hit"] != np.inf
    pts = f.dirs[v] * r["t_hit"][v, None].astype(np.float64)
    t1 = time.perf_counter()
    print(f"stepwise cast_rays + host filter {1e3*(t1-t0):.3f} ms, hits {v.sum()}", flush=True)
print("hit points agree:", np.abs(pts - out["points"]).max())
