"""Where refine_pose_with_icp spends a config-5 frame: wall time (device synchronised) of its stages and of every batched
registration inside them: python tools/refine_breakdown.py"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pedp_hip import _lib, icp_refine, registration as reg, synth
from pedp_hip.frame_chain import bench_frame_setup

f = synth.Frame("bench_100k")
m = _lib.Mesh(_lib.default_context(), f.verts_posed, f.tris)
t_hit = m.cast_rays(f.rays6, want_uv=False)["t_hit"]
chain, depth_m, heat, init_pose = bench_frame_setup(f, t_hit)
import queue
from pedp_hip import viewer_wire
viewer_wire.attach_queues(viewer_wire.LatestQueue())
acc, calls = collections.defaultdict(float), collections.defaultdict(list)

def timed(mod, name, label=None, describe=None):
    fn = getattr(mod, name)
    def wrapper(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t0)
        acc[label or name] += dt
        calls[label or name].append((dt, describe(*a, **k) if describe else ""))
        return r
    setattr(mod, name, wrapper)

for n in ("preprocess_target", "preprocess_source", "predict_z_axis_adjustment", "improve_result"):
    timed(icp_refine, n)
timed(reg, "registration_icp_batch", describe=lambda s, t, radii, starts, *a, **k: f"{len(starts)} poses, {a[1].max_iteration if len(a) > 1 else 30} its")
for _ in range(3):
    chain.process(depth_m, init_pose(), heat, seed=0)
acc.clear(); calls.clear()
N = 5
t0 = time.perf_counter()
for _ in range(N):
    chain.process(depth_m, init_pose(), heat, seed=0, timed=True)
print({k: round(v, 3) for k, v in chain.stage_ms.items()})
for k, v in acc.items():
    print(f"{k:28s} {v / N:7.3f} ms per frame, {len(calls[k]) / N:.1f} calls")
per = collections.defaultdict(list)
for dt, d in calls["registration_icp_batch"]:
    per[d].append(dt)
for d, v in per.items():
    print(f"   batch {d:18s} x{len(v) / N:4.1f} per frame, median {np.median(v):.3f} ms")
