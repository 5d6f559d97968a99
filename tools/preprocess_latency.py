"""preprocess_source on the config-5 frame's scene cloud (365 k points): the one-call device-resident chain from a host
array and from a device tensor, against the step-by-step path: python tools/preprocess_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pedp_hip import _lib, cloud_ops, icp_refine, synth
from pedp_hip.compat import PointCloud, preprocess_source

ctx = _lib.default_context()
f = synth.Frame("bench_100k")
m = _lib.Mesh(ctx, f.verts_posed, f.tris)
t_hit = m.cast_rays(f.rays6, want_uv=False)["t_hit"]
rng = np.random.default_rng(0)
z = np.where(np.isfinite(t_hit), t_hit * f.dirs[:, 2], 600.0) + rng.normal(0.0, 0.5, t_hit.shape)
pts = f.dirs * (z / f.dirs[:, 2])[:, None]
param = {"preprocess_source": {"down_sample": 2, "plane_removal": {"distance_threshold": 2.0, "num_iterations": 500}}, "box": False, "mesh": False}
dev = torch.from_numpy(pts).cuda()
pin = torch.empty(pts.shape, dtype=torch.float64, pin_memory=True); pin.copy_(torch.from_numpy(pts)); pinned = pin.numpy()

def med(fn, n=12):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts[2:])), r

for name, src in (("pageable host array", pts), ("pinned host array", pinned), ("device tensor", dev)):
    t, r = med(lambda: preprocess_source(PointCloud(src), None, dict(param), i=0))
    print(f"one call, {name:20s}: {t:6.2f} ms  -> {len(r[0].points)} points")
icp_refine._FORCE_STEPS = True
t, r = med(lambda: preprocess_source(PointCloud(pinned), None, dict(param), i=0), 6)
print(f"step by step, pinned host array : {t:6.2f} ms  -> {len(r[0].points)} points")
