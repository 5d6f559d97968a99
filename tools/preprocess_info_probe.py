"""preprocess_source of the config-5 frame under run.py's arguments (background cloud, INFO): wall time per call, first and
tracking frames, quiet against INFO.  Under rocprofv3 --kernel-trace the kernels of one call are listed by
tools/prof_timeline.py <db> ransac_plane_kernel"""
import os, sys, time, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pedp_hip import _lib, synth
from pedp_hip.compat import PointCloud, preprocess_source

ctx = _lib.default_context()
f = synth.Frame("bench_100k")
m = _lib.Mesh(ctx, f.verts_posed, f.tris)
t_hit = m.cast_rays(f.rays6, want_uv=False)["t_hit"]
rng = np.random.default_rng(0)
z = np.where(np.isfinite(t_hit), t_hit * f.dirs[:, 2], 600.0) + rng.normal(0.0, 0.5, t_hit.shape)
pts = f.dirs * (z / f.dirs[:, 2])[:, None]
pin = torch.empty(pts.shape, dtype=torch.float64, pin_memory=True); pin.copy_(torch.from_numpy(pts)); pinned = pin.numpy()
bg = PointCloud(f.dirs / f.dirs[:, 2:3] * 600.0)
root = logging.getLogger()
sink = logging.StreamHandler(open(os.devnull, "w")); sink.setFormatter(logging.Formatter("[%(funcName)s()] %(message)s")); root.addHandler(sink)
def param():
    return {"preprocess_source": {"down_sample": 2, "plane_removal": {"distance_threshold": 2.0, "num_iterations": 500}}, "box": False, "mesh": False}
for level in (logging.WARNING, logging.INFO):
    root.setLevel(level)
    for i in (0, 1):
        ts = []
        for _ in range(14):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = preprocess_source(PointCloud(pinned), bg, param(), i=i)
            ts.append(1e3 * (time.perf_counter() - t0))
        print(f"{logging.getLevelName(level):8s} i={i}: {np.median(ts[3:]):.3f} ms -> {len(r[0].points)} points")
