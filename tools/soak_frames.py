"""Soak: N frames of the config-5 chain, device memory before / after and the frame time's drift: python tools/soak_frames.py [N]"""
import os, sys, time, queue
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pedp_hip import _lib, synth, viewer_wire
from pedp_hip.frame_chain import bench_frame_setup
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
f = synth.Frame("bench_100k")
m = _lib.Mesh(_lib.default_context(), f.verts_posed, f.tris)
t_hit = m.cast_rays(f.rays6, want_uv=False)["t_hit"]
chain, depth_m, heat, init_pose = bench_frame_setup(f, t_hit)
q = queue.Queue()
viewer_wire.attach_queues(q)
rng = np.random.default_rng(0)
for _ in range(5):
    chain.process(depth_m, init_pose(), heat, seed=0)
free0 = torch.cuda.mem_get_info()[0]
ts = []
for k in range(n):
    d = (depth_m + rng.normal(0, 0.0002, depth_m.shape).astype(np.float32))   # a new image every frame
    t0 = time.perf_counter()
    if k % 3 == 0:
        out = chain.process(d, init_pose(), heat, seed=k)                      # a re-detection: frame 0's branch
        pose = np.linalg.inv(out["icp"].transformation); pose[:3, 3] += (0.8, -0.5, 1.0)
    else:
        del chain.intersection_pcds[1:]
        out = chain.process_tracking(d, pose.copy(), heat, i=k, seed=k)        # tracking frames in between
    ts.append(time.perf_counter() - t0)
    while not q.empty():
        q.get_nowait()
free1 = torch.cuda.mem_get_info()[0]
ts = 1e3 * np.array(ts)
import resource
print(f"host RSS peak {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024:.0f} MiB; pinned pool {dict((k, len(v)) for k, v in _lib._HOST_POOL.items())}; p99 {np.percentile(ts, 99):.2f} ms")
print(f"{n} frames: first 50 median {np.median(ts[:50]):.2f} ms, last 50 median {np.median(ts[-50:]):.2f} ms, max {ts.max():.2f} ms; "
      f"device memory free {free0 / 2**20:.0f} -> {free1 / 2**20:.0f} MiB; last fitness {out['icp'].fitness:.4f}")
