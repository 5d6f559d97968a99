"""What run.py's arguments cost the frame chain: frame 0 and a tracking frame with / without the background cloud and
with the root logger at INFO / WARNING, stage by stage; cProfile of the run.py configuration:
python tools/frame_args_probe.py [--profile]"""
import os, sys, time, logging, collections, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pedp_hip import _lib, icp_refine, synth, viewer_wire
from pedp_hip.frame_chain import bench_frame_setup
import queue

f = synth.Frame("bench_100k")
m = _lib.Mesh(_lib.default_context(), f.verts_posed, f.tris)
t_hit = m.cast_rays(f.rays6, want_uv=False)["t_hit"]
viewer_wire.attach_queues(viewer_wire.LatestQueue())
root = logging.getLogger()
sink = logging.StreamHandler(open(os.devnull, "w"))
sink.setFormatter(logging.Formatter("[%(funcName)s()] %(message)s"))
root.addHandler(sink)
acc = collections.defaultdict(list)

def timed(mod, name):
    fn = getattr(mod, name)
    def wrapper(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[name].append(1e3 * (time.perf_counter() - t0))
        return r
    setattr(mod, name, wrapper)

for n in ("preprocess_target", "preprocess_source", "predict_z_axis_adjustment", "improve_result"):
    timed(icp_refine, n)
from pedp_hip import compat
for n in ("preprocess_source", "improve_result"):
    setattr(compat, n, getattr(icp_refine, n))

for bg in (False, True):
    for level in (logging.WARNING, logging.INFO):
        chain, depth_m, heat, init_pose = bench_frame_setup(f, t_hit, with_background=bg)
        root.setLevel(level)
        for _ in range(3):
            chain.process(depth_m, init_pose(), heat, seed=0)
        acc.clear()
        ts = []
        for _ in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = chain.process(depth_m, init_pose(), heat, seed=0)
            torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        chain.process(depth_m, init_pose(), heat, seed=0, timed=True)
        print(f"frame 0   background={bg!s:5} level={logging.getLevelName(level):7}: {np.median(ts):.2f} ms  "
              + "  ".join(f"{k} {np.median(v):.2f}" for k, v in acc.items()))
        print("          stages:", {k: round(v, 2) for k, v in chain.stage_ms.items()})
        pose = np.linalg.inv(out["icp"].transformation); pose[:3, 3] += (0.8, -0.5, 1.0)
        for k in range(3):
            chain.process_tracking(depth_m, pose.copy(), heat, i=1 + k, seed=k)
        acc.clear(); ts = []
        for k in range(8):
            del chain.intersection_pcds[1:]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            chain.process_tracking(depth_m, pose.copy(), heat, i=4 + k, seed=k)
            torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        chain.process_tracking(depth_m, pose.copy(), heat, i=20, seed=0, timed=True)
        print(f"tracking  background={bg!s:5} level={logging.getLevelName(level):7}: {np.median(ts):.2f} ms  "
              + "  ".join(f"{k} {np.median(v):.2f}" for k, v in acc.items()))
        print("          stages:", {k: round(v, 2) for k, v in chain.stage_ms.items()})
if "--profile" in sys.argv:
    for what in ("process", "process_tracking"):
        pr = cProfile.Profile()
        pr.enable()
        for k in range(20):
            if what == "process":
                chain.process(depth_m, init_pose(), heat, seed=0)
            else:
                del chain.intersection_pcds[1:]
                chain.process_tracking(depth_m, pose.copy(), heat, i=30 + k, seed=k)
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
        print(what, "x20\n", s.getvalue())
