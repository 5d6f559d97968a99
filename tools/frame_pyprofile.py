"""cProfile of the frame chain's host code (run.py's arguments), sorted by own time: python tools/frame_pyprofile.py"""
import os, sys, logging, queue, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth, viewer_wire
from pedp_hip.frame_chain import bench_frame_setup
f = synth.Frame("bench_100k")
m = _lib.Mesh(_lib.default_context(), f.verts_posed, f.tris)
t_hit = m.cast_rays(f.rays6, want_uv=False)["t_hit"]
viewer_wire.attach_queues(viewer_wire.LatestQueue())
root = logging.getLogger(); sink = logging.StreamHandler(open(os.devnull, "w")); sink.setFormatter(logging.Formatter("[%(funcName)s()] %(message)s")); root.addHandler(sink); root.setLevel(logging.INFO)
chain, depth_m, heat, init_pose = bench_frame_setup(f, t_hit)
for k in range(5):
    chain.process(depth_m, init_pose(), heat, seed=0)
pr = cProfile.Profile(); pr.enable()
for k in range(40):
    chain.process(depth_m, init_pose(), heat, seed=0)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(70); print(s.getvalue())
