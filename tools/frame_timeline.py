"""One frame of the chain under rocprofv3 --kernel-trace (tools/prof_timeline.py <db> erode_kernel lists its kernels)."""
import os, sys, logging, queue
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth, viewer_wire
from pedp_hip.frame_chain import bench_frame_setup
f = synth.Frame("bench_100k")
m = _lib.Mesh(_lib.default_context(), f.verts_posed, f.tris)
t_hit = m.cast_rays(f.rays6, want_uv=False)["t_hit"]
viewer_wire.attach_queues(viewer_wire.LatestQueue())
root = logging.getLogger(); sink = logging.StreamHandler(open(os.devnull, "w")); root.addHandler(sink); root.setLevel(logging.INFO)
chain, depth_m, heat, init_pose = bench_frame_setup(f, t_hit)
for k in range(12):
    chain.process(depth_m, init_pose(), heat, seed=0)
print("done")
