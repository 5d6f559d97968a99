"""Inclusive wall time per frame of the chain's host-side functions (wrappers, no profiler): python tools/frame_hosttime.py"""
import os, sys, logging, queue, time, collections, functools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth, viewer_wire, icp_refine, registration, cloud_ops, geometry, ray_projection, compat, depth_filters
from pedp_hip.frame_chain import bench_frame_setup

acc = collections.defaultdict(lambda: [0.0, 0])
def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or f"{getattr(obj, '__name__', obj.__class__.__name__)}.{name}"
    @functools.wraps(fn)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc[label]; e[0] += time.perf_counter() - t; e[1] += 1
    setattr(obj, name, w)

f = synth.Frame("bench_100k")
m = _lib.Mesh(_lib.default_context(), f.verts_posed, f.tris)
t_hit = m.cast_rays(f.rays6, want_uv=False)["t_hit"]
viewer_wire.attach_queues(viewer_wire.LatestQueue())
root = logging.getLogger(); sink = logging.StreamHandler(open(os.devnull, "w")); sink.setFormatter(logging.Formatter("[%(funcName)s()] %(message)s")); root.addHandler(sink); root.setLevel(logging.INFO)
chain, depth_m, heat, init_pose = bench_frame_setup(f, t_hit)
for k in range(5):
    chain.process(depth_m, init_pose(), heat, seed=0)
for mod, names in [(icp_refine, ["preprocess_source", "predict_z_axis_adjustment", "improve_result", "refine_pose_with_icp", "transform_object", "preprocess_target"]),
                   (registration, ["registration_icp_batch", "registration_icp", "upload"]),
                   (cloud_ops, ["preprocess_source_fused"]),
                   (_lib, ["icp_batched_ex", "icp", "host_array"]),
                   (_lib.Mesh, ["project_heatmap", "posed_vertices", "set_pose"]),
                   (_lib.Cloud, ["__init__"]),
                   (ray_projection.FrameProjector, ["project", "posed_mesh"]),
                   (geometry.PointCloud, ["__init__", "adopt", "_own"]),
                   (logging.Logger, ["info"]),
                   (np, ["array2string"]), (np.linalg, ["inv"]),
                   (depth_filters, ["depth_to_scene"]), (compat, ["update_dash_data"])]:
    for n in names:
        if hasattr(mod, n):
            wrap(mod, n)
# compat re-exports: the chain calls compat.refine_pose_with_icp
for n in ("refine_pose_with_icp", "preprocess_source", "improve_result"):
    if hasattr(compat, n) and hasattr(icp_refine, n):
        setattr(compat, n, getattr(icp_refine, n))
N = 40
t0 = time.perf_counter()
for k in range(N):
    chain.process(depth_m, init_pose(), heat, seed=0)
tot = time.perf_counter() - t0
print(f"frame {1e3 * tot / N:.3f} ms")
for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:50s} {1e3 * t / N:7.3f} ms/frame  {c / N:5.1f} calls  {1e6 * t / max(c, 1):8.1f} us/call")
