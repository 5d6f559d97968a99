"""Timing of the preprocess_source operations on the bench frame's scene cloud (368,640 points:
object + back plane), host arrays in and out: python tools/cloudops_only.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, cloud_ops, synth
from pedp_hip.compat import PointCloud, preprocess_source

ctx = _lib.default_context()
f = synth.Frame("bench_100k")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
scene = f.scene(mesh.cast_rays(f.rays6, want_uv=False)["t_hit"])


def timed(label, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); best = min(best, time.perf_counter() - t0)
    print(f"{label:58s} {1e3 * best:9.2f} ms", flush=True)
    return out


down, _ = timed(f"voxel_down_sample 5 mm ({len(scene)} points)", lambda: cloud_ops.voxel_down_sample(scene, 5.0))
print("   ->", len(down), "voxels")
plane, inl = timed(f"segment_plane 1000 iterations ({len(down)} points)", lambda: cloud_ops.segment_plane(down, 2.0, 3, 1000))
rest = np.delete(down, inl, axis=0)
labels = timed(f"cluster_dbscan eps 10 min 10 ({len(rest)} points)", lambda: cloud_ops.cluster_dbscan(rest, 10, 10))
big = rest[labels == np.bincount(labels[labels >= 0]).argmax()]
keep = timed(f"remove_statistical_outlier k 75 ({len(big)} points)", lambda: cloud_ops.remove_statistical_outlier(big, 75, 0.01))
fine, _ = timed("voxel_down_sample 1 mm", lambda: cloud_ops.voxel_down_sample(scene, 1.0))
timed(f"cluster_dbscan eps 10 min 10 ({len(fine)} points)", lambda: cloud_ops.cluster_dbscan(fine, 10, 10), reps=2)
timed(f"knn_mean_distance k 75 ({len(fine[::4])} points)", lambda: cloud_ops.knn_mean_distance(fine[::4], 75), reps=2)
param = {"preprocess_source": {"down_sample": 5, "plane_removal": {"distance_threshold": 2.0, "num_iterations": 1000}}, "box": False, "mesh": False}
out = timed("preprocess_source (whole chain, first frame)", lambda: preprocess_source(PointCloud(scene), None, param, i=0)[0])
print("   ->", len(out.points), "points reach ICP")
