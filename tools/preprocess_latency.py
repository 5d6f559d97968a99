"""Stage times of preprocess_source (src/pose_estimation.py:186-268) on the bench frame's raw scene cloud."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth
from pedp_hip import icp_refine as R
from pedp_hip.compat import PointCloud

ctx = _lib.default_context()
f = synth.Frame("bench_100k")
m = _lib.Mesh(ctx, f.verts_posed, f.tris)
t_hit = m.cast_rays(f.rays6, want_uv=False)["t_hit"]
scene = f.scene(t_hit)
if "--filtered" in sys.argv:   # the scene as the depth pre-filters hand it over (tests/test_stream_gpu.py)
    import torch
    from pedp_hip import compat
    rng = np.random.default_rng(0)
    z_mm = np.where(np.isfinite(t_hit), t_hit * f.dirs[:, 2], 600.0) + rng.normal(0.0, 0.5, t_hit.shape)
    d = torch.from_numpy((z_mm / 1000.0).reshape(f.height, f.width).astype(np.float32)).cuda()
    d = compat.bilateral_filter_depth(compat.erode_depth(d, radius=2), radius=2)
    xyz = compat.depth2xyzmap_batch(d[None], torch.as_tensor(f.K.astype(np.float32), device="cuda")[None], zfar=np.inf)[0]
    scene = (xyz[xyz[..., 2] >= 0.001].double() * 1000.0).cpu().numpy()
params = {"down_sample": 2, "plane_removal": {"distance_threshold": 2.0, "num_iterations": 500}}
for rep in range(5):
    T = [time.perf_counter()]
    lap = lambda: T.append(time.perf_counter())
    pcd = PointCloud(scene.copy() if "--fresh" in sys.argv else scene)
    down = pcd.voxel_down_sample(voxel_size=2); lap()
    plane, inl = R.perform_plane_segmentation(down, params["plane_removal"]); lap()
    R.estimate_normals(down, params); lap()
    avg = R.compute_average_normal(down); lap()
    plane, _ = R.flip_plane_normal_if_needed(plane, avg)
    rest = R.remove_plane(down, inl); lap()
    big = R.filter_largest_cluster(rest); lap()
    clean = R.remove_statistical_outliers(big, nb_neighbors=75, std_ratio=0.01); lap()
    R.estimate_normals(clean, params); lap()
    names = ["voxel 2mm", "plane RANSAC", "normals(down)", "avg normal", "remove_plane", "dbscan largest", "outliers k=75", "normals(clean)"]
    dt = np.diff(T) * 1e3
    print(f"rep {rep}: total {dt.sum():.1f} ms ({len(down.points)} -> {len(rest.points)} -> {len(big.points)} -> {len(clean.points)}) | " +
          " | ".join(f"{n} {v:.2f}" for n, v in zip(names, dt)), flush=True)
