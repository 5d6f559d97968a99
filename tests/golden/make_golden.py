"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle.

The reference cannot be imported (its first statement is `import open3d`, absent here) and
holds no fixtures of its own, so these vectors pin the ORACLE against regressions and give
the GPU tests fixed inputs/outputs; they do not pin the oracle to the reference ("parity
unpinned", see oracle/pedp_oracle.h).  Run from the repository root:
    python tests/golden/make_golden.py
Seeds: numpy default_rng(0) inside synth.scene_from_depth; np.random.seed(0) for G6.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import pedp_oracle as oracle  # noqa: E402
from pedp_hip import synth  # noqa: E402


def icosphere42():
    """12 icosahedron vertices + 30 edge midpoints, on the unit sphere (one subdivision)."""
    p = (1 + 5 ** 0.5) / 2
    v = np.array([[-1, p, 0], [1, p, 0], [-1, -p, 0], [1, -p, 0], [0, -1, p], [0, 1, p], [0, -1, -p], [0, 1, -p],
                  [p, 0, -1], [p, 0, 1], [-p, 0, -1], [-p, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    d = np.linalg.norm(v[:, None] - v[None], axis=2)
    edge = np.min(d[d > 0])
    mids = [(v[i] + v[j]) / 2 for i in range(12) for j in range(i + 1, 12) if abs(d[i, j] - edge) < 1e-9]
    mids = np.array(mids)
    mids /= np.linalg.norm(mids, axis=1, keepdims=True)
    return np.vstack([v, mids])


def rotation_grid():
    """The 252-pose grid of estimater.py:104-116 (42 look-at views x 6 in-plane rotations)."""
    verts = icosphere42()
    grid = []
    for c in verts:
        z = -c / np.linalg.norm(c)
        x = np.cross([0.0, 0.0, 1.0], z)
        if np.allclose(x, 0):
            x = np.array([1.0, 0.0, 0.0])
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        y /= np.linalg.norm(y)
        cam_in_ob = np.eye(4)
        cam_in_ob[:3, 0], cam_in_ob[:3, 1], cam_in_ob[:3, 2], cam_in_ob[:3, 3] = x, y, z, c
        for a in np.deg2rad(np.arange(0, 360, 60)):
            Rz = np.eye(4)
            Rz[:3, :3] = synth.rot_z(a)
            grid.append(np.linalg.inv(cam_in_ob @ Rz))
    return np.asarray(grid)


def main():
    # G1: single triangle / unit cube ray tables with analytic answers
    tri_v = np.array([[0, 0, 5], [1, 0, 5], [0, 1, 5]], np.float32)
    tri_t = np.array([[0, 1, 2]], np.uint32)
    tri_rays = np.array([[0.25, 0.25, 0, 0, 0, 1], [0.25, 0.25, 0, 0, 0, -1], [0, 0, 0, 0, 0, 1], [0.5, 0.5, 0, 0, 0, 1],
                         [0.6, 0.6, 0, 0, 0, 1], [0.25, 0.25, 10, 0, 0, -1], [0.25, 0.25, 0, 0, 0, 2],
                         [0.25, 0.25, 0, 1, 0, 0]], np.float32)
    r = oracle.raycast(tri_v, tri_t, tri_rays)
    np.savez(os.path.join(HERE, "g1_triangle.npz"), verts=tri_v, tris=tri_t, rays=tri_rays, t_hit=r["t_hit"],
             ids=r["primitive_ids"], uv=r["primitive_uvs"],
             t_analytic=np.array([5, np.inf, 5, 5, np.inf, 5, 2.5, np.inf], np.float32))

    # G2: 1,000-triangle torus x 64x48 rays
    verts, tris, normals = synth.bumpy_torus(25, 20)
    vp = synth.posed_vertices(verts, synth.gt_pose())
    dirs = synth.pixel_rays(64, 48, 50.4, 31.5, 23.5)
    rays6 = synth.rays6_from_dirs(dirs)
    r = oracle.raycast(vp, tris, rays6)
    np.savez_compressed(os.path.join(HERE, "g2_torus_rays.npz"), verts_posed=vp, tris=tris, rays6=rays6,
                        t_hit=r["t_hit"], ids=r["primitive_ids"], uv=r["primitive_uvs"])

    # G3/G4: ICP traces on the tiny frame (noise-free known answer, and seeded noise, both estimators)
    f = synth.Frame("tiny")
    depth = oracle.raycast(f.verts_posed, f.tris, f.rays6)["t_hit"]
    hit = np.isfinite(depth)
    clean = synth.scene_from_depth(depth[hit], f.dirs[hit], noise_sigma=0.0)
    noisy = f.scene(depth)
    out = {"scene_clean": clean, "scene_noisy": noisy, "model": f.model_points, "normals": f.normals,
           "init": f.icp_init(), "T_gt": f.T_gt}
    for name, scene in (("clean", clean), ("noisy", noisy)):
        for est in (0, 1):
            o = oracle.icp(scene, f.model_points, f.normals, 10.0, f.icp_init(), estimator=est, max_iter=20,
                           rel_fitness=-1, rel_rmse=-1, kdtree=False)
            out[f"trace_{name}_{est}"] = o["trace"]
            out[f"corr_{name}_{est}"] = o["corr"]
    np.savez_compressed(os.path.join(HERE, "g3g4_icp_traces.npz"), **out)

    # G5: cluster_poses on the 252-pose rotation grid
    grid = rotation_grid().astype(np.float32)
    sym_id = np.eye(4, dtype=np.float32)[None]
    sym_z2 = np.stack([np.eye(4), np.diag([-1.0, -1.0, 1.0, 1.0])]).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "g5_cluster_poses.npz"), grid=grid, sym_id=sym_id, sym_z2=sym_z2,
                        keep_id=oracle.cluster_poses(30, 99999, grid, sym_id),
                        keep_z2=oracle.cluster_poses(30, 99999, grid, sym_z2),
                        keep_tight=oracle.cluster_poses(5, 99999, grid, sym_id))

    # G6: improve_result trace with np.random.seed(0) (RNG consumption order pin)
    param = {"refine_registration": {"distance_threshold": 8.0},
             "run_icp": {"fitness_threshold": 0.999, "rmse_threshold": 0.05}}
    np.random.seed(0)
    trace = []
    res = oracle.improve_result(clean, f.model_points, f.normals, f.T_start, param, trace=trace)
    np.savez_compressed(os.path.join(HERE, "g6_improve_result.npz"),
                        thresholds=np.array([t[0] for t in trace]), fitness=np.array([t[1] for t in trace]),
                        rmse=np.array([t[2] for t in trace]), T=np.array([t[3] for t in trace]),
                        best_T=res.transformation, best_fitness=res.fitness, best_rmse=res.inlier_rmse,
                        rng_after=np.random.uniform())
    g7()
    g8()
    print("golden fixtures written to", HERE)


def g8_cloud(seed=0):
    """A table plane, an object blob, a small far blob and scattered outliers (mm), shuffled."""
    rng = np.random.default_rng(seed)
    plane = np.column_stack([rng.uniform(-60, 60, 900), rng.uniform(-40, 40, 900), 400 + rng.normal(0, 0.3, 900)])
    obj = rng.normal([5, -3, 370], [9, 7, 5], (500, 3))
    small = rng.normal([50, 30, 385], 2, (60, 3))
    noise = rng.uniform([-80, -60, 300], [80, 60, 430], (40, 3))
    pts = np.vstack([plane, obj, small, noise])
    return pts[rng.permutation(len(pts))]


def g8():
    """G8: point-cloud operations of preprocess_source on a 1,500-point cloud."""
    pts = g8_cloud()
    down, _ = oracle.voxel_down_sample(pts, 3.0)
    plane, inl = oracle.segment_plane(pts, 1.0, 100, seed=7)
    np.savez_compressed(os.path.join(HERE, "g8_cloud_ops.npz"), points=pts, voxel3=down,
                        dbscan_6_8=oracle.cluster_dbscan(pts, 6.0, 8), knn20=oracle.knn_mean_distance(pts, 20),
                        sor_20_1=oracle.remove_statistical_outlier(pts, 20, 1.0), plane=plane, plane_inliers=inl,
                        normals_8_12=oracle.estimate_normals(pts, 8.0, 12))


def g7():
    """G7: depth pre-filters on a 48 x 64 sensor-like image (default parameters and one
    non-default set), back-projection single and batched."""
    d = synth.depth_image(48, 64, seed=7)
    K = np.array([[60.0, 0, 31.5], [0, 61.0, 23.5], [0, 0, 1]])
    d2 = synth.depth_image(48, 64, seed=8, nan=False)
    Ks = np.stack([K, K * np.array([[1.1], [0.9], [1.0]])]).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "g7_depth_filters.npz"), depth=d, K=K, depths=np.stack([d, d2]), Ks=Ks,
                        erode=oracle.erode_depth(d), erode_r3=oracle.erode_depth(d, 3, 0.002, 0.5, 1.0),
                        bilateral=oracle.bilateral_filter_depth(d),
                        bilateral_r1=oracle.bilateral_filter_depth(d, 1, 1.0, 1.5, 0.02),
                        xyz=oracle.depth2xyzmap(d, K), xyz_batch=oracle.depth2xyzmap_batch(np.stack([d, d2]), Ks, 0.9))


if __name__ == "__main__":
    main()
