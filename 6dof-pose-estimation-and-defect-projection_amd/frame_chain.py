"""The reference's per-frame geometry chain (BASELINE config 5) as one object, every stage on the GPU:

    depth image -> erode_depth -> bilateral_filter_depth -> depth2xyzmap_batch        estimater.py:255-259
    -> scene cloud (mm) -> preprocess_source -> z search + randomised ICP restarts    run.py:95-99
       (refine_pose_with_icp, src/pose_estimation.py:749-822)
    -> posed mesh -> ray projection of the heat map                                   run.py:109-119
    -> update_dash_data message                                                       run.py:131

The FoundationPose networks that supply the start pose are out of scope: the caller passes one.
Used by tests/test_stream_gpu.py (every stage against the oracle's chain), by bench.py's `frame_chain`
region (the driver's clock) and by tools/stream_latency.py (stage times)."""
import time

import numpy as np

from . import compat
from .compat import PointCloud
from .ray_projection import FrameProjector


class FrameChain:
    def __init__(self, model_points, model_normals, triangles, intrinsic, K32, color_to_depth, params, heat_threshold=0.75):
        import torch

        self.torch = torch
        self.model = PointCloud(model_points, normals=model_normals)
        self.mesh = compat.TriangleMesh(model_points, triangles)
        self.params = params
        self.color_to_depth = np.asarray(color_to_depth, np.float64)
        self.proj = FrameProjector(self.mesh, intrinsic, self.color_to_depth)
        self.K32 = torch.as_tensor(np.asarray(K32, np.float32), device="cuda")[None]
        self.heat_threshold = heat_threshold
        self.host_pts = [None, None]  # pinned, live across frames: a pageable destination makes the runtime pin and unpin
        self.frame_no = 0             # 9 MB per frame, which holds up the next submissions by 20-30 ms (DESIGN s6); TWO of
        self.stage_ms = {}            # them in turn, so a frame's scene cloud stays valid while the next one is written

    def process(self, depth_m, init, heat, seed=0, device_scene=False, timed=False):
        """One frame.  depth_m: H x W float32 metres (numpy or CUDA tensor); init: start pose scene -> model (mm);
        heat: H x W heat map of the colour camera.  Returns a dict of every stage's product."""
        torch = self.torch
        laps = [time.perf_counter()]

        def lap(name):
            if timed:
                torch.cuda.synchronize()
                laps.append(time.perf_counter())
                self.stage_ms[name] = 1e3 * (laps[-1] - laps[-2])

        d = depth_m if torch.is_tensor(depth_m) else torch.from_numpy(depth_m)
        d = d.cuda()
        d = compat.erode_depth(d, radius=2, device="cuda")
        d = compat.bilateral_filter_depth(d, radius=2, device="cuda")
        xyz = compat.depth2xyzmap_batch(d[None], self.K32, zfar=np.inf)[0]
        lap("depth filters + back-projection")
        dev_pts = xyz[xyz[..., 2] >= 0.001].double() * 1000.0                  # scene cloud in mm (run.py works in mm)
        if device_scene:
            source, pts = PointCloud(dev_pts), None
        else:
            k = self.frame_no & 1
            self.frame_no += 1
            if self.host_pts[k] is None or len(self.host_pts[k]) < len(dev_pts):
                self.host_pts[k] = torch.empty((max(len(dev_pts), d.numel()), 3), dtype=torch.float64, pin_memory=True)
            self.host_pts[k][: len(dev_pts)].copy_(dev_pts)
            pts = self.host_pts[k][: len(dev_pts)].numpy()   # the pinned array itself (no second 9-MB copy): valid until
            source = PointCloud(pts)                         # the frame after next is processed
        lap("scene cloud")
        np.random.seed(seed)
        _, icp, z, _ = compat.refine_pose_with_icp(source, self.model, None, init, self.params)   # run.py:95-99
        lap("refine_pose_with_icp")
        model_in_scene = np.linalg.inv(icp.transformation)                                        # run.py:109-119
        mesh_copy = compat.transform_object(self.mesh, model_in_scene)
        cloud = self.proj.project(model_in_scene, heat, self.heat_threshold)
        cloud.transform(self.color_to_depth)
        lap("posed mesh + projection")
        msg = compat.update_dash_data([cloud], mesh_copy)                                         # run.py:131
        lap("viewer message")
        return {"depth": d, "xyz": xyz, "points": pts, "n_points": int(len(dev_pts)), "init": init, "icp": icp, "z": z,
                "cloud": cloud, "mesh": mesh_copy, "message": msg}


def bench_frame_setup(frame, t_hit):
    """The synthetic config-5 set-up on a bench frame: depth image in metres (0.5 mm noise, 600 mm background), model,
    intrinsics, colour-to-depth offset, heat map, parameters.  Returns (chain, depth_m, heat, init_fn)."""
    from . import synth
    from .compat import PinholeCameraIntrinsic

    f = frame
    rng = np.random.default_rng(0)
    z_mm = np.where(np.isfinite(t_hit), t_hit * f.dirs[:, 2], 600.0) + rng.normal(0.0, 0.5, t_hit.shape)
    depth_m = (z_mm / 1000.0).reshape(f.height, f.width).astype(np.float32)   # the filters work in metres
    intr = PinholeCameraIntrinsic(f.width, f.height, intrinsic_matrix=f.K)
    color_to_depth = np.eye(4)
    color_to_depth[:3, 3] = (2.0, -1.0, 0.5)
    heat = np.zeros((f.height, f.width))
    heat[200:380, 220:420] = np.linspace(0.76, 1.0, 200)[None, :]
    params = {"preprocess_target": {"max_pcd": 100000, "keep_normals": True},
              "preprocess_source": {"down_sample": 2, "plane_removal": {"distance_threshold": 2.0, "num_iterations": 500}},
              "box": False, "mesh": False,
              "refine_registration": {"distance_threshold": 6.0}, "run_icp": {"fitness_threshold": 0.97, "rmse_threshold": 0.8}}
    chain = FrameChain(f.model_points, f.normals, f.tris, intr, f.K.astype(np.float32), color_to_depth, params)

    def init_pose():
        init = synth.start_pose()
        init[2, 3] += 5.0
        return init

    return chain, depth_m, heat, init_pose
