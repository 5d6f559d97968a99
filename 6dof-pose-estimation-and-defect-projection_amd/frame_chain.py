"""The reference's per-frame geometry chain (BASELINE config 5) as one object, every stage on the GPU -- both branches
of run.py's loop, with the arguments run.py passes (the reader's background cloud; the root logger at INFO is the
caller's business, bench.py and the tests set it):

  frame 0 (run.py:79-131), `process`
    depth image -> erode_depth -> bilateral_filter_depth -> depth2xyzmap_batch        estimater.py:255-259
    -> scene cloud (mm) -> refine_pose_with_icp(source, target, background, ...)      run.py:95-101
       (preprocess_source, z search, randomised ICP restarts; src/pose_estimation.py:749-822)
    -> delta_pose, posed mesh -> ray projection of the heat map                       run.py:104-119
    -> update_dash_data message                                                       run.py:131

  a tracking frame with a defect detection (run.py:132-207), `process_tracking`
    depth image -> ... -> scene cloud -> preprocess_source(source, background, parameters, i)   run.py:154-156
       (the call sets parameters['preprocess_source']['down_sample'] = 5 in the reader's dict, for good)
    -> improve_result(source_processed, target_processed, initial_transformation)     run.py:168-171  (a bare 4x4 seed)
    -> delta_pose, posed mesh, relative_transformation                                run.py:176-184
    -> ray projection -> every earlier hit cloud moved by relative_transformation     run.py:187-201
    -> update_dash_data message                                                       run.py:206
  a tracking frame without one (run.py:208-210): `track_only`, a 4x4 product.

The FoundationPose networks that supply the start pose are out of scope: the caller passes one.
Used by tests/test_stream_gpu.py (every stage against the oracle's chain), by bench.py's `frame_chain` and
`tracking_frame` regions (the driver's clock) and by tools/stream_latency.py (stage times)."""
import gc
import time

import numpy as np

from . import compat, depth_filters
from .compat import PointCloud
from .ray_projection import FrameProjector


class FrameChain:
    def __init__(self, model_points, model_normals, triangles, intrinsic, K32, color_to_depth, params, heat_threshold=0.75,
                 background=None, freeze_gc=True):
        import torch

        self.torch = torch
        self.model = PointCloud(model_points, normals=model_normals)                # reader.target
        self.mesh = compat.TriangleMesh(model_points, triangles)                    # reader.target_mesh
        self.background = background                                                 # reader.background (datareader.py:313-320)
        self.params = params                                                         # reader.parameters: ONE dict for the whole run
        self.color_to_depth = np.asarray(color_to_depth, np.float64)
        self.proj = FrameProjector(self.mesh, intrinsic, self.color_to_depth)
        self.K32 = torch.as_tensor(np.asarray(K32, np.float32), device="cuda")[None]
        self.K32_host = np.asarray(K32, np.float32).reshape(3, 3)
        self._depth_buffers = {}
        self.heat_threshold = heat_threshold
        self.host_pts = [None, None]  # pinned, live across frames: a pageable destination makes the runtime pin and unpin
        self.frame_no = 0             # 9 MB per frame, which holds up the next submissions by 20-30 ms (DESIGN s6); TWO of
        self.stage_ms = {}            # them in turn, so a frame's scene cloud stays valid while the next one is written
        self.reset()
        if freeze_gc:
            # A frame allocates a few thousand containers (dicts, tuples, holders); after a few dozen frames the collector
            # runs a FULL collection, and with torch and numpy imported that walks every object of the process: 43 ms
            # measured (tools/frame_gc_probe.py) -- ten frames' time, on one frame.  Everything alive now (modules,
            # the model, the mesh) is long-lived: gc.freeze() takes it out of the collector's walks for good; full
            # collections then see only what the loop itself made (< 0.1 ms).  Reference counting frees as before.
            gc.collect()
            gc.freeze()

    def reset(self):
        """The state run.py's loop carries from frame to frame (run.py:61, :104-107, :129)."""
        self.intersection_pcds = []
        self.delta_pose = self.current_transformation = self.previous_transformation = self.target_processed = None

    # ------------------------------------------------------------------ the stages both branches share
    def _laps(self, timed):
        torch, laps, marks = self.torch, [time.perf_counter()], {}

        def lap(name):
            if timed:
                torch.cuda.synchronize()
                laps.append(time.perf_counter())
                if name.startswith("  "):          # a part of the stage that follows: timed, and the stage still spans it
                    marks.setdefault("t0", laps[-2])
                    self.stage_ms[name] = 1e3 * (laps[-1] - laps[-2])
                else:
                    self.stage_ms[name] = 1e3 * (laps[-1] - marks.pop("t0", laps[-2]))

        return lap

    def _scene(self, depth_m, device_scene, lap):
        """estimater.py:255-259 and reader.get_source: filtered depth, back-projection, the valid points in mm."""
        torch = self.torch
        # erode_depth -> bilateral_filter_depth -> depth2xyzmap_batch -> the valid points in mm, as ONE library call
        # (pedp_depth_to_scene: one upload of the image, no trip to the host between the kernels; bit for bit what the three
        # calls and torch's `xyz[xyz[..., 2] >= 0.001].double() * 1000.0` give -- tests/test_depth_gpu.py)
        d, xyz, dev_pts = depth_filters.depth_to_scene(depth_m, self.K32_host, buffers=self._depth_buffers)
        lap("depth filters + back-projection")
        if device_scene:
            source, pts = PointCloud(dev_pts), None
        else:
            k = self.frame_no & 1
            self.frame_no += 1
            if self.host_pts[k] is None or len(self.host_pts[k]) < len(dev_pts):
                self.host_pts[k] = torch.empty((max(len(dev_pts), d.numel()), 3), dtype=torch.float64, pin_memory=True)
            self.host_pts[k][: len(dev_pts)].copy_(dev_pts)
            pts = self.host_pts[k][: len(dev_pts)].numpy()   # the pinned array itself (no second 9-MB copy): valid until
            source = PointCloud.borrowed(pts, device=dev_pts)   # the frame after next is processed (no copy; the device twin
                                                                # spares preprocess_source the upload of the bytes just downloaded)
        lap("scene cloud")
        return d, xyz, pts, int(len(dev_pts)), source

    def _project(self, model_in_scene, heat, lap=None):
        """transform_object + ray_tracing + the move into the depth camera's frame (run.py:109-118, :179-200)."""
        mesh_copy = self.proj.posed_mesh(model_in_scene, self.mesh)           # transform_object(reader.target_mesh, ...)
        if lap:
            lap("  posed mesh (viewer's copy)")
        cloud = self.proj.project(model_in_scene, heat, self.heat_threshold, into=self.color_to_depth)
        if lap:
            lap("  heat-map projection")
        return mesh_copy, cloud

    # ------------------------------------------------------------------ run.py:79-131
    def process(self, depth_m, init, heat, seed=0, device_scene=False, timed=False):
        """Frame 0.  depth_m: H x W float32 metres (numpy or CUDA tensor); init: FoundationPose's estimate, model ->
        depth camera (mm), receives the z adjustment in place like run.py's `initial_transformation`; heat: H x W heat
        map of the colour camera.  Returns a dict of every stage's product and starts the loop's state over."""
        lap = self._laps(timed)
        self.reset()
        d, xyz, pts, n_pts, source = self._scene(depth_m, device_scene, lap)
        np.random.seed(seed)
        _, icp, z, self.target_processed = compat.refine_pose_with_icp(source, self.model, self.background, init,
                                                                       self.params)                     # run.py:99-101
        lap("refine_pose_with_icp")
        model_in_scene = np.linalg.inv(icp.transformation)
        self.delta_pose = np.linalg.inv(init) @ model_in_scene                                          # run.py:104-105
        self.current_transformation = self.previous_transformation = icp.transformation                 # run.py:107, :129
        mesh_copy, cloud = self._project(model_in_scene, heat, lap if timed else None)                  # run.py:109-118
        if cloud is not None:
            self.intersection_pcds.append(cloud)
        lap("posed mesh + projection")
        msg = compat.update_dash_data(self.intersection_pcds, mesh_copy)                                # run.py:131
        lap("viewer message")
        return {"depth": d, "xyz": xyz, "points": pts, "n_points": n_pts, "init": init, "icp": icp, "z": z, "cloud": cloud,
                "mesh": mesh_copy, "message": msg, "delta_pose": self.delta_pose}

    # ------------------------------------------------------------------ run.py:132-207
    def process_tracking(self, depth_m, init, heat, i=1, seed=None, device_scene=False, timed=False):
        """Frame i > 0 with a defect detection pending.  init: the tracker's pose model -> depth camera (mm; not
        changed).  `seed` re-seeds the global numpy RNG (a test's handle on improve_result's draws; run.py seeds once,
        at start).  Returns a dict of every stage's product; the hit clouds of earlier detections have moved."""
        if self.target_processed is None:
            raise RuntimeError("process_tracking: no frame 0 yet (run.py:99 supplies target_processed)")
        lap = self._laps(timed)
        d, xyz, pts, n_pts, source = self._scene(depth_m, device_scene, lap)
        source_processed, _, _ = compat.preprocess_source(source, self.background, self.params, i=i)    # run.py:154-156
        lap("preprocess_source")
        if seed is not None:
            np.random.seed(seed)
        result = compat.improve_result(source_processed, self.target_processed, init, self.params)      # run.py:168-171
        lap("improve_result")
        current = result.transformation
        model_in_scene = np.linalg.inv(current)
        self.delta_pose = np.linalg.inv(init) @ model_in_scene                                          # run.py:176-178
        relative = model_in_scene @ self.previous_transformation                                        # run.py:183-184
        mesh_copy, cloud = self._project(model_in_scene, heat, lap if timed else None)                  # run.py:179-193
        for earlier in self.intersection_pcds:                                                          # run.py:196-197
            earlier.transform(relative)
        if cloud is not None:
            self.intersection_pcds.append(cloud)                                                        # run.py:200-201
        self.current_transformation = self.previous_transformation = current                            # run.py:174, :204
        lap("posed mesh + projection")
        msg = compat.update_dash_data(self.intersection_pcds, mesh_copy)                                # run.py:206
        lap("viewer message")
        return {"depth": d, "xyz": xyz, "points": pts, "n_points": n_pts, "source_processed": source_processed, "icp": result,
                "cloud": cloud, "mesh": mesh_copy, "message": msg, "relative": relative, "delta_pose": self.delta_pose}

    def track_only(self, init):
        """Frame i > 0 without a detection (run.py:208-210): the pose follows the tracker through delta_pose."""
        self.current_transformation = np.linalg.inv(np.asarray(init, np.float64) @ self.delta_pose)
        return self.current_transformation


def bench_frame_setup(frame, t_hit, with_background=True):
    """The synthetic config-5 set-up on a bench frame: depth image in metres (0.5 mm noise, 600 mm background), model,
    intrinsics, colour-to-depth offset, heat map, parameters, and -- like run.py's reader.background, the empty scene
    in front of the camera (datareader.py:313-320, :716-718) -- the back plane without the object, one point per pixel.
    Returns (chain, depth_m, heat, init_fn)."""
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
    background = None
    if with_background:
        z_bg = 600.0 + np.random.default_rng(5).normal(0.0, 0.5, t_hit.shape)
        background = PointCloud(f.dirs / f.dirs[:, 2:3] * z_bg[:, None])
    chain = FrameChain(f.model_points, f.normals, f.tris, intr, f.K.astype(np.float32), color_to_depth, params,
                       background=background)

    def init_pose():
        init = synth.start_pose()
        init[2, 3] += 5.0
        return init

    return chain, depth_m, heat, init_pose
