"""bench.py -- the hot path on synthetic 640x576 frames against a 100k-triangle mesh.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config bench_100k] [--mode shard|replica]
                    [--no-cpu-baseline] [--no-extras]

`--gpus N` without a torchrun environment starts the N ranks itself (fresh child processes,
spawned before anything touches the GPU; non-zero exit if a rank fails).  Under
`python -m torch.distributed.run ... bench.py --gpus N` the ranks come from the environment.

One STEP = one frame of the hot path with inputs resident in HBM:
    ICP refinement   20 point-to-plane iterations (early exit disabled), 368,640 scene points
                     against the 50,000 model vertices                         (pedp_icp)
    ray projection   368,640 camera rays against all 100,000 triangles          (pedp_raycast)
The two stages are independent (scene cloud vs mesh): each runs on its own context and HIP
stream and they overlap on the device; a step ends when both have finished.

N > 1, --mode shard (default; SURVEY s8e, north_star): ONE frame is split over the ranks -- rays in contiguous blocks,
triangle records replicated, an all-gather of the 8-byte hit records (RCCL over xGMI, issued on the library's stream).
Strong scaling.  The workload of the N > 1 headline is the one the split is built for, BASELINE config 4: the
1280x720 dense frame (921,600 rays) against the 1M-triangle mesh, `value` = its rays / the slowest rank's time per
sharded cast (`config.workload` says so; the same cast on ONE GPU is timed in the same run, `one_gpu_same_workload`,
so the line carries its own strong-scaling ratio -- the N = 1 line of this file stays on the metric's own 100k
configuration, as the contract asks).  The camera-size frame split the same way is timed beside it ("frame_100k":
a 0.05-ms ray stage and a chain of 21 dependent 33-us passes -- neither pays for a collective, the registration is
replicated on every rank below a million scene points and every rank gets the identical pose; the scene-sharded
registration with one 29-double all-reduce per pass is timed under "icp_scene_sharded"), the pose batch of BASELINE
config 3 sharded over the ranks under "icp_batched", and the rate of the same ranks each on its own whole frame
(no collective, weak scaling) under "replica".
--mode replica makes that last one the headline instead (value = frames of all ranks x rays / time).

Timed regions, each bracketed by barrier + synchronize, max over ranks:
    headline     K steps of the default path (triangle-driven ray stage, chunked ICP: one launch per pass)
                 on ONE resident frame                                              -> value
    fresh_frame  (N = 1) K steps in which everything a new camera frame brings is inside the step: a new scene
                 cloud handle from a device array with new noise (copy, box, spatial order, chunk spheres),
                 a new mesh pose (records rebuilt on the device), then the same ICP + cast   -> "fresh_frame"
    frame_chain  (N = 1) K frames of BASELINE config 5's geometry chain (pedp_hip.frame_chain: depth filters,
                 back-projection, preprocess_source, z search, randomised restarts, posed mesh, heat-map
                 projection, viewer message), a new depth image per frame              -> "frame_chain"
    exhaustive   a few steps of the all-pairs path north_star's targets are set on: every ray x
                 every triangle (sweep variant 1) and every scene point x every model point in
                 every ICP pass (pedp_icp_configure(exhaustive))                    -> "exhaustive"
Every `roofline*` entry describes a kernel AS IT RUNS in the region it names, timed there with
the library's HIP events on the launch stream; kernel_ms x launches_per_step <= that region's
ms_per_step.  `cpu_baseline`: the CPU oracle (BVH rays + KD-tree ICP built once per
registration, radius-bounded search) on this host's cores, median of 10 steps after 2 warm-ups,
rank 0 at N = 1 only.
"""
import argparse
import csv
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: vector == f32-MFMA dense peak
FLOP_PER_PAIR_BF16 = 64    # executed by the dense NN sweep on the bf16 pipe: 16 x 16 x 32 x 2 flop per 256 pairs
PEAK_HBM_GBS = 8000.0
FLOP_PER_TEST = 46        # Moeller-Trumbore with stored (v0, e1, e2)
FLOP_PER_TEST_MFMA = 192     # matrix-pipe filter (round 4): 3 x v_mfma_f32_16x16x32_bf16 (16 x 16 x 32 x 2 flop) per 256 tests
FLOP_PER_TEST_EXECUTED_R03 = 19  # round 3's packed fp32 loop (variant 5): three dot products (5 flop each), one slack fma, one min3
PEAK_BF16_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak (~2.5 PFLOP/s)
FLOP_PER_PAIR = 8         # one K=4 fp32 MFMA dot per (scene, model) pair
ICP_ITERS = 20
TIMED_PASS = -3           # headline: ONE HIP-event pair around all 21 launches of the pass kernel: span / 21
TIMED_PASS_EX = -2        # exhaustive region: events around the sweep kernel of every fourth pass (1, 5, 9, 13, 17): their mean


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="bench_100k")
    ap.add_argument("--mode", choices=["shard", "replica"], default="shard")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the exhaustive / batched / 1M-triangle regions")
    return ap.parse_args()


def self_launch(args):
    """Start one fresh process per GPU.  Nothing in this process has touched the GPU (torch is
    not even imported), the children are ordinary subprocesses, and the exit code is non-zero
    if any rank fails."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = set(range(len(procs)))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"[bench] rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                    for q in pending:
                        procs[q].terminate()   # exact children of this process, by handle
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def cpu_baseline(frame, depth):
    """The oracle as CPU baseline ("port"): BVH closest hit (build included, like the
    reference's per-call add_triangles) + KD-tree point-to-plane ICP (tree built once per
    registration like registration_icp, search bounded by the correspondence radius), all host
    cores, whole steps of the same workload; median of 10 after 2 warm-ups."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pedp_oracle as oracle

    scene = frame.scene(depth)
    # threads: os.cpu_count() counts the host's cores, not this container's share of them, and an
    # oversubscribed OpenMP team is many times slower -- take the fastest of a few team sizes on a
    # short registration and report that count as `cores`
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    trial = {}
    for n in sorted({min(avail, n) for n in (8, 16, 32, 64, 128, avail)}):
        t0 = time.perf_counter()
        oracle.icp(scene, frame.model_points, frame.normals, frame.max_correspondence_distance, frame.icp_init(),
                   max_iter=3, rel_fitness=-1, rel_rmse=-1, kdtree=True, nthreads=n, want_trace=False)
        trial[n] = time.perf_counter() - t0
    cores = min(trial, key=trial.get)
    ray_s, icp_s = [], []
    for k in range(12):
        t0 = time.perf_counter()
        oracle.raycast(frame.verts_posed, frame.tris, frame.rays6, nthreads=cores, bvh=True)
        t1 = time.perf_counter()
        oracle.icp(scene, frame.model_points, frame.normals, frame.max_correspondence_distance, frame.icp_init(),
                   max_iter=ICP_ITERS, rel_fitness=-1, rel_rmse=-1, kdtree=True, nthreads=cores, want_trace=False)
        t2 = time.perf_counter()
        if k >= 2:
            ray_s.append(t1 - t0)
            icp_s.append(t2 - t1)
        if t2 - t0 > 15.0 and k >= 4:   # a slow host: stay within the sample budget
            break
    step = float(np.median(np.array(ray_s) + np.array(icp_s)))
    return {
        "value": frame.n_rays / step / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": f"median of {len(ray_s)} whole steps after 2 warm-ups: BVH build + cast of {frame.n_rays} rays x "
                  f"{frame.n_tris} tris (median {np.median(ray_s):.3f} s) + {ICP_ITERS}-iteration KD-tree ICP, tree built "
                  f"once per registration, search bounded by the radius (median {np.median(icp_s):.3f} s), OpenMP x{cores} "
                  f"(fastest of team sizes {sorted(trial)} on {avail} visible cores)",
        "ray_mrays_per_s": frame.n_rays / float(np.median(ray_s)) / 1e6,
        "icp_iters_per_s": ICP_ITERS / float(np.median(icp_s)),
        "note": "restatement of the Open3D/Embree algorithms (oracle/), not Open3D itself: context, not a target",
    }


def run(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    from pedp_hip import _lib, synth
    from pedp_hip import dist as pdist

    # PEDP_BENCH_REHEARSAL=1: every rank on GPU 0 with gloo as the transport -- the N > 1 code path on a
    # one-GPU box (RCCL refuses two ranks per device).  A rehearsal, never a measurement: the line says so.
    rehearsal = os.environ.get("PEDP_BENCH_REHEARSAL") == "1"
    rank, world, local = pdist.init_from_env("gloo" if rehearsal else "nccl")
    if rehearsal:
        local = 0
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for the wrong N",
                  file=sys.stderr)
        return 2
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    mode = "single" if world == 1 else args.mode
    # Explicit HIP streams shared by the library's kernels AND the collectives.  The two stages of
    # a step are independent (ICP works on the scene cloud, the ray stage on the mesh), so each
    # gets its own context and stream and they overlap on the device.
    icp_be = pdist.HipBackend(local)
    ray_be = pdist.HipBackend(local)
    ctx, ray_ctx = icp_be.ctx, ray_be.ctx
    # The collectives of the one-frame split go through torch.distributed (backend nccl = RCCL) on the library's streams
    # unless PEDP_BENCH_NATIVE_RCCL=1 asks for the library's own communicators: those have only ever run on one-rank
    # groups (no multi-GPU node was available to this build), and a first multi-rank run belongs in a test, not between
    # the driver and its scaling curve.
    native = False
    if world > 1 and not rehearsal and os.environ.get("PEDP_BENCH_NATIVE_RCCL") == "1":
        native = bool(icp_be.init_comm()) & bool(ray_be.init_comm())

    frame = synth.Frame(args.config)
    n_rays, n_tris = frame.n_rays, frame.n_tris
    radius = frame.max_correspondence_distance
    init = frame.icp_init()

    # ---- whole frame resident on this rank (single / replica steps, rendering of the depth frame)
    mesh = _lib.Mesh(ray_ctx, frame.verts_posed, frame.tris)
    rays = ray_be.to_device(frame.rays6)
    t_hit = torch.empty(n_rays, dtype=torch.float32, device=dev)
    prim = torch.empty(n_rays, dtype=torch.int32, device=dev)

    # the camera's rays are a resident ray set (pedp_rayset_*): the grid of the triangle-driven ray stage is built once,
    # a cast is the triangle kernels and the result kernel.  PEDP_BENCH_PLAIN_CAST=1: pedp_raycast on the ray array per call
    rayset = None if os.environ.get("PEDP_BENCH_PLAIN_CAST") == "1" else _lib.RaySet(ray_ctx, device_ptr=rays.data_ptr(), n=n_rays)
    plain_cast = [False]

    def cast_full():
        if rayset is not None and not plain_cast[0]:
            mesh.cast_rayset_device(rayset, t_hit.data_ptr(), prim.data_ptr())
        else:
            mesh.cast_rays_device(rays.data_ptr(), n_rays, t_hit.data_ptr(), prim.data_ptr())

    cast_full()
    ray_ctx.synchronize()
    depth = t_hit.cpu().numpy()
    scene = frame.scene(depth)  # rendered by the HIP ray caster; noise from default_rng(0)
    n_scene, n_model = len(scene), len(frame.model_points)
    src = _lib.Cloud(ctx, scene)
    tgt = _lib.Cloud(ctx, frame.model_points, frame.normals)
    icp_kw = dict(estimator=_lib.POINT_TO_PLANE, max_iteration=ICP_ITERS, relative_fitness=-1.0, relative_rmse=-1.0)

    # ---- the same frame sharded over the ranks (strong scaling)
    sharded = None
    if world > 1:   # built in both modes: the headline of one, a timed region beside the headline of the other
        sharded = pdist.ShardedFrame(ray_be, frame.verts_posed, frame.tris, frame.rays6, scene, frame.model_points,
                                     frame.normals, icp_backend=icp_be)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    icp_wall = [0.0]

    def step_whole():
        """Whole frame on this rank (N = 1, and the replica region)."""
        a = time.perf_counter()
        _lib.icp_begin(ctx, src, tgt, radius, init, **icp_kw)   # every pass enqueued on the ICP stream, returns at once
        cast_full()                                # enqueued on the ray stream while the first passes run
        res = _lib.icp_end(ctx)                    # returns after the ICP stream has finished
        icp_wall[0] = 1e3 * (time.perf_counter() - a)   # (the registration's wall time, the cast's enqueue inside it)
        ray_ctx.synchronize()                      # the step ends when both stages have finished
        return res

    replicate_icp = n_scene < 1_000_000   # one registration of a camera-size scene: every rank runs it whole

    def step_sharded(force_scene_shards=False):
        sharded.cast()                             # this rank's ray block + all-gather, on the ray stream
        a = time.perf_counter()
        if replicate_icp and not force_scene_shards:
            res = _lib.icp(ctx, src, tgt, radius, init, **icp_kw)
        else:
            res = sharded.icp(init, radius, max_iteration=ICP_ITERS, rel_fitness=-1.0, rel_rmse=-1.0)
        icp_wall[0] = 1e3 * (time.perf_counter() - a)
        ray_ctx.synchronize()
        return res

    def timed_region(step, steps, warmup):
        """W warm-ups, then K steps between barrier + synchronize; returns (seconds max over ranks,
        last result, per-step [icp wall ms, ray stage ms, timed sweep kernel ms])."""
        for _ in range(warmup):
            step()
        rows = []
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step()
            # read-backs of HIP events already recorded on the streams (both streams are idle here)
            rows.append((icp_wall[0], _lib.raycast_last_sweep_ms(ray_ctx), _lib.nn_last_sweep_ms(ctx)))
        barrier()
        return max_over_ranks(time.perf_counter() - t0), res, np.array(rows)

    def step_serial():
        """The same step with the two stages one after the other: the exhaustive region quotes
        kernel rooflines, so its kernels must not share the chip with each other."""
        if mode == "shard":
            sharded.cast()
            ray_ctx.synchronize()
            return sharded.icp(init, radius, max_iteration=ICP_ITERS, rel_fitness=-1.0, rel_rmse=-1.0)
        cast_full()
        ray_ctx.synchronize()
        return _lib.icp(ctx, src, tgt, radius, init, **icp_kw)

    step = step_sharded if mode == "shard" else step_whole
    _lib.icp_configure(ctx, exhaustive=False, timed_pass=TIMED_PASS)
    elapsed, res, rows = timed_region(step, args.steps, args.warmup)
    passes, pairs_swept, fb_points = _lib.icp_last_stats(ctx)
    ms_per_step = 1e3 * elapsed / args.steps
    # which ray stage answered the timed casts
    ray_variant, grid_status = rayset.last_variant() if (rayset is not None and mode != "shard") else _lib.raycast_last_variant(ray_ctx)

    extras = {}
    # ---- N > 1, shard mode: the headline workload is BASELINE config 4 -- the 1280x720 dense frame against the 1M-triangle
    # mesh, rays in contiguous blocks per rank + all-gather of the hit records; K timed casts after W warm-ups
    big_headline = None
    if mode == "shard" and args.config == "bench_100k":
        big = synth.Frame("bench_1m")
        bf = pdist.ShardedFrame(ray_be, big.verts_posed, big.tris, big.rays6)
        for _ in range(max(args.warmup, 1)):
            bf.cast()
            ray_ctx.synchronize()
        big_rows = []
        barrier()
        tb = time.perf_counter()
        for _ in range(args.steps):
            bf.cast()
            ray_ctx.synchronize()
            big_rows.append(_lib.raycast_last_sweep_ms(ray_ctx))
        barrier()
        big_elapsed = max_over_ranks(time.perf_counter() - tb)
        big_variant = _lib.raycast_last_variant(ray_ctx)
        t_all, prim_all = bf.hits()                       # host arrays of the whole frame, from the gathered records
        # the same cast whole on this rank's GPU (no split, no collective): the line's own one-GPU reference
        whole = _lib.Mesh(ray_ctx, big.verts_posed, big.tris)
        rays_w = ray_be.to_device(big.rays6)
        t_w = torch.empty(big.n_rays, dtype=torch.float32, device=dev)
        p_w = torch.empty(big.n_rays, dtype=torch.int32, device=dev)
        for _ in range(2):
            whole.cast_rays_device(rays_w.data_ptr(), big.n_rays, t_w.data_ptr(), p_w.data_ptr())
        ray_ctx.synchronize()
        barrier()
        tw = time.perf_counter()
        for _ in range(args.steps):
            whole.cast_rays_device(rays_w.data_ptr(), big.n_rays, t_w.data_ptr(), p_w.data_ptr())
            ray_ctx.synchronize()
        one_gpu = max_over_ranks(time.perf_counter() - tw)
        same = bool(np.array_equal(t_all.view(np.uint32), t_w.cpu().numpy().view(np.uint32))
                    and np.array_equal(prim_all, p_w.cpu().numpy().view(np.uint32)))
        big_headline = {"rays": big.n_rays, "tris": big.n_tris, "elapsed": big_elapsed, "sweep_ms": float(np.mean(big_rows)),
                        "variant": big_variant, "one_gpu_elapsed": one_gpu, "equals_one_gpu": same,
                        "width": big.width, "height": big.height}
        del bf, big, whole, rays_w, t_w, p_w, t_all, prim_all
    if not args.no_extras:
        # ---- exhaustive region: the all-pairs path, same resident inputs, same sharding
        ex_steps = max(2, min(args.steps, 4))
        plain_cast[0] = True                      # (the exhaustive variants are pedp_raycast's)
        _lib.raycast_configure(ray_ctx, 0, 1)
        _lib.icp_configure(ctx, exhaustive=True, timed_pass=TIMED_PASS_EX)
        try:
            ex_elapsed, ex_res, ex_rows = timed_region(step_serial, ex_steps, 1)
            ex_passes, ex_pairs, _ = _lib.icp_last_stats(ctx)
        finally:
            _lib.raycast_configure(ray_ctx, 0, 0)
            _lib.icp_configure(ctx, exhaustive=False, timed_pass=TIMED_PASS)
        extras["exhaustive"] = (ex_steps, ex_elapsed, ex_res, ex_rows, ex_passes, ex_pairs)
        if mode != "shard":   # round 3's packed fp32 loop on the same rays, for the side field
            _lib.raycast_configure(ray_ctx, 0, 5)
            try:
                v5 = []
                for _ in range(3):
                    cast_full()
                    ray_ctx.synchronize()
                    v5.append(_lib.raycast_last_sweep_ms(ray_ctx))
                extras["sweep_variant5_ms"] = float(np.median(v5))
            finally:
                _lib.raycast_configure(ray_ctx, 0, 0)
        if mode != "shard":   # the per-call form of the headline's ray stage (pedp_raycast: rays read twice, chains rebuilt), for the side field
            pc = []
            for _ in range(5):
                cast_full()
                ray_ctx.synchronize()
                pc.append(_lib.raycast_last_sweep_ms(ray_ctx))
            extras["plain_cast_ms"] = float(np.median(pc[1:]))
        plain_cast[0] = False
        # ---- replica region (N > 1): every rank its own whole frame, no collective
        if mode == "shard":
            rp_elapsed, _, rp_rows = timed_region(step_whole, args.steps, 1)
            extras["replica"] = (rp_elapsed, rp_rows)
        elif mode == "replica":   # ---- ONE frame split over the ranks (rays in blocks + all-gather), beside the headline
            fs_elapsed, fs_res, fs_rows = timed_region(step_sharded, args.steps, 1)
            extras["frame_sharded"] = (fs_elapsed, fs_res, fs_rows)
        if world > 1 and replicate_icp:   # the scene-sharded registration (one all-reduce per pass) beside the replicated one
            ss_elapsed, ss_res, ss_rows = timed_region(lambda: step_sharded(True), max(2, min(args.steps, 4)), 1)
            extras["scene_sharded"] = (max(2, min(args.steps, 4)), ss_elapsed, ss_res, ss_rows)
        if world == 1:
            # ---- fresh_frame region: what a NEW camera frame costs -- scene handle from a device array with new
            # noise, new mesh pose, then the same ICP + cast
            base_dev = torch.from_numpy(scene).to(dev)
            noise_dev = [torch.from_numpy(np.random.default_rng(100 + k).normal(0.0, 0.05, scene.shape)).to(dev) for k in range(4)]
            poses = []
            for k in range(4):
                D = np.eye(4)
                D[:3, :3] = synth.axis_angle(np.array([0.0, 0.0, 1.0]), np.deg2rad(0.01 * k))
                poses.append(frame.T_gt @ D)
            pmesh = _lib.Mesh(ray_ctx, frame.model_points, frame.tris, posable=True)
            fresh_wall = []

            def step_fresh():
                k = len(fresh_wall)
                a = time.perf_counter()
                pts = base_dev + noise_dev[k % 4]                 # the frame's new points (torch's stream) ...
                torch.cuda.current_stream().synchronize()         # ... are complete before the library reads them
                pmesh.set_pose(poses[k % 4])                      # records rebuilt on the device (ray stream)
                if rayset is not None:
                    pmesh.cast_rayset_device(rayset, t_hit.data_ptr(), prim.data_ptr())
                else:
                    pmesh.cast_rays_device(rays.data_ptr(), n_rays, t_hit.data_ptr(), prim.data_ptr())
                b = time.perf_counter()
                s_k = _lib.Cloud.from_device(ctx, pts.data_ptr(), len(pts))   # copy, box, then (inside pedp_icp) order + spheres
                r_k = _lib.icp(ctx, s_k, tgt, radius, init, **icp_kw)
                c = time.perf_counter()
                ray_ctx.synchronize()
                s_k.close()                                       # buffers back to the context's pool
                icp_wall[0] = 1e3 * (c - b)
                fresh_wall.append((1e3 * (b - a), 1e3 * (c - b)))
                return r_k

            fr_elapsed, fr_res, _ = timed_region(step_fresh, args.steps, 2)
            extras["fresh_frame"] = (fr_elapsed, fr_res, np.array(fresh_wall[-args.steps:]), _lib.raycast_last_variant(ray_ctx))
            del pmesh, base_dev, noise_dev
            # ---- frame_chain / tracking_frame regions: BASELINE config 5's geometry chain, a new depth image per frame,
            # both branches of run.py's loop with the arguments run.py passes: the reader's background cloud
            # (run.py:99-101, :154-156) and the root logger at INFO in the reference's format (run.py:252, :260;
            # Utils.py:94-99) -- the lines are formatted and written like the reference's, to a null sink
            import logging
            from pedp_hip import viewer_wire
            from pedp_hip.frame_chain import bench_frame_setup
            chain, depth_m, heat, init_pose = bench_frame_setup(frame, depth)
            viewer_wire.attach_queues(viewer_wire.LatestQueue())   # a consumer that keeps up (an unread queue.Queue would keep every frame's mesh alive)
            root_log = logging.getLogger()
            log_level, log_sink = root_log.level, logging.StreamHandler(open(os.devnull, "w"))
            log_sink.setFormatter(logging.Formatter("[%(funcName)s()] %(message)s"))
            root_log.addHandler(log_sink)
            root_log.setLevel(logging.INFO)
            drng = np.random.default_rng(7)
            depth_k = [(depth_m + drng.normal(0.0, 2e-4, depth_m.shape)).astype(np.float32) for _ in range(4)]
            n_done = [0]

            frame_wall = []

            def step_chain():
                a = time.perf_counter()
                out_k = chain.process(depth_k[n_done[0] % 4], init_pose(), heat, seed=n_done[0])
                frame_wall.append(1e3 * (time.perf_counter() - a))
                n_done[0] += 1
                return {"T": out_k["icp"].transformation, "fitness": out_k["icp"].fitness, "n_hits": len(out_k["cloud"].points)}

            fc_steps = max(3, min(args.steps, 10))
            fc_elapsed, fc_res, _ = timed_region(step_chain, fc_steps, 2)
            fc_wall = list(frame_wall[-fc_steps:])
            chain.process(depth_k[0], init_pose(), heat, seed=0, timed=True)     # stage times, outside the timed region
            stage_ms = dict(chain.stage_ms)
            n_done[0] = 0                                                         # the same frames, the scene cloud never on the host

            def step_chain_dev():
                out_k = chain.process(depth_k[n_done[0] % 4], init_pose(), heat, seed=n_done[0], device_scene=True)
                n_done[0] += 1
                return {"T": out_k["icp"].transformation}

            fd_elapsed, fd_res, _ = timed_region(step_chain_dev, fc_steps, 2)
            # run.py:132-207: every step a tracking frame with a defect detection pending -- preprocess_source(i > 0),
            # improve_result from the tracker's bare 4x4, delta_pose, posed mesh, projection of a new heat map, the earlier
            # hit clouds moved by relative_transformation (two detections are kept: the list of run.py:61 grows with the
            # run; the bench holds it at frame 0's cloud plus the last one), update_dash_data
            out0 = chain.process(depth_k[0], init_pose(), heat, seed=0)          # frame 0 of this run (run.py:79-131)
            tracker_pose = np.linalg.inv(out0["icp"].transformation)
            tracker_pose[:3, 3] += (0.8, -0.5, 1.0)                               # est.track_one's estimate: close, not equal
            heat_t = np.zeros_like(heat)
            heat_t[150:330, 260:460] = np.linspace(0.76, 1.0, 200)[None, :]
            n_done[0] = 1

            def step_tracking():
                del chain.intersection_pcds[1:]
                a = time.perf_counter()
                out_k = chain.process_tracking(depth_k[n_done[0] % 4], tracker_pose.copy(), heat_t, i=n_done[0], seed=n_done[0])
                frame_wall.append(1e3 * (time.perf_counter() - a))
                n_done[0] += 1
                return {"T": out_k["icp"].transformation, "fitness": out_k["icp"].fitness, "n_hits": len(out_k["cloud"].points),
                        "n_processed": len(out_k["source_processed"].points)}

            tf_elapsed, tf_res, _ = timed_region(step_tracking, fc_steps, 2)
            tf_wall = list(frame_wall[-fc_steps:])
            chain.process_tracking(depth_k[0], tracker_pose.copy(), heat_t, i=n_done[0], seed=0, timed=True)
            extras["frame_chain"] = (fc_steps, fc_elapsed, fc_res, stage_ms, fd_elapsed, fc_wall)
            extras["tracking_frame"] = (fc_steps, tf_elapsed, tf_res, dict(chain.stage_ms), tf_wall)
            viewer_wire.attach_queues(None)
            root_log.removeHandler(log_sink)
            root_log.setLevel(log_level)
        # ---- BASELINE config 3: 256 start poses refined concurrently (groups of 32 share launches), poses sharded over the ranks
        inits = np.stack([np.linalg.inv(T) for T in synth.batched_start_poses(256)])
        lo, hi = pdist.shard_bounds(len(inits), rank, world)
        batch_s = []
        for _ in range(3):
            barrier()
            tb = time.perf_counter()
            if hi > lo:
                _lib.icp_batched(ctx, src, tgt, radius, inits[lo:hi], max_iteration=ICP_ITERS)
            barrier()
            batch_s.append(max_over_ranks(time.perf_counter() - tb))
        extras["batched"] = (len(inits), min(batch_s))
        # ---- BASELINE config 4: 1M-triangle mesh, 1280x720 dense frame, rays sharded + all-gather
        if args.config == "bench_100k" and big_headline is None:
            big = synth.Frame("bench_1m")
            bf = pdist.ShardedFrame(ray_be, big.verts_posed, big.tris, big.rays6)
            for _ in range(2):
                bf.cast()
            barrier()
            tb = time.perf_counter()
            for _ in range(5):
                bf.cast()
                ray_ctx.synchronize()
            barrier()
            extras["bench_1m"] = (big.n_rays, big.n_tris, max_over_ranks(time.perf_counter() - tb) / 5,
                                  _lib.raycast_last_sweep_ms(ray_ctx))
            del bf, big

    if rank == 0:
        traffic = {}
        traffic_file = None
        for name in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):   # PMC passes committed under profiles/
            try:
                with open(os.path.join(ROOT, "profiles", name)) as fh:
                    traffic = json.load(fh)
                traffic_file = "profiles/" + name
                break
            except OSError:
                pass
        traffic_source = (f"{traffic_file}: separate rocprofv3 --pmc passes over tools/pmc_target.py, committed -- NOT measured in this "
                          "run (the live figures of this line are the HIP-event kernel times)") if traffic_file else None
        trace_us = None     # the same kernel in the committed rocprofv3 kernel trace of this command
        import glob, re
        stats = [p for p in glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats_v*.csv"))
                 if re.search(r"r(\d+)_bench_kernel_stats_v(\d+)\.csv$", p)]
        stats.sort(key=lambda p: tuple(int(g) for g in re.search(r"r(\d+)_bench_kernel_stats_v(\d+)\.csv$", p).groups()))
        for name in stats[-1:]:   # the newest round / version committed
            try:
                with open(name) as fh:
                    for row in csv.DictReader(fh):
                        if "icp_pass_kernel<8, false>" in row["Name"]:
                            trace_us = float(row["AverageNs"]) / 1e3
            except (OSError, KeyError, ValueError):
                pass
        icp_ms, ray_ms, sweep_ms = (float(v) for v in rows.mean(axis=0))
        n_pass = max(passes, 1)
        pairs_pass = pairs_swept / n_pass
        sweep_tflops = FLOP_PER_PAIR * pairs_pass / (sweep_ms * 1e-3) / 1e12
        par = {"single": "one GPU; ray stage and ICP of a step overlap on two HIP streams",
               "shard": f"one frame sharded over {world} GPUs: contiguous ray blocks + all-gather of hit records "
                        f"({'library-issued RCCL' if native else 'torch.distributed'}); the frame's one registration "
                        + ("replicated on every rank (a camera-size scene: passes too short to pay for an all-reduce each; the "
                           "scene-sharded variant is timed under icp_scene_sharded, the pose batch under icp_batched)"
                           if replicate_icp else "with scene shards + one 29-double all-reduce per pass"),
               "replica": f"{world} GPUs, each its own whole frame, no collective"}[mode]
        frames_per_step = world if mode == "replica" else 1
        out = {
            "metric": "Mrays/s ray-mesh + ICP iters/s, 640x576 vs 100k-tri mesh",
            "value": frames_per_step * n_rays * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak" if mode == "replica" else "strong", "vs_baseline": None,
            "dtype": "f32 rays / f64 ICP (f32 MFMA filter)", "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo, not a measurement)" if rehearsal else ""),
            "config": {"workload": f"{args.config}: {frame.width}x{frame.height} frame, {n_rays} rays x {n_tris} "
                                   f"triangles + {ICP_ITERS}-iteration point-to-plane ICP ({n_scene} scene x "
                                   f"{n_model} model points) per step",
                       "parallelism": par, "mode": mode},
            "ray_stage_ms": ray_ms, "ray_stage_mrays_per_s": (n_rays / world if mode == "shard" else n_rays) / (ray_ms * 1e-3) / 1e6,
            "icp_ms": icp_ms, "icp_iters_per_s": ICP_ITERS / (icp_ms * 1e-3),
            "icp_passes": passes, "icp_pairs_swept_per_pass": pairs_pass,
            "icp_fallback_points_per_pass": fb_points / n_pass,
            "icp_fitness": res["fitness"], "icp_inlier_rmse": res["inlier_rmse"],
            "pose_error_vs_gt": float(np.abs(np.linalg.inv(res["T"]) - frame.T_gt).max()),
            # dominant kernel of the HEADLINE step as it runs there (rank 0's share in shard mode)
            "ray_variant": ray_variant, "ray_grid_status": grid_status,
            "ray_stage_form": "resident ray set (pedp_raycast_rayset: grid chains built once; 4 launches per cast)" if rayset is not None
                              else "pedp_raycast per call (rays read twice, chains rebuilt; 6 launches)",
            "ray_stage_ms_per_call_form": extras.get("plain_cast_ms"),
            "roofline": {"kernel": "icp_pass_kernel (one launch per pass: per-chunk transform, culling, MFMA sweep, selection, partial "
                                   "sums; the last workgroup sums, solves and updates the pose)", "region": "headline", "bound": "mfma",
                         "achieved": sweep_tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": sweep_tflops / PEAK_FP32_TFLOPS,
                         "traffic": traffic.get("icp_pass_kernel", {}).get("hbm_bytes_per_launch"),
                         "traffic_source": traffic_source,
                         "mfma_util_pmc_committed_profile": traffic.get("icp_pass_kernel", {}).get("mfma_util"),
                         "kernel_ms": sweep_ms, "launches_per_step": passes,
                         "kernel_ms_x_launches": sweep_ms * passes, "region_ms_per_step": ms_per_step,
                         "kernel_us_committed_kernel_trace": trace_us,
                         "note": f"{FLOP_PER_PAIR} flop x {pairs_pass:.4g} (scene slot, model point) pairs the kernel swept per "
                                 "pass (16 x 16 per MFMA, mean over the passes) / mean launch-to-launch time of the kernel: ONE "
                                 "HIP-event pair around the 21 launches of every registration of the timed loop, span / 21, "
                                 "launch boundaries included (kernel_us_committed_kernel_trace is its mean dispatch-to-completion "
                                 "span in the rocprofv3 kernel trace under profiles/).  The launch is "
                                 "the WHOLE pass (the finish kernel of round 2 is inside it); it is one workgroup per live scene "
                                 "chunk and is bound by its chain of dependent memory accesses and by the serial close of the "
                                 "pass (DESIGN s4.2), not by the matrix pipe: the MFMA fraction says how little of it is arithmetic"},
        }
        if "exhaustive" in extras:
            ex_steps, ex_elapsed, ex_res, ex_rows, ex_passes, ex_pairs = extras["exhaustive"]
            ex_ms = 1e3 * ex_elapsed / ex_steps
            _, ex_ray_ms, ex_sweep_ms = (float(v) for v in ex_rows.mean(axis=0))
            share = (1.0 / world) if mode == "shard" else 1.0
            tests = float(n_rays) * n_tris * share                      # this rank's ray block x all triangles
            pairs = float(n_scene) * n_model * share                    # this rank's scene shard x all model points
            nn_tflops = FLOP_PER_PAIR * pairs / (ex_sweep_ms * 1e-3) / 1e12
            ray_exec = FLOP_PER_TEST_MFMA * tests / (ex_ray_ms * 1e-3) / 1e12
            ray_algo = FLOP_PER_TEST * tests / (ex_ray_ms * 1e-3) / 1e12
            stream_bytes = -(-int(n_rays * share) // 64) * n_tris * 36.0
            hbm_frac = stream_bytes / (ex_ray_ms * 1e-3) / 1e9 / PEAK_HBM_GBS
            out["exhaustive"] = {
                "steps": ex_steps, "ms_per_step": ex_ms, "value_mrays_per_s": n_rays * ex_steps / ex_elapsed / 1e6,
                "ray_stage_ms": ex_ray_ms, "icp_iters_per_s": ICP_ITERS * ex_steps / ex_elapsed,
                "icp_passes": ex_passes, "icp_pairs_swept_per_pass": ex_pairs / max(ex_passes, 1),
                "pose_equals_headline": bool(np.abs(ex_res["T"] - res["T"]).max() < 1e-9),
                "note": "every ray x every triangle (sweep variant 1), then every scene point x every model point in every "
                        "ICP pass (culling off), one stage after the other; same results as the headline path",
            }
            nn_bf16 = os.environ.get("PEDP_NN_F32") != "1"     # the dense sweep's form (csrc/pedp_icp.hip: nn_bf16_sweep)
            nn_key = "nn_sweep_bf16_kernel" if nn_bf16 else "nn_sweep_kernel"
            nn_exec = (FLOP_PER_PAIR_BF16 if nn_bf16 else FLOP_PER_PAIR) * pairs / (ex_sweep_ms * 1e-3) / 1e12
            nn_peak = PEAK_BF16_TFLOPS if nn_bf16 else PEAK_FP32_TFLOPS
            out["roofline_exhaustive_nn"] = {
                "kernel": ("nn_sweep_bf16_kernel<4,2> (all pairs: one v_mfma_f32_16x16x32_bf16 per 16 x 16 pairs over exact three-way bf16 pieces)"
                           if nn_bf16 else "nn_sweep_kernel<4,2> (all pairs, f32-input MFMA)"), "region": "exhaustive", "bound": "mfma",
                "achieved": nn_exec, "peak": nn_peak, "unit": "TFLOP/s", "frac": nn_exec / nn_peak,
                "traffic": traffic.get(nn_key, {}).get("hbm_bytes_per_launch"),
                "traffic_source": traffic_source,
                "mfma_util_pmc_committed_profile": traffic.get(nn_key, {}).get("mfma_util"),
                "kernel_ms": ex_sweep_ms, "launches_per_step": ex_passes, "kernel_ms_x_launches": ex_sweep_ms * ex_passes,
                "region_ms_per_step": ex_ms,
                "algorithmic_8flop_tflops": nn_tflops, "algorithmic_8flop_frac_of_fp32_mfma_peak": nn_tflops / PEAK_FP32_TFLOPS,
                "note": (f"frac counts the {FLOP_PER_PAIR_BF16} bf16 flop per pair the kernel EXECUTES (K = 32: 27 products of the coordinates' "
                         "pieces + 3 of |t'|^2's, exact; the fp32 accumulation inside the pipe is covered by the slot's error bound, "
                         "winners are re-scored in float64) against the dense bf16 peak; SURVEY s8d's 8 flop per pair (one K = 4 dot) "
                         f"x {int(n_scene * share)} scene x {n_model} model points per launch beside it: the f32-input form of rounds 1-3 "
                         "(PEDP_NN_F32=1) ran 1.69 ms = 0.555 of the fp32-MFMA peak, this one 0.84 ms with the same bits in every result"
                         if nn_bf16 else
                         f"{FLOP_PER_PAIR} flop x {int(n_scene * share)} scene x {n_model} model points per launch (SURVEY s8d)")}
            out["roofline_ray_sweep"] = {
                "kernel": "ray_sweep_mfma_kernel<8> (every ray x every triangle: bf16 MFMA filter, exact test on the pairs that pass)",
                "region": "exhaustive", "bound": "mfma",
                "achieved": ray_exec, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ray_exec / PEAK_BF16_TFLOPS,
                "traffic": traffic.get("ray_sweep_mfma_kernel", {}).get("hbm_bytes_per_launch"), "traffic_source": traffic_source,
                "mfma_busy_pmc_committed_profile": traffic.get("ray_sweep_mfma_kernel", {}).get("mfma_busy"),
                "kernel_ms": ex_ray_ms, "launches_per_step": 1, "kernel_ms_x_launches": ex_ray_ms,
                "region_ms_per_step": ex_ms, "mrays_per_s": n_rays * share / (ex_ray_ms * 1e-3) / 1e6,
                "algorithmic_46flop_tflops": ray_algo, "algorithmic_46flop_frac_of_fp32_vector_peak": ray_algo / PEAK_FP32_TFLOPS,
                "packed_fp32_loop_ms": extras.get("sweep_variant5_ms"),
                "packed_fp32_loop_frac_of_fp32_vector_peak": (None if not extras.get("sweep_variant5_ms") else
                    FLOP_PER_TEST_EXECUTED_R03 * tests / (extras["sweep_variant5_ms"] * 1e-3) / 1e12 / PEAK_FP32_TFLOPS),
                "north_star_hbm_stream_gbs": stream_bytes / (ex_ray_ms * 1e-3) / 1e9, "north_star_hbm_stream_frac": hbm_frac,
                "north_star_hbm_target_met": bool(hbm_frac >= 0.5),
                "note": f"frac counts the {FLOP_PER_TEST_MFMA} bf16 flop per test the filter EXECUTES on the matrix pipe (three "
                        "v_mfma_f32_16x16x32_bf16 per 16 rays x 16 triangles: the three edge scores as K = 32 contractions over exact "
                        "three-way bf16 splits and a slack slot) against the dense bf16 MFMA peak; the exact Moeller-Trumbore test runs "
                        f"on the pairs that pass (1.001x the accepted pairs).  The {FLOP_PER_TEST}-flop algorithmic figure of SURVEY s8d "
                        "against the fp32 vector peak and round 3's packed fp32 loop (variant 5, timed in the same region) are side "
                        "fields.  north_star's >= 50 % of the HBM roofline reads in its triangle-stream accounting "
                        "(ceil(N_r/64) x N_f x 36 B per launch over 8 TB/s): north_star_hbm_stream_frac"}
        rast = traffic.get("ray_stage_rayset" if rayset is not None else "ray_stage_rast")
        ray_bytes = 24.0 * n_rays + 48.0 * n_tris + 8.0 * n_rays       # rays in, triangle records in, (t, id) out
        if mode != "shard":
            out["roofline_ray_stage"] = {
                "kernel": ("triangle-driven ray stage against the resident ray set (rast_tri / rast_item / rast_full / ray_finalize)" if rayset is not None
                           else "triangle-driven ray stage (rast_bounds / rast_insert / rast_tri / rast_item / rast_full / ray_finalize)") if ray_variant == 4
                          else f"ray stage, variant {ray_variant}", "region": "headline", "bound": "hbm",
                "achieved": ray_bytes / (ray_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": ray_bytes / (ray_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "traffic": None if not rast else rast.get("hbm_bytes_per_cast"), "traffic_source": traffic_source,
                "kernel_ms": ray_ms, "launches_per_step": 1, "kernel_ms_x_launches": ray_ms, "region_ms_per_step": ms_per_step,
                "pmc": rast,
                "note": "algorithmic bytes of one cast = 24 B per ray in + 48 B per triangle record in + 8 B per ray out "
                        f"= {ray_bytes / 1e6:.1f} MB (the per-call accounting, kept for comparison: against a resident ray set the rays are "
                        "not read at all), over the HIP-event span of the cast's kernels.  A few short launches: the "
                        "stage is bound by launch boundaries and by the chains of dependent accesses inside them (cell "
                        "heads -> nodes -> atomicMin keys), not by HBM; see DESIGN s4.1 for the PMC reading"}
        if "fresh_frame" in extras:
            fr_elapsed, fr_res, fr_wall, fr_variant = extras["fresh_frame"]
            out["fresh_frame"] = {
                "ms_per_frame": 1e3 * fr_elapsed / args.steps, "value_mrays_per_s": n_rays * args.steps / fr_elapsed / 1e6,
                "new_points_pose_cast_enqueue_ms": float(fr_wall[:, 0].mean()), "cloud_order_icp_ms": float(fr_wall[:, 1].mean()),
                "icp_iters_per_s": ICP_ITERS / (float(fr_wall[:, 1].mean()) * 1e-3),
                "ray_variant": fr_variant[0], "ray_grid_status": fr_variant[1],
                "icp_fitness": fr_res["fitness"], "pose_error_vs_gt": float(np.abs(np.linalg.inv(fr_res["T"]) - frame.T_gt).max()),
                "note": "every step: new scene points on the device (base + fresh noise), pedp_cloud_create_device (copy, box), "
                        "spatial order + chunk spheres of the new handle, pedp_mesh_set_pose with a new pose (records rebuilt), "
                        f"{ICP_ITERS}-iteration registration, full-frame cast; the handle's buffers go back to the pool"}
        if "frame_chain" in extras:
            fc_steps, fc_elapsed, fc_res, fc_stage, fd_elapsed, fc_wall = extras["frame_chain"]
            out["frame_chain"] = {
                "frames": fc_steps, "ms_per_frame": 1e3 * fc_elapsed / fc_steps, "frames_per_s": fc_steps / fc_elapsed,
                "ms_slowest_frame": max(fc_wall),   # FrameChain freezes the collector's old generations: no 40-ms full collection lands on a frame (DESIGN s6)
                "ms_per_frame_device_scene": 1e3 * fd_elapsed / fc_steps,   # PointCloud over the CUDA tensor: no 9 MB down and up again
                "stage_ms": fc_stage, "icp_fitness": fc_res["fitness"], "projected_hits": fc_res["n_hits"],
                "pose_error_vs_gt": float(np.abs(np.linalg.inv(fc_res["T"]) - frame.T_gt).max()),
                "arguments": "run.py's: reader.background passed (368,640-point empty scene), root logger at INFO",
                "note": "BASELINE config 5, geometry part, run.py:79-131 (pedp_hip.frame_chain; tests/test_stream_gpu.py checks "
                        "every stage against the oracle): a new 640x576 depth image per frame -> depth filters -> back-projection "
                        "-> preprocess_source -> z search -> randomised ICP restarts -> posed mesh -> heat-map projection -> "
                        "viewer message.  stage_ms from one extra, synchronised frame outside the timed region"}
        if "tracking_frame" in extras:
            tf_steps, tf_elapsed, tf_res, tf_stage, tf_wall = extras["tracking_frame"]
            out["tracking_frame"] = {
                "frames": tf_steps, "ms_per_frame": 1e3 * tf_elapsed / tf_steps, "frames_per_s": tf_steps / tf_elapsed,
                "ms_slowest_frame": max(tf_wall),
                "stage_ms": tf_stage, "icp_fitness": tf_res["fitness"], "projected_hits": tf_res["n_hits"],
                "processed_points": tf_res["n_processed"],
                "pose_error_vs_gt": float(np.abs(np.linalg.inv(tf_res["T"]) - frame.T_gt).max()),
                "arguments": "run.py's: reader.background passed, root logger at INFO, parameters mutated to down_sample = 5",
                "note": "run.py:132-207, a tracking frame with a defect detection pending: depth filters -> back-projection -> "
                        "preprocess_source(i > 0) -> improve_result from the tracker's bare 4x4 -> delta_pose -> posed mesh -> "
                        "projection of a new heat map -> earlier hit clouds moved by relative_transformation -> viewer message"}
        if "scene_sharded" in extras:
            ss_steps, ss_elapsed, ss_res, ss_rows = extras["scene_sharded"]
            out["icp_scene_sharded"] = {
                "ms_per_step": 1e3 * ss_elapsed / ss_steps, "icp_iters_per_s": ICP_ITERS / (float(ss_rows[:, 0].mean()) * 1e-3),
                "pose_equals_headline": bool(np.abs(ss_res["T"] - res["T"]).max() < 1e-9),
                "note": "the same step with the registration's scene points sharded over the ranks and one 29-double all-reduce "
                        "per pass (SURVEY s8e row 2): the right shape for scenes of millions of points, slower than the "
                        "replicated registration at camera size"}
        if "frame_sharded" in extras:
            fs_elapsed, fs_res, fs_rows = extras["frame_sharded"]
            out["frame_sharded"] = {
                "value_mrays_per_s": n_rays * args.steps / fs_elapsed / 1e6, "scaling": "strong", "ms_per_step": 1e3 * fs_elapsed / args.steps,
                "ray_stage_ms": float(fs_rows[:, 1].mean()), "icp_ms": float(fs_rows[:, 0].mean()),
                "pose_equals_headline": bool(np.abs(fs_res["T"] - res["T"]).max() < 1e-9),
                "note": f"ONE frame split over the {world} ranks: rays in contiguous blocks + all-gather of the hit records "
                        f"({'library-issued RCCL' if native else 'torch.distributed'}), the registration replicated (camera-size scene); "
                        "not the headline: a 0.05-ms ray stage does not pay for a collective"}
        if "replica" in extras:
            rp_elapsed, rp_rows = extras["replica"]
            out["replica"] = {"value_mrays_per_s": world * n_rays * args.steps / rp_elapsed / 1e6, "scaling": "weak",
                              "ms_per_step": 1e3 * rp_elapsed / args.steps,
                              "note": "every rank its own whole frame, no collective (not the headline)"}
        if "batched" in extras:
            nb, sec = extras["batched"]
            out["icp_batched"] = {"poses": nb, "ms_per_registration": 1e3 * sec / nb,
                                  "iters_per_s": ICP_ITERS * nb / sec,
                                  "note": f"pedp_icp_batched, poses in contiguous blocks over {world} rank(s)"}
        if "bench_1m" in extras:
            br, bt, bsec, bsweep = extras["bench_1m"]
            out["bench_1m"] = {"workload": f"{br} rays x {bt} triangles (BASELINE config 4), ray stage only",
                               "ms_per_frame": 1e3 * bsec, "mrays_per_s": br / bsec / 1e6, "sweep_ms_rank0": bsweep,
                               "parallelism": f"rays in {world} contiguous block(s)" + (" + all-gather" if world > 1 else "")}
        if big_headline is not None:
            # the N > 1 headline: BASELINE config 4's cast, sharded; what the 100k frame did in the same split moves aside
            bh = big_headline
            out["frame_100k"] = {k: out[k] for k in ("value", "ms_per_step", "ray_stage_ms", "ray_stage_mrays_per_s", "icp_ms",
                                                      "icp_iters_per_s", "icp_passes", "icp_fitness", "pose_error_vs_gt", "ray_variant",
                                                      "ray_grid_status", "roofline")}
            out["frame_100k"]["workload"] = out["config"]["workload"]
            out["frame_100k"]["note"] = ("the camera-size frame in the same split (rays in blocks + all-gather, the registration "
                                         "replicated): a 0.05-ms ray stage does not pay for a collective")
            for k in ("ray_stage_ms", "ray_stage_mrays_per_s", "icp_ms", "icp_iters_per_s", "icp_passes", "icp_pairs_swept_per_pass",
                      "icp_fallback_points_per_pass", "icp_fitness", "icp_inlier_rmse", "pose_error_vs_gt"):
                out.pop(k, None)
            ms_big = 1e3 * bh["elapsed"] / args.steps
            out["value"] = bh["rays"] * args.steps / bh["elapsed"] / 1e6
            out["ms_per_step"] = ms_big
            out["metric"] = "Mrays/s ray-mesh, 1280x720 dense frame vs 1M-tri mesh (BASELINE config 4), rays sharded + all-gather"
            out["dtype"] = "f32"
            out["config"]["workload"] = (f"bench_1m (BASELINE config 4): {bh['width']}x{bh['height']} dense frame, {bh['rays']} rays x "
                                         f"{bh['tris']} triangles per step, ray stage (the split's workload; the 100k frame under frame_100k)")
            out["ray_variant"], out["ray_grid_status"] = bh["variant"]
            out["hits_equal_one_gpu_cast"] = bh["equals_one_gpu"]
            out["one_gpu_same_workload"] = {"ms_per_step": 1e3 * bh["one_gpu_elapsed"] / args.steps,
                                            "value_mrays_per_s": bh["rays"] * args.steps / bh["one_gpu_elapsed"] / 1e6,
                                            "note": "the whole cast on one of these GPUs, no split, no collective, same run"}
            out["strong_scaling_speedup"] = bh["one_gpu_elapsed"] / bh["elapsed"]
            algo = 24.0 * bh["rays"] / world + 48.0 * bh["tris"] + 8.0 * bh["rays"] / world
            gbs = algo / (bh["sweep_ms"] * 1e-3) / 1e9
            out["roofline"] = {"kernel": "triangle-driven ray stage (rast_* kernels of one cast) on this rank's ray block", "region": "headline",
                               "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                               "traffic": None, "kernel_ms": bh["sweep_ms"], "launches_per_step": 1, "kernel_ms_x_launches": bh["sweep_ms"],
                               "region_ms_per_step": ms_big,
                               "note": "algorithmic bytes of rank 0's share of a cast (24 B per ray in, 48 B per triangle record, 8 B per "
                                       "ray out) / the HIP-event span of the cast's kernels on the ray stream; the rest of a step is the "
                                       "all-gather of the hit records"}
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(frame, depth)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    return run(args)


if __name__ == "__main__":
    sys.exit(main())
