"""bench.py -- the hot path on synthetic 640x576 frames against a 100k-triangle mesh.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config bench_100k] [--no-cpu-baseline]

One STEP = one frame of the hot path with inputs resident in HBM:
    ICP refinement   20 point-to-plane iterations (early exit disabled), 368,640 scene
                     points against the 50,000 model vertices                (pedp_icp)
    ray projection   368,640 camera rays against all 100,000 triangles       (pedp_raycast)
The two stages are independent (scene cloud vs mesh), so each runs on its own context and HIP
stream and they overlap on the device; a step ends when both have finished.
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), weak scaling -- every
rank processes its own frame, then the ranks all-gather their hit records (t_hit, id) on the
ray stream.

`value` is whole-job rays per second over the WHOLE step (ICP time included), i.e. frames/s
x rays per frame; the per-stage rates are reported next to it.  The JSON line also carries
`roofline` (the step's dominant kernel, the MFMA nearest-neighbour sweep, at the all-pairs
per-iteration workload), `roofline_ray_sweep` (the exhaustive every-ray-x-every-triangle
sweep, the kernel north_star's target is set on; algorithmic FLOPs per launch / HIP-event
kernel time), `roofline_hbm_stream` (north_star's triangle-stream accounting) and
`cpu_baseline` (the CPU oracle: BVH rays + KD-tree ICP on this host's cores, one full step).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: vector == f32-MFMA dense peak
PEAK_HBM_GBS = 8000.0
FLOP_PER_TEST = 46        # Moeller-Trumbore with stored (v0, e1, e2)
FLOP_PER_PAIR = 8         # one K=4 fp32 MFMA dot per (scene, model) pair
ICP_ITERS = 20


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="bench_100k")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(frame, depth):
    """The oracle as CPU baseline ("port"): BVH closest hit (build included, like the
    reference's per-call add_triangles) + KD-tree point-to-plane ICP, all host cores, one
    full step of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pedp_oracle as oracle

    cores = os.cpu_count() or 1
    scene = frame.scene(depth)
    t0 = time.perf_counter()
    oracle.raycast(frame.verts_posed, frame.tris, frame.rays6, nthreads=cores, bvh=True)
    t1 = time.perf_counter()
    oracle.icp(scene, frame.model_points, frame.normals, frame.max_correspondence_distance, frame.icp_init(),
               max_iter=ICP_ITERS, rel_fitness=-1, rel_rmse=-1, kdtree=True, nthreads=cores, want_trace=False)
    t2 = time.perf_counter()
    step = t2 - t0
    return {
        "value": frame.n_rays / step / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": f"1 full step: BVH build+cast of {frame.n_rays} rays x {frame.n_tris} tris ({t1 - t0:.2f} s) + "
                  f"{ICP_ITERS}-iteration KD-tree ICP ({t2 - t1:.2f} s), OpenMP x{cores}",
        "ray_mrays_per_s": frame.n_rays / (t1 - t0) / 1e6, "icp_iters_per_s": ICP_ITERS / (t2 - t1),
    }


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    from pedp_hip import _lib, synth
    from pedp_hip import dist as pdist

    rank, world, local = pdist.init_from_env("nccl")
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    # Explicit HIP streams shared by the library's kernels AND the collectives (torch's default
    # stream has handle 0 = "library creates its own stream", which NCCL would not order with).
    # The two stages of a step are independent (ICP works on the scene cloud, the ray stage on the
    # mesh), so each gets its own context and stream and they overlap on the device.
    stream = torch.cuda.Stream(device=dev)
    ctx = _lib.Context(local, stream=stream.cuda_stream)          # ICP
    ray_stream = torch.cuda.Stream(device=dev)
    ray_ctx = _lib.Context(local, stream=ray_stream.cuda_stream)  # ray stage + all-gather of its records

    frame = synth.Frame(args.config)
    n_rays, n_tris = frame.n_rays, frame.n_tris
    mesh = _lib.Mesh(ray_ctx, frame.verts_posed, frame.tris)
    rays = torch.from_numpy(frame.rays6).to(dev)
    t_hit = torch.empty(n_rays, dtype=torch.float32, device=dev)
    prim = torch.empty(n_rays, dtype=torch.int32, device=dev)

    def cast():
        mesh.cast_rays_device(rays.data_ptr(), n_rays, t_hit.data_ptr(), prim.data_ptr())

    cast()
    torch.cuda.synchronize()
    depth = t_hit.cpu().numpy()
    scene = frame.scene(depth)  # rendered by the HIP ray caster; noise from default_rng(0)
    src = _lib.Cloud(ctx, scene)
    tgt = _lib.Cloud(ctx, frame.model_points, frame.normals)
    init = frame.icp_init()
    gathered_t = torch.empty(world * n_rays, dtype=torch.float32, device=dev) if world > 1 else None
    gathered_i = torch.empty(world * n_rays, dtype=torch.int32, device=dev) if world > 1 else None

    sweep_ms, icp_ms = [], []

    def step(record):
        cast()                                     # enqueued on the ray stream, returns at once
        if world > 1:
            with torch.cuda.stream(ray_stream):    # ordered behind the sweep on the same stream
                dist.all_gather_into_tensor(gathered_t, t_hit)
                dist.all_gather_into_tensor(gathered_i, prim)
        a = time.perf_counter()
        res = _lib.icp(ctx, src, tgt, frame.max_correspondence_distance, init, estimator=_lib.POINT_TO_PLANE,
                       max_iteration=ICP_ITERS, relative_fitness=-1.0, relative_rmse=-1.0)
        b = time.perf_counter()
        ray_ctx.synchronize()                      # the step ends when both stages have finished
        if record:
            icp_ms.append(1e3 * (b - a))           # pedp_icp returns after its stream sync
            sweep_ms.append(_lib.raycast_last_sweep_ms(ray_ctx))  # HIP events around the sweep kernels
        return res

    for _ in range(args.warmup):
        step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- kernel-level measurements outside the timed region (same resident inputs) ----
    passes, pairs_swept, fb_points = _lib.icp_last_stats(ctx)
    # (a) the all-pairs MFMA sweep at the SURVEY's per-iteration workload: every scene point x
    #     every model point, no bounding-box culling (pedp_nn = one correspondence pass)
    nn_ms = []
    for _ in range(4):
        _lib.nn(ctx, src, tgt, init)
        nn_ms.append(_lib.nn_last_sweep_ms(ctx))
    # (b) the exhaustive ray sweep: every ray x every triangle (variant 1), same frame
    _lib.raycast_configure(ray_ctx, 0, 1)
    brute_ms = []
    for _ in range(4):
        cast()
        brute_ms.append(_lib.raycast_last_sweep_ms(ray_ctx))
    _lib.raycast_configure(ray_ctx, 0, 0)
    # (c) BASELINE config 3 in small: 32 start poses refined concurrently (pedp_icp_batched)
    inits = np.stack([np.linalg.inv(T) for T in synth.batched_start_poses(32)])
    batch_s = []
    for _ in range(3):
        tb = time.perf_counter()
        _lib.icp_batched(ctx, src, tgt, frame.max_correspondence_distance, inits, max_iteration=ICP_ITERS)
        batch_s.append(time.perf_counter() - tb)

    if rank == 0:
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE in separate runs, tools/pmc_target.py + tools/summarize_pmc.py)
        traffic = {}
        try:
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as fh:
                traffic = json.load(fh)
        except OSError:
            pass
        ms_per_step = 1e3 * elapsed / args.steps
        ray_stage = float(np.mean(sweep_ms))
        icp_mean = float(np.mean(icp_ms))
        tests = float(n_rays) * n_tris
        brute = float(np.median(brute_ms[1:]))
        stream_bytes = -(-n_rays // 64) * n_tris * 36.0
        nn = float(np.median(nn_ms[1:]))
        pairs = float(len(scene)) * len(frame.model_points)
        nn_tflops = FLOP_PER_PAIR * pairs / (nn * 1e-3) / 1e12
        out = {
            "metric": "Mrays/s ray-mesh + ICP iters/s, 640x576 vs 100k-tri mesh",
            "value": world * n_rays * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 rays / f64 ICP (f32 MFMA filter)", "data": "synthetic",
            "config": {"workload": f"{args.config}: {frame.width}x{frame.height} frame, {n_rays} rays x {n_tris} "
                                   f"triangles + {ICP_ITERS}-iteration point-to-plane ICP ({len(scene)} scene x "
                                   f"{len(frame.model_points)} model points) per step",
                       "parallelism": f"frames sharded over {world} GPU(s), all-gather of hit records; ray stage and ICP of a "
                                      "step overlap on two HIP streams"},
            "ray_stage_ms": ray_stage, "ray_stage_mrays_per_s": n_rays / (ray_stage * 1e-3) / 1e6,
            "icp_ms": icp_mean, "icp_iters_per_s": ICP_ITERS / (icp_mean * 1e-3),
            "icp_batched_ms_per_registration": 1e3 * min(batch_s) / len(inits),
            "icp_batched_iters_per_s": ICP_ITERS * len(inits) / min(batch_s),
            "icp_passes": passes, "icp_pairs_swept_per_pass": pairs_swept / max(passes, 1),
            "icp_fallback_points_per_pass": fb_points / max(passes, 1),
            "icp_fitness": res["fitness"], "icp_inlier_rmse": res["inlier_rmse"],
            "pose_error_vs_gt": float(np.abs(np.linalg.inv(res["T"]) - frame.T_gt).max()),
            # dominant kernel of the step by time: the MFMA nearest-neighbour sweep
            "roofline": {"kernel": "nn_sweep_kernel", "bound": "mfma", "achieved": nn_tflops, "peak": PEAK_FP32_TFLOPS,
                         "unit": "TFLOP/s", "frac": nn_tflops / PEAK_FP32_TFLOPS,
                         "traffic": traffic.get("nn_sweep_kernel", {}).get("hbm_bytes_per_launch"), "kernel_ms": nn,
                         "mfma_util_pmc": traffic.get("nn_sweep_kernel", {}).get("mfma_util"),
                         "note": f"{FLOP_PER_PAIR} flop x {len(scene)} scene x {len(frame.model_points)} model points "
                                 "(all pairs, one correspondence pass) / HIP-event duration of the sweep kernel; "
                                 "inside a step the same kernel runs on the bounding-box survivors only "
                                 "(icp_pairs_swept_per_pass)"},
            # the exhaustive ray sweep (variant 1: every ray x every triangle), SURVEY s8d accounting
            "roofline_ray_sweep": {"kernel": "ray_sweep_rpl_kernel<shared origin>", "bound": "valu_fp32",
                                   "achieved": FLOP_PER_TEST * tests / (brute * 1e-3) / 1e12, "peak": PEAK_FP32_TFLOPS,
                                   "unit": "TFLOP/s", "frac": FLOP_PER_TEST * tests / (brute * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                                   "traffic": traffic.get("ray_sweep_rpl_kernel", {}).get("hbm_bytes_per_launch"),
                                   "kernel_ms": brute, "mrays_per_s": n_rays / (brute * 1e-3) / 1e6,
                                   "executed_tflops": 21.0 * tests / (brute * 1e-3) / 1e12,
                                   "note": f"{FLOP_PER_TEST} algorithmic flop per test (SURVEY s8d); the shared-origin "
                                           "kernel hoists the origin-dependent terms per triangle and executes 21 "
                                           "flop per test, hence frac can exceed 1; the step itself uses the culled "
                                           "variant (ray_stage_ms)"},
            "roofline_hbm_stream": {"bound": "hbm", "achieved": stream_bytes / (brute * 1e-3) / 1e9,
                                    "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": stream_bytes / (brute * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                    "note": "north_star accounting on the exhaustive sweep: ceil(N_r/64) x N_f x 36 B "
                                            "triangle stream per launch; the records are L2-resident, real HBM traffic "
                                            "is about the compulsory 15 MB"},
        }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(frame, depth)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
