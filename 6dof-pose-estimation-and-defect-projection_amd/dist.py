"""One process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI).

Sharding of the hot path:
  * rays are independent: contiguous row blocks per rank, triangle buffer replicated, one
    all-gather of 8-byte hit records (t_hit f32, primitive id u32) at the end;
  * ICP has one exchange per correspondence pass: scene points are sharded, the model is
    replicated, and the 29-double partial-sum packet is summed over ranks (all-reduce) before
    every rank solves the same 6x6 system -- so every rank holds the identical pose.
No collective is added anywhere else.  The compute calls go through a small backend object
so the same driver runs on gloo/CPU in tests (with a CPU stand-in supplied BY THE TEST);
the default and only product backend is the HIP library.
"""
import os

import numpy as np

from . import _lib


def shard_bounds(n, rank, world):
    """Contiguous block [lo, hi) of n items for `rank`; earlier ranks take the remainder."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).
    Returns (rank, world, local_rank).  world == 1 needs no process group."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class _DevicePacket:
    """Zero-copy torch view of a raw device pointer (the library's packet buffer)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class HipBackend:
    """Product backend: libpedp_hip.so on this rank's GPU.  The library's kernels and the
    collectives of torch.distributed share ONE explicit HIP stream (a torch.cuda.Stream the
    context is bound to): a collective enqueued inside `ordered()` waits for the kernels before
    it and holds back the kernels after it, with no host synchronisation.  (torch's default
    stream has handle 0, which the C ABI reads as "create your own non-blocking stream" -- work
    on such a stream is NOT ordered with torch's.)"""

    def __init__(self, device=None):
        import torch

        self.torch = torch
        self.device = torch.cuda.current_device() if device is None else device
        self.stream = torch.cuda.Stream(device=self.device)
        self.ctx = _lib.Context(self.device, stream=self.stream.cuda_stream)

    def ordered(self):
        """Context manager: torch work issued inside runs on the library's stream."""
        return self.torch.cuda.stream(self.stream)

    # ---- rays
    def make_mesh(self, vertices_f32, triangles):
        return _lib.Mesh(self.ctx, vertices_f32, triangles)

    def cast(self, mesh, rays6):
        """rays6: host float32 [n,6] -> (t_hit f32[n], ids u32[n]) host arrays."""
        r = mesh.cast_rays(rays6, want_uv=False)
        return r["t_hit"], r["primitive_ids"]

    def hit_records_tensor(self, t_hit, ids):
        rec = np.empty((len(t_hit), 2), np.uint32)
        rec[:, 0] = t_hit.view(np.uint32)
        rec[:, 1] = ids
        with self.ordered():
            return self.torch.from_numpy(rec.view(np.int32)).to(f"cuda:{self.device}")

    # ---- ICP
    def make_cloud(self, points, normals=None):
        return _lib.Cloud(self.ctx, points, normals)

    def icp(self, src, tgt, radius, init, estimator, max_iteration, rel_fitness, rel_rmse, allreduce,
            n_source_global):
        return _lib.icp(self.ctx, src, tgt, radius, init, estimator=estimator, max_iteration=max_iteration,
                        relative_fitness=rel_fitness, relative_rmse=rel_rmse, allreduce=allreduce,
                        n_source_global=n_source_global)

    def packet_tensor(self, ptr, n):
        return self.torch.as_tensor(_DevicePacket(ptr, n), device=f"cuda:{self.device}")


def _ordered(backend):
    """The backend's stream scope (HIP) or nothing (the CPU stand-ins of the gloo tests)."""
    import contextlib

    return backend.ordered() if hasattr(backend, "ordered") else contextlib.nullcontext()


def sharded_cast_rays(backend, vertices_f32, triangles, rays6, group=None, gather=True, always_collective=False):
    """Cast `rays6` (the SAME full array on every rank) with rows sharded over the ranks and
    all-gather the hit records.  Returns (t_hit, primitive_ids) for all rays (or only this
    rank's block when gather=False).  always_collective runs the all-gather even on one rank
    (exercises the stream ordering with RCCL on a single GPU)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = len(rays6)
    lo, hi = shard_bounds(n, rank, world)
    mesh = backend.make_mesh(vertices_f32, triangles)
    t_loc, id_loc = backend.cast(mesh, rays6[lo:hi])
    if (world == 1 and not always_collective) or not gather:
        return t_loc, id_loc
    # equal-size blocks for all_gather_into_tensor: pad to the largest shard
    width = shard_bounds(n, 0, world)[1]
    with _ordered(backend):
        rec = backend.hit_records_tensor(t_loc, id_loc)
        if rec.shape[0] < width:
            pad = torch.zeros((width - rec.shape[0], 2), dtype=rec.dtype, device=rec.device)
            rec = torch.cat([rec, pad])
        out = torch.empty((world * width, 2), dtype=rec.dtype, device=rec.device)
        dist.all_gather_into_tensor(out, rec.contiguous(), group=group)
        out = out.cpu().numpy().view(np.uint32).reshape(world, width, 2)
    t_all = np.empty(n, np.float32)
    id_all = np.empty(n, np.uint32)
    for r in range(world):
        a, b = shard_bounds(n, r, world)
        t_all[a:b] = out[r, : b - a, 0].view(np.float32)
        id_all[a:b] = out[r, : b - a, 1]
    return t_all, id_all


def sharded_registration_icp(backend, source_points, target_points, target_normals, radius, init,
                             estimator=_lib.POINT_TO_PLANE, max_iteration=30, rel_fitness=1e-6, rel_rmse=1e-6,
                             group=None, always_collective=False):
    """registration_icp with the scene sharded over the ranks (every rank passes the SAME
    full arrays and takes its block).  All ranks return the identical result dict."""
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = len(source_points)
    lo, hi = shard_bounds(n, rank, world)
    src = backend.make_cloud(np.asarray(source_points)[lo:hi])
    tgt = backend.make_cloud(target_points, target_normals)

    def allreduce(ptr, count, stream):
        # called between the reduce and the solve kernel of a pass, which are on `stream`
        with _ordered(backend):
            t = backend.packet_tensor(ptr, count)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    hook = allreduce if (world > 1 or always_collective) else None
    return backend.icp(src, tgt, radius, init, estimator, max_iteration, rel_fitness, rel_rmse, hook, n)
