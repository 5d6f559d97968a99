"""One process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI).

Sharding of the hot path:
  * rays are independent: contiguous row blocks per rank, triangle buffer replicated, one
    all-gather of 8-byte hit records (t_hit f32, primitive id u32) at the end;
  * ICP has one exchange per correspondence pass: scene points are sharded, the model is
    replicated, and the 29-double partial-sum packet is summed over ranks (all-reduce) before
    every rank solves the same 6x6 system -- so every rank holds the identical pose;
  * batched pose hypotheses (FoundationPose sizing, estimater.py:104-122) are independent: the
    poses are split into contiguous blocks per rank, every rank refines its block with
    pedp_icp_batched, one final all-gather of B x (4x4 + fitness + rmse) float64;
  * improve_result's restarts depend on each other (RNG order, compounding threshold): replicas
    only, no function here.
No collective is added anywhere else.  The compute calls go through a small backend object
so the same driver runs on gloo/CPU in tests (with a CPU stand-in supplied BY THE TEST);
the default and only product backend is the HIP library.

Collectives: the product backend gives its context an RCCL communicator of its own
(pedp_comm_create; the unique id travels through torch.distributed's rendezvous), so the
all-gather and the per-pass all-reduce are issued by the library on its stream -- no Python
between the kernels of a registration.  If that communicator cannot be made on every rank
(all ranks agree through one all-reduce), torch.distributed's collectives on the same stream
are used instead; the CPU stand-ins of the tests always take that route.
"""
import os

import numpy as np

from . import _lib


def shard_bounds(n, rank, world):
    """Contiguous block [lo, hi) of n items for `rank`; earlier ranks take the remainder."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_width(n, world):
    """Length of the largest block (the padded width of an equal-size all-gather)."""
    return shard_bounds(n, 0, world)[1] if world > 0 else n


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


def _world_rank(group=None):
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


class _DevicePacket:
    """Zero-copy torch view of a raw device pointer (the library's packet buffer)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class HipBackend:
    """Product backend: libpedp_hip.so on this rank's GPU.  The library's kernels and the
    collectives share ONE explicit HIP stream (a torch.cuda.Stream the context is bound to):
    a collective enqueued on it waits for the kernels before it and holds back the kernels after
    it, with no host synchronisation.  (torch's default stream has handle 0, which the C ABI
    reads as "create your own non-blocking stream" -- work on such a stream is NOT ordered with
    torch's.)"""

    def __init__(self, device=None, stream=None):
        import torch

        self.torch = torch
        self.device = torch.cuda.current_device() if device is None else device
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
        self.ctx = _lib.Context(self.device, stream=self.stream.cuda_stream)
        self.native = False  # the context owns an RCCL communicator

    def ordered(self):
        """Context manager: torch work issued inside runs on the library's stream."""
        return self.torch.cuda.stream(self.stream)

    def init_comm(self, group=None, force=False):
        """Give the context its own RCCL communicator over the ranks of `group`.  Collective.
        Returns True when every rank has one (native collectives from then on), False when any
        rank failed (torch.distributed's collectives stay in use).  force=True also builds a
        one-rank communicator (tests of the native path on a single GPU)."""
        import torch.distributed as dist

        world, rank = _world_rank(group)
        if self.native:
            return True
        if world == 1:
            if force:
                _lib.comm_create(self.ctx, _lib.comm_unique_id(), 1, 0)
                self.native = True
            return self.native
        # Two steps, because ncclCommInitRank is itself a collective: a rank that failed BEFORE it (librccl does not
        # load, a symbol is missing) would leave the others waiting inside it.  (1) every rank probes locally -- making
        # a unique id needs the library and nothing else -- and the ranks agree on the outcome through one all-reduce;
        # (2) only if every probe succeeded do all of them enter the communicator's creation.
        box, ok = [None], 1
        try:
            probe = _lib.comm_unique_id()
        except _lib.PedpError:
            probe, ok = b"", 0
        if rank == 0:
            box[0] = probe
        flag = self.torch.tensor([ok], dtype=self.torch.int32, device=f"cuda:{self.device}")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) != 1:
            return False
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        try:
            _lib.comm_create(self.ctx, box[0], world, rank)
        except _lib.PedpError:
            ok = 0
        flag = self.torch.tensor([ok], dtype=self.torch.int32, device=f"cuda:{self.device}")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 1:
            self.native = True
        elif ok:
            _lib.comm_destroy(self.ctx)
        return self.native

    # ---- collectives on device memory, on the library's stream
    def all_gather_device(self, send, recv, group=None):
        """send: contiguous CUDA tensor, recv: world x send bytes.  Only enqueues."""
        import torch.distributed as dist

        if self.native:
            _lib.comm_allgather(self.ctx, send.data_ptr(), recv.data_ptr(), send.numel() * send.element_size())
        else:
            with self.ordered():  # flat views: gloo checks the shapes literally
                dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)

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
        use_comm = allreduce is not None and self.native
        return _lib.icp(self.ctx, src, tgt, radius, init, estimator=estimator, max_iteration=max_iteration,
                        relative_fitness=rel_fitness, relative_rmse=rel_rmse,
                        allreduce=None if use_comm else allreduce, n_source_global=n_source_global, use_comm=use_comm)

    def icp_batched(self, src, tgt, radius, inits, estimator, max_iteration):
        return _lib.icp_batched(self.ctx, src, tgt, radius, inits, estimator=estimator, max_iteration=max_iteration)

    def packet_tensor(self, ptr, n):
        return self.torch.as_tensor(_DevicePacket(ptr, n), device=f"cuda:{self.device}")

    def to_device(self, array):
        with self.ordered():
            return self.torch.from_numpy(np.ascontiguousarray(array)).to(f"cuda:{self.device}")


def _ordered(backend):
    """The backend's stream scope (HIP) or nothing (the CPU stand-ins of the gloo tests)."""
    import contextlib

    return backend.ordered() if hasattr(backend, "ordered") else contextlib.nullcontext()


def _gather_rows(backend, rows, world, group):
    """All-gather equal-size row blocks (a torch tensor per rank) -> numpy [world, rows, cols]."""
    import torch
    import torch.distributed as dist

    out = torch.empty((world * rows.shape[0],) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    if getattr(backend, "native", False):
        backend.all_gather_device(rows.contiguous(), out, group)
        with _ordered(backend):
            host = out.cpu()
    else:
        with _ordered(backend):
            dist.all_gather_into_tensor(out.view(-1), rows.contiguous().view(-1), group=group)
            host = out.cpu()
    return host.numpy().reshape((world,) + tuple(rows.shape))


def sharded_cast_rays(backend, vertices_f32, triangles, rays6, group=None, gather=True, always_collective=False):
    """Cast `rays6` (the SAME full array on every rank) with rows sharded over the ranks and
    all-gather the hit records.  Returns (t_hit, primitive_ids) for all rays (or only this
    rank's block when gather=False).  always_collective runs the all-gather even on one rank
    (exercises the stream ordering with RCCL on a single GPU)."""
    import torch

    world, rank = _world_rank(group)
    n = len(rays6)
    lo, hi = shard_bounds(n, rank, world)
    mesh = backend.make_mesh(vertices_f32, triangles)
    t_loc, id_loc = backend.cast(mesh, rays6[lo:hi])
    if (world == 1 and not always_collective) or not gather:
        return t_loc, id_loc
    # equal-size blocks for the all-gather: pad to the largest shard
    width = shard_width(n, world)
    with _ordered(backend):
        rec = backend.hit_records_tensor(t_loc, id_loc)
        if rec.shape[0] < width:
            pad = torch.zeros((width - rec.shape[0], 2), dtype=rec.dtype, device=rec.device)
            rec = torch.cat([rec, pad])
    out = _gather_rows(backend, rec, world, group).view(np.uint32)
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

    world, rank = _world_rank(group)
    n = len(source_points)
    lo, hi = shard_bounds(n, rank, world)
    src = backend.make_cloud(np.asarray(source_points)[lo:hi])
    tgt = backend.make_cloud(target_points, target_normals)

    def allreduce(ptr, count, stream):
        # called between the reduce and the solve kernel of a pass, which are on `stream`
        # (only when the backend has no communicator of its own)
        with _ordered(backend):
            t = backend.packet_tensor(ptr, count)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    hook = allreduce if (world > 1 or always_collective) else None
    return backend.icp(src, tgt, radius, init, estimator, max_iteration, rel_fitness, rel_rmse, hook, n)


def sharded_icp_batched(backend, source_points, target_points, target_normals, radius, inits,
                        estimator=_lib.POINT_TO_PLANE, max_iteration=30, group=None, always_collective=False):
    """B start poses (B x 4 x 4, the SAME array on every rank) share one scene
    and one model; rank r refines the contiguous block shard_bounds(B, r, world) with
    pedp_icp_batched (no exchange inside a registration) and one all-gather of
    B x (16 + 2) float64 hands every rank all results.  Returns (T [B,4,4], fitness [B], rmse [B]),
    identical on every rank and equal to the single-GPU batched call bit for bit."""
    import torch

    world, rank = _world_rank(group)
    inits = np.ascontiguousarray(inits, dtype=np.float64).reshape(-1, 4, 4)
    B = len(inits)
    lo, hi = shard_bounds(B, rank, world)
    src = backend.make_cloud(source_points)
    tgt = backend.make_cloud(target_points, target_normals)
    rows = np.zeros((shard_width(B, world), 18), np.float64)
    if hi > lo:
        T, fit, rmse = backend.icp_batched(src, tgt, radius, inits[lo:hi], estimator, max_iteration)
        rows[: hi - lo, :16] = np.asarray(T).reshape(-1, 16)
        rows[: hi - lo, 16] = fit
        rows[: hi - lo, 17] = rmse
    if world == 1 and not always_collective:
        return rows[:B, :16].reshape(B, 4, 4).copy(), rows[:B, 16].copy(), rows[:B, 17].copy()
    dev = backend.to_device(rows) if hasattr(backend, "to_device") else torch.from_numpy(rows)
    out = _gather_rows(backend, dev, world, group)
    T_all = np.empty((B, 4, 4))
    fit_all, rmse_all = np.empty(B), np.empty(B)
    for r in range(world):
        a, b = shard_bounds(B, r, world)
        T_all[a:b] = out[r, : b - a, :16].reshape(-1, 4, 4)
        fit_all[a:b] = out[r, : b - a, 16]
        rmse_all[a:b] = out[r, : b - a, 17]
    return T_all, fit_all, rmse_all


class ShardedFrame:
    """Device-resident form of one sharded frame for repeated steps (bench.py, a camera loop):
    this rank's ray block, its hit-record buffer and the gathered records stay in HBM, the scene
    shard and the model are uploaded once.

        frame = ShardedFrame(backend, verts_f32, tris, rays6, scene, model, normals)
        frame.cast()      # sweep on this rank's block + all-gather of (t_hit | id) records
        frame.icp(init)   # scene-sharded registration, one packet all-reduce per pass

    Record layout: every rank owns 2 x width int32 -- row 0 the float32 bits of t_hit, row 1 the
    triangle ids -- so ONE all-gather moves both; `hits()` reassembles the frame on the host."""

    def __init__(self, backend, vertices_f32, triangles, rays6, scene=None, model=None, normals=None, group=None,
                 icp_backend=None):
        import torch

        self.be, self.group = backend, group
        self.icp_be = icp_backend or backend  # a second context / stream lets the two stages overlap
        self.world, self.rank = _world_rank(group)
        self.n_rays = len(rays6)
        self.lo, self.hi = shard_bounds(self.n_rays, self.rank, self.world)
        self.width = shard_width(self.n_rays, self.world)
        self.mesh = backend.make_mesh(vertices_f32, triangles)
        dev = f"cuda:{backend.device}"
        block = np.zeros((self.width, 6), np.float32)  # pad rays: zero direction, never hit
        block[: self.hi - self.lo] = np.asarray(rays6[self.lo:self.hi], np.float32)
        with backend.ordered():
            self.rays = torch.from_numpy(block).to(dev)
            self.rec = torch.empty((2, self.width), dtype=torch.int32, device=dev)
            self.gathered = torch.empty((self.world, 2, self.width), dtype=torch.int32, device=dev)
        # the rank's ray block as a resident ray set (the product backend: grid chains built once, four launches per cast)
        self.rayset = None
        if hasattr(backend, "ctx") and self.hi > self.lo:
            with backend.ordered():
                self.rayset = _lib.RaySet(backend.ctx, device_ptr=self.rays.data_ptr(), n=self.hi - self.lo)
        self.n_scene = 0
        if scene is not None:
            self.n_scene = len(scene)
            a, b = shard_bounds(self.n_scene, self.rank, self.world)
            self.src = self.icp_be.make_cloud(np.asarray(scene)[a:b])
            self.tgt = self.icp_be.make_cloud(model, normals)

    def cast(self, gather=True):
        """Enqueue the sweep of this rank's block and the all-gather of the records."""
        n = self.hi - self.lo
        if n > 0 and self.rayset is not None:
            self.mesh.cast_rayset_device(self.rayset, self.rec[0].data_ptr(), self.rec[1].data_ptr())
        elif n > 0:
            self.mesh.cast_rays_device(self.rays.data_ptr(), n, self.rec[0].data_ptr(), self.rec[1].data_ptr())
        if gather and self.world > 1:
            self.be.all_gather_device(self.rec, self.gathered, self.group)

    def hits(self):
        """(t_hit f32 [n_rays], primitive ids u32 [n_rays]) of the whole frame, on the host."""
        with self.be.ordered():
            g = (self.gathered if self.world > 1 else self.rec[None]).cpu().numpy().view(np.uint32)
        t_all, id_all = np.empty(self.n_rays, np.float32), np.empty(self.n_rays, np.uint32)
        for r in range(self.world):
            a, b = shard_bounds(self.n_rays, r, self.world)
            t_all[a:b] = g[r, 0, : b - a].view(np.float32)
            id_all[a:b] = g[r, 1, : b - a]
        return t_all, id_all

    def icp(self, init, radius, max_iteration=30, rel_fitness=1e-6, rel_rmse=1e-6, estimator=_lib.POINT_TO_PLANE):
        import torch.distributed as dist

        be, group = self.icp_be, self.group

        def allreduce(ptr, count, stream):
            with be.ordered():
                dist.all_reduce(be.packet_tensor(ptr, count), op=dist.ReduceOp.SUM, group=group)

        hook = allreduce if self.world > 1 else None
        return be.icp(self.src, self.tgt, radius, init, estimator, max_iteration, rel_fitness, rel_rmse, hook,
                      self.n_scene)
