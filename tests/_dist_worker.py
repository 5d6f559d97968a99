"""Worker of tests/test_dist_gloo.py: one rank of a world_size-2 gloo group on the CPU.
The compute backend is a TEST-LOCAL stand-in built on the oracle (the product backend is
HIP-only); what is under test is pedp_hip.dist's sharding, padding, all-gather reassembly and
the per-pass packet all-reduce protocol."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


class OracleBackend:
    def __init__(self):
        import pedp_oracle
        import torch

        self.o, self.torch = pedp_oracle, torch

    def make_mesh(self, v, t):
        return (v, t)

    def cast(self, mesh, rays6):
        r = self.o.raycast(mesh[0], mesh[1], rays6, nthreads=2)
        return r["t_hit"], r["primitive_ids"]

    def hit_records_tensor(self, t_hit, ids):
        rec = np.empty((len(t_hit), 2), np.uint32)
        rec[:, 0] = t_hit.view(np.uint32)
        rec[:, 1] = ids
        return self.torch.from_numpy(rec.view(np.int32))

    def make_cloud(self, points, normals=None):
        return (np.asarray(points, np.float64).reshape(-1, 3), normals)

    def packet_tensor(self, ptr, n):
        return self.torch.from_numpy(ptr)      # "device pointer" = the numpy packet itself

    def icp_batched(self, src, tgt, radius, inits, estimator, max_iteration):
        """pedp_icp_batched's contract: B independent registrations, no early exit."""
        rs = [self.o.icp(src[0], tgt[0], tgt[1], radius, T0, max_iter=max_iteration, rel_fitness=-1, rel_rmse=-1)
              for T0 in inits]
        return (np.stack([r["T"] for r in rs]), np.array([r["fitness"] for r in rs]),
                np.array([r["inlier_rmse"] for r in rs]))

    def icp(self, src, tgt, radius, init, estimator, max_iteration, rel_fitness, rel_rmse, allreduce, n_global):
        """Per-pass protocol of pedp_icp: local packet (29 doubles, same layout) -> hook ->
        every rank solves the same system."""
        o = self.o
        pts, (model, normals) = src[0], tgt
        T = np.array(init, dtype=np.float64)
        P = o.transform(T, pts) if len(pts) else pts
        prev = None
        for p in range(max_iteration + 1):
            packet = np.zeros(29)
            if len(pts):
                j, d2 = o.nn(P, model, kdtree=True, nthreads=2)
                m = d2 < radius * radius
                s, t, n = P[m], model[j[m]], normals[j[m]]
                J = np.hstack([np.cross(s, n), n])
                A = J.T @ J
                packet[:21] = A[np.triu_indices(6)]
                packet[21:27] = J.T @ ((s - t) * n).sum(1)
                packet[27], packet[28] = d2[m].sum(), m.sum()
            if allreduce is not None:
                allreduce(packet, 29, None)
            K = packet[28]
            fit, rmse = (K / n_global, np.sqrt(packet[27] / K)) if K > 0 else (0.0, 0.0)
            done = p >= max_iteration or (prev is not None and abs(prev[0] - fit) < rel_fitness and abs(prev[1] - rmse) < rel_rmse)
            prev = (fit, rmse)
            if done:
                return {"T": T, "fitness": fit, "inlier_rmse": rmse, "iters": p}
            A = np.zeros((6, 6))
            A[np.triu_indices(6)] = packet[:21]
            A = A + np.triu(A, 1).T
            ok, x = o.solve6(A, -packet[21:27])
            U = o.vec6_to_T(x) if (ok and K > 0) else np.eye(4)
            T = U @ T
            P = o.transform(U, P) if len(pts) else P


def main(rank, world, port, out_dir, hip=False):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0" if hip else str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from pedp_hip import dist as pdist
    from pedp_hip import synth

    r, w, _ = pdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    # hip: the PRODUCT backend, both ranks on GPU 0 (gloo moves the CUDA tensors; RCCL refuses two
    # ranks on one device) -- the multi-rank protocol on real kernels and streams
    be = pdist.HipBackend(0) if hip else OracleBackend()
    f = synth.Frame("tiny")
    rays = f.rays6[: 48 * 40 - 3]                       # ragged: 1917 rays over 2 ranks
    t_all, id_all = pdist.sharded_cast_rays(be, f.verts_posed, f.tris, rays)
    g = np.load(os.path.join(ROOT, "tests", "golden", "g3g4_icp_traces.npz"))
    res = pdist.sharded_registration_icp(be, g["scene_noisy"][:-5], g["model"], g["normals"], 10.0, g["init"],
                                         max_iteration=6, rel_fitness=-1, rel_rmse=-1)
    # 5 start poses over 2 ranks (3 + 2), one final all-gather
    rng = np.random.default_rng(3)
    inits = np.empty((5, 4, 4))
    for b in range(5):
        D = np.eye(4)
        D[:3, :3] = synth.axis_angle(rng.normal(size=3), np.deg2rad(2.0) * rng.uniform())
        D[:3, 3] = rng.uniform(-1, 1, 3)
        inits[b] = D @ g["init"]
    bT, bfit, brmse = pdist.sharded_icp_batched(be, g["scene_noisy"], g["model"], g["normals"], 10.0, inits, max_iteration=4)
    # device-resident sharded frame (bench.py's form) on the product backend
    extra = {}
    if hip:
        native = be.init_comm()          # gloo group: no RCCL possible with two ranks on one GPU -> stays False
        sf = pdist.ShardedFrame(be, f.verts_posed, f.tris, rays, g["scene_noisy"][:-5], g["model"], g["normals"])
        sf.cast()
        st, sid = sf.hits()
        sres = sf.icp(g["init"], 10.0, max_iteration=6, rel_fitness=-1, rel_rmse=-1)
        extra = dict(native=native, st=st, sid=sid, sT=sres["T"], sfit=sres["fitness"])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), t=t_all, ids=id_all, T=res["T"], fitness=res["fitness"],
             rmse=res["inlier_rmse"], iters=res["iters"], bT=bT, bfit=bfit, brmse=brmse, binits=inits, **extra)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], hip=len(sys.argv) > 5 and sys.argv[5] == "hip")
