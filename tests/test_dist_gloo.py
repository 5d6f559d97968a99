"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): rays shard by contiguous
blocks + all-gather of hit records; ICP shards the scene and all-reduces the 29-double packet
once per pass.  Both must reproduce the single-process oracle result."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_sharding_matches_single_process(oracle, tmp_path):
    from pedp_hip import synth

    port = _free_port()
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), str(r), "2", str(port),
                               str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)

    f = synth.Frame("tiny")
    rays = f.rays6[: 48 * 40 - 3]
    ref = oracle.raycast(f.verts_posed, f.tris, rays)
    g = np.load(os.path.join(ROOT, "tests", "golden", "g3g4_icp_traces.npz"))
    ricp = oracle.icp(g["scene_noisy"][:-5], g["model"], g["normals"], 10.0, g["init"], max_iter=6, rel_fitness=-1,
                      rel_rmse=-1)
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(2))
    for r in (r0, r1):
        assert np.array_equal(r["ids"], ref["primitive_ids"])
        assert np.array_equal(r["t"].view(np.uint32), ref["t_hit"].view(np.uint32))
        assert float(r["fitness"]) == ricp["fitness"] and int(r["iters"]) == 6
        assert np.abs(r["T"] - ricp["T"]).max() < 1e-9 and abs(float(r["rmse"]) - ricp["inlier_rmse"]) < 1e-9
    assert np.array_equal(r0["T"], r1["T"])      # every rank holds the identical pose
    # batched-pose sharding: poses split 3 + 2, all-gather of 18 doubles per pose
    for b in range(5):
        one = oracle.icp(g["scene_noisy"], g["model"], g["normals"], 10.0, r0["binits"][b], max_iter=4, rel_fitness=-1,
                         rel_rmse=-1)
        for r in (r0, r1):
            assert np.array_equal(r["bT"][b], one["T"]) and r["bfit"][b] == one["fitness"]
            assert r["brmse"][b] == one["inlier_rmse"]


def test_shard_bounds_cover_and_pad():
    from pedp_hip import dist as pdist

    for n in (0, 1, 7, 8, 9, 1917, 368640):
        for world in (1, 2, 3, 8):
            blocks = [pdist.shard_bounds(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            assert max(b - a for a, b in blocks) == pdist.shard_width(n, world)
            assert max(b - a for a, b in blocks) - min(b - a for a, b in blocks) <= 1
