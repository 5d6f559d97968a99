"""GPU parity of the ICP path against the float64 oracle, through the C ABI
(pedp_nn / pedp_icp).  Tolerance (BASELINE.md s6 / north_star): refined 4x4 pose within
1e-5 abs; here the observed agreement is ~1e-10 and correspondence sets are identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-5


def _frame_scene(oracle, config):
    from pedp_hip import synth

    f = synth.Frame(config)
    depth = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)["t_hit"]
    return f, f.scene(depth)


@pytest.mark.parametrize("config", ["tiny", "parity"])
def test_nn_exact(ctx, oracle, config):
    from pedp_hip import _lib

    f, scene = _frame_scene(oracle, config)
    src = _lib.Cloud(ctx, scene)
    tgt = _lib.Cloud(ctx, f.model_points, f.normals)
    T = f.icp_init()
    idx, d2 = _lib.nn(ctx, src, tgt, T)
    ridx, rd2 = oracle.nn(oracle.transform(T, scene), f.model_points, kdtree=True)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(d2.view(np.uint64), rd2.view(np.uint64))  # same formula, same bits


def test_nn_adversarial_ties(ctx, oracle):
    """Many target points at (near-)equal distance: the fp32 filter cannot decide and the
    float64 re-score / fallback must: lattice targets, queries on cell centres and faces."""
    from pedp_hip import _lib

    g = np.arange(-6, 7, dtype=np.float64) * 4.0
    tgt = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3) + 100.0
    rng = np.random.default_rng(3)
    q = tgt[rng.integers(0, len(tgt), 1500)] + np.array([2.0, 2.0, 2.0])       # 8-way exact ties
    q2 = tgt[rng.integers(0, len(tgt), 1500)] + np.array([2.0, 0.0, 0.0])      # 2-way exact ties
    q3 = tgt[rng.integers(0, len(tgt), 1500)] + rng.normal(0, 1e-7, (1500, 3)) + [2.0, 2.0, 0]
    src = np.vstack([q, q2, q3])
    idx, d2 = _lib.nn(ctx, _lib.Cloud(ctx, src), _lib.Cloud(ctx, tgt))
    ridx, rd2 = oracle.nn(src, tgt)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(d2, rd2)


@pytest.mark.parametrize("estimator", [0, 1])
@pytest.mark.parametrize("config", ["tiny", "parity"])
def test_icp_trace_matches_oracle(ctx, oracle, config, estimator):
    from pedp_hip import _lib

    f, scene = _frame_scene(oracle, config)
    src = _lib.Cloud(ctx, scene)
    tgt = _lib.Cloud(ctx, f.model_points, f.normals)
    res = _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), estimator=estimator, max_iteration=20,
                   relative_fitness=-1, relative_rmse=-1, want_corr=True, want_trace=True)
    ref = oracle.icp(scene, f.model_points, f.normals, 10.0, f.icp_init(), estimator=estimator, max_iter=20,
                     rel_fitness=-1, rel_rmse=-1)
    assert res["iters"] == ref["iters"] == 20
    assert np.array_equal(res["corr"], ref["corr"])
    assert res["fitness"] == ref["fitness"]
    assert abs(res["inlier_rmse"] - ref["inlier_rmse"]) < 1e-9
    assert np.abs(res["T"] - ref["T"]).max() < POSE_TOL
    assert np.abs(res["trace"][:, :2] - ref["trace"][:, :2]).max() < 1e-9      # per-pass fitness / rmse
    assert np.abs(res["trace"][:, 2:] - ref["trace"][:, 2:]).max() < POSE_TOL  # per-pass T


def test_icp_default_criteria_early_exit(ctx, oracle):
    """Open3D defaults (30 iterations, 1e-6 / 1e-6): same stopping pass as the oracle.  A
    noise-free partial view converges and trips the relative criteria before 30."""
    from pedp_hip import _lib, synth

    f = synth.Frame("parity")
    depth = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)["t_hit"]
    hit = np.isfinite(depth)
    scene = synth.scene_from_depth(depth[hit], f.dirs[hit], noise_sigma=0.0)
    res = _lib.icp(ctx, _lib.Cloud(ctx, scene), _lib.Cloud(ctx, f.model_points, f.normals), 10.0, f.icp_init(),
                   want_corr=True)
    ref = oracle.icp(scene, f.model_points, f.normals, 10.0, f.icp_init())
    assert res["iters"] == ref["iters"]
    assert ref["iters"] < 30
    assert np.abs(res["T"] - ref["T"]).max() < POSE_TOL
    assert res["fitness"] == ref["fitness"] and np.array_equal(res["corr"], ref["corr"])


def test_icp_known_answer_noise_free(ctx):
    """Scene = rigidly moved model subset, exact correspondences exist: the recovered
    transform is the inverse motion."""
    from pedp_hip import _lib, synth

    verts, tris, normals = synth.bumpy_torus(50, 40)
    model = verts.astype(np.float64)
    M = np.eye(4)
    M[:3, :3] = synth.axis_angle([0.3, -1.0, 0.5], np.deg2rad(1.5))
    M[:3, 3] = [0.4, -0.3, 0.5]
    scene = (model[::3] @ M[:3, :3].T) + M[:3, 3]
    for est in (0, 1):
        res = _lib.icp(ctx, _lib.Cloud(ctx, scene), _lib.Cloud(ctx, model, normals), 5.0, np.eye(4),
                       estimator=est, max_iteration=60, relative_fitness=1e-12, relative_rmse=1e-12)
        assert res["fitness"] == 1.0
        assert np.abs(res["T"] - np.linalg.inv(M)).max() < 1e-8
        assert res["inlier_rmse"] < 1e-8


def test_icp_error_and_degenerate_cases(ctx, oracle):
    from pedp_hip import _lib, synth

    verts, tris, normals = synth.bumpy_torus(16, 12)
    model = verts.astype(np.float64)
    scene = model[:50] + 0.1
    src = _lib.Cloud(ctx, scene)
    with pytest.raises(_lib.PedpError, match="normals"):
        _lib.icp(ctx, src, _lib.Cloud(ctx, model), 5.0, np.eye(4), estimator=0)
    tgt = _lib.Cloud(ctx, model, normals)
    # nothing within reach: fitness 0, rmse 0, identity update, stops after one iteration
    far = np.eye(4)
    far[:3, 3] = 1e4
    res = _lib.icp(ctx, src, tgt, 1.0, far, want_corr=True)
    ref = oracle.icp(scene, model, normals, 1.0, far)
    assert res["fitness"] == ref["fitness"] == 0.0 and res["inlier_rmse"] == 0.0
    assert res["iters"] == ref["iters"] == 1 and np.array_equal(res["T"], far)
    assert (res["corr"] == -1).all()
    # max_iteration = 0 returns the initial pass
    res0 = _lib.icp(ctx, src, tgt, 5.0, np.eye(4), max_iteration=0)
    ref0 = oracle.icp(scene, model, normals, 5.0, np.eye(4), max_iter=0)
    assert res0["iters"] == 0 and res0["fitness"] == ref0["fitness"] and np.array_equal(res0["T"], np.eye(4))
    # empty source
    res_e = _lib.icp(ctx, _lib.Cloud(ctx, np.zeros((0, 3))), tgt, 5.0, np.eye(4))
    assert res_e["fitness"] == 0.0


def test_icp_batched_matches_single(ctx, oracle):
    from pedp_hip import _lib, synth

    f, scene = _frame_scene(oracle, "tiny")
    src = _lib.Cloud(ctx, scene)
    tgt = _lib.Cloud(ctx, f.model_points, f.normals)
    inits = np.stack([np.linalg.inv(T) for T in synth.batched_start_poses(19)])
    T, fit, rmse = _lib.icp_batched(ctx, src, tgt, 10.0, inits, max_iteration=5)
    for b in range(len(inits)):
        ref = oracle.icp(scene, f.model_points, f.normals, 10.0, inits[b], max_iter=5, rel_fitness=-1, rel_rmse=-1)
        assert np.abs(T[b] - ref["T"]).max() < POSE_TOL and fit[b] == ref["fitness"]


def test_cluster_poses_matches_oracle(oracle):
    from pedp_hip import _lib, synth

    rng = np.random.default_rng(0)
    poses = np.tile(np.eye(4, dtype=np.float32), (60, 1, 1))
    for k in range(60):
        poses[k, :3, :3] = synth.axis_angle(rng.normal(size=3), rng.uniform(0, np.pi)).astype(np.float32)
    syms = np.stack([np.eye(4), np.diag([-1.0, -1.0, 1.0, 1.0])]).astype(np.float32)
    got = _lib.cluster_poses(30, 99999, poses, syms)
    assert np.array_equal(got, oracle.cluster_poses(30, 99999, poses, syms))

    # and an independent statement of mycpp/src/app/pybind_api.cpp:24-68 in numpy float32 (the product
    # code and oracle/cluster.c are two short texts by one author: this one shares nothing with them)
    def restated(angle, dist, ps, tfs):
        keep = [0]
        thr = np.float32(angle / 180.0 * np.pi)
        for i in range(1, len(ps)):
            old = False
            for c in keep:
                if np.linalg.norm(ps[c][:3, 3] - ps[i][:3, 3]) >= dist:
                    continue
                for tf in tfs:
                    R = (ps[i] @ tf)[:3, :3]
                    cos = np.float32((np.trace(R @ ps[c][:3, :3].T) - 1) / 2.0)
                    if np.arccos(np.clip(cos, -1, 1)) < thr:
                        old = True
                        break
                if old:
                    break
            if not old:
                keep.append(i)
        return np.array(keep, np.int32)

    for ang in (30, 61, 95):
        assert np.array_equal(_lib.cluster_poses(ang, 99999, poses, syms), restated(ang, 99999, poses, syms))
    poses[::3, :3, 3] = 500.0                       # translation gate: far poses never merge
    assert np.array_equal(_lib.cluster_poses(61, 10.0, poses, syms), restated(61, 10.0, poses, syms))


def test_device_resident_chain_depth_to_registration(ctx, oracle):
    """Camera-rate plumbing: depth image on the device -> erode -> bilateral -> depth2xyzmap_batch ->
    valid points (torch, device) -> Cloud.from_device -> registration.  Nothing but the 4x4 comes back.
    The registration equals the one on the same points uploaded from the host, and the oracle's."""
    import torch
    from pedp_hip import _lib, compat, synth

    f = synth.Frame("parity")
    t_hit = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)["t_hit"]
    # the filters work in metres (validity 0.001, bilateral window 0.01), the registration in mm
    z = np.where(np.isfinite(t_hit), t_hit * f.dirs[:, 2] / 1000.0, 0.0).reshape(f.height, f.width).astype(np.float32)
    d = torch.from_numpy(z).cuda()
    K = torch.as_tensor(f.K, dtype=torch.float32, device="cuda")
    dm = compat.bilateral_filter_depth(compat.erode_depth(d, radius=2, depth_diff_thres=0.02, ratio_thres=0.8, zfar=100),
                                       radius=2, zfar=100, sigmaD=2, sigmaR=100000)
    xyz = compat.depth2xyzmap_batch(dm[None], K[None], zfar=np.inf)[0]
    pts = (xyz[xyz[..., 2] > 0].double() * 1000.0).contiguous()
    torch.cuda.synchronize()
    assert pts.is_cuda and len(pts) > 500
    host_pts = pts.cpu().numpy()
    src_dev = _lib.Cloud.from_device(ctx, pts.data_ptr(), len(pts))
    pts.fill_(float("nan"))                        # the copy is complete when from_device returns: overwriting
    del pts                                        # or freeing the source right away must not reach the cloud
    junk = torch.full((len(host_pts), 3), 7.0, dtype=torch.float64, device="cuda")   # (takes the freed block)
    torch.cuda.synchronize()
    tgt = _lib.Cloud(ctx, f.model_points, f.normals)
    pts = torch.from_numpy(host_pts).cuda()
    a = _lib.icp(ctx, src_dev, tgt, 10.0, f.icp_init(), max_iteration=8, relative_fitness=-1, relative_rmse=-1, want_corr=True)
    b = _lib.icp(ctx, _lib.Cloud(ctx, host_pts), tgt, 10.0, f.icp_init(), max_iteration=8, relative_fitness=-1,
                 relative_rmse=-1, want_corr=True)
    ref = oracle.icp(host_pts, f.model_points, f.normals, 10.0, f.icp_init(), max_iter=8, rel_fitness=-1, rel_rmse=-1)
    assert np.array_equal(a["corr"], ref["corr"]) and np.array_equal(b["corr"], ref["corr"])
    assert a["fitness"] == ref["fitness"] and np.abs(a["T"] - ref["T"]).max() < 1e-5 and np.abs(a["T"] - b["T"]).max() < 1e-9
    assert np.abs(np.linalg.inv(a["T"]) - f.T_gt).max() < 0.5
    # with normals, empty cloud
    n = torch.from_numpy(f.normals).cuda().contiguous(); m = torch.from_numpy(f.model_points).cuda().contiguous()
    tgt_dev = _lib.Cloud.from_device(ctx, m.data_ptr(), len(m), n.data_ptr())
    c = _lib.icp(ctx, src_dev, tgt_dev, 10.0, f.icp_init(), max_iteration=8, relative_fitness=-1, relative_rmse=-1)
    assert c["fitness"] == ref["fitness"] and np.abs(c["T"] - ref["T"]).max() < 1e-5
    empty = _lib.Cloud.from_device(ctx, pts.data_ptr(), 0)
    assert _lib.icp(ctx, empty, tgt, 10.0, np.eye(4))["fitness"] == 0.0


def test_batched_graph_is_not_replayed_on_recreated_clouds(oracle):
    """pedp_icp_batched caches one captured hipGraph per sub-context.  A new cloud of the same size
    can receive the address of a destroyed one, so the cache is keyed by a creation counter, not by
    the handle: after closing both clouds and creating new ones of equal N with different data the
    batch must equal the per-pose calls on the NEW data."""
    from pedp_hip import _lib, synth

    ctx = _lib.Context(0)
    f = synth.Frame("parity")
    depth = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)["t_hit"]
    inits = np.stack([np.linalg.inv(T) for T in synth.batched_start_poses(4)])
    for seed in (0, 7, 11):                      # same sizes every round, different points
        scene = synth.scene_from_depth(depth, f.dirs, seed=seed)
        shift = np.array([0.3, -0.2, 0.1]) * seed
        src, tgt = _lib.Cloud(ctx, scene + shift), _lib.Cloud(ctx, f.model_points, f.normals)
        T, fit, rmse = _lib.icp_batched(ctx, src, tgt, 10.0, inits, max_iteration=4)
        for b in range(4):
            one = _lib.icp(ctx, src, tgt, 10.0, inits[b], max_iteration=4, relative_fitness=-1, relative_rmse=-1)
            assert np.array_equal(T[b], one["T"]) and fit[b] == one["fitness"] and rmse[b] == one["inlier_rmse"]
        src.close()
        tgt.close()
    ctx.close()


def test_handles_outlive_their_context():
    """Python may collect a Context before its clouds and meshes: destroying those afterwards must
    not touch the freed context (the handles keep the device ordinal themselves)."""
    from pedp_hip import _lib, synth

    f = synth.Frame("tiny")
    ctx = _lib.Context(0)
    cloud = _lib.Cloud(ctx, f.model_points, f.normals)
    mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
    ctx.close()
    cloud.close()
    mesh.close()


def test_exhaustive_mode_sweeps_all_pairs_with_identical_results(ctx, oracle):
    """pedp_icp_configure(exhaustive=1): every (scene point, model point) pair goes through the MFMA
    sweep in every pass; correspondences, fitness and pose are those of the culled run."""
    from pedp_hip import _lib, synth

    f = synth.Frame("parity")
    depth = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)["t_hit"]
    scene = f.scene(depth)
    src, tgt = _lib.Cloud(ctx, scene), _lib.Cloud(ctx, f.model_points, f.normals)
    kw = dict(max_iteration=5, relative_fitness=-1, relative_rmse=-1, want_corr=True)
    culled = _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), **kw)
    _, pairs_culled, _ = _lib.icp_last_stats(ctx)
    _lib.icp_configure(ctx, exhaustive=True, timed_pass=1)
    try:
        full = _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), **kw)
        passes, pairs, _ = _lib.icp_last_stats(ctx)
        ms = _lib.nn_last_sweep_ms(ctx)
    finally:
        _lib.icp_configure(ctx)
    assert np.array_equal(full["corr"], culled["corr"]) and full["fitness"] == culled["fitness"]
    assert np.abs(full["T"] - culled["T"]).max() < 1e-9
    assert pairs >= passes * len(scene) * len(f.model_points) > pairs_culled   # padded all-pairs count
    assert 0 < ms < 1000


def test_batched_ex_per_pose_radius_and_criteria(ctx, oracle):
    """pedp_icp_batched_ex: every start pose has its own radius and stops by its own criteria (what
    improve_result's restarts need); results and iteration counts equal the single calls, which
    equal the oracle."""
    from pedp_hip import _lib, synth

    f, scene = _frame_scene(oracle, "parity")
    src, tgt = _lib.Cloud(ctx, scene), _lib.Cloud(ctx, f.model_points, f.normals)
    inits = np.stack([np.linalg.inv(T) for T in synth.batched_start_poses(11)])
    radii = 6.0 * np.cumprod(np.random.default_rng(2).uniform(0.8, 1.2, 11))     # a threshold walk like the reference's
    T, fit, rmse, its = _lib.icp_batched_ex(ctx, src, tgt, radii, inits, max_iteration=30)
    assert len(set(its.tolist())) > 1 and its.max() <= 30                        # they stop at different iterations
    for b in range(11):
        one = _lib.icp(ctx, src, tgt, radii[b], inits[b], max_iteration=30)
        assert np.array_equal(T[b], one["T"]) and fit[b] == one["fitness"] and rmse[b] == one["inlier_rmse"]
        assert its[b] == one["iters"]
    for b in (0, 7):
        ref = oracle.icp(scene, f.model_points, f.normals, radii[b], inits[b])
        assert fit[b] == ref["fitness"] and its[b] == ref["iters"] and np.abs(T[b] - ref["T"]).max() < POSE_TOL
    # a radius too large for the fused pass in the mix: the batch falls back to one-by-one, same answers
    radii2 = radii.copy()
    radii2[3] = 500.0
    T2, fit2, _, its2 = _lib.icp_batched_ex(ctx, src, tgt, radii2, inits, max_iteration=6)
    one = _lib.icp(ctx, src, tgt, 500.0, inits[3], max_iteration=6)
    assert np.array_equal(T2[3], one["T"]) and fit2[3] == one["fitness"] and its2[3] == one["iters"]


def test_batched_graphs_survive_a_change_of_the_iteration_limit(ctx, oracle):
    """Two batched calls on the same handles and group size whose iteration limits differ (50, then 40): the
    per-pose workspace layout a cached batch graph bakes in depends on the limit's capacity class only, so
    the second call must neither replay a graph with stale offsets nor differ from the single calls
    (the key used to ignore the limit)."""
    from pedp_hip import _lib, synth

    f, scene = _frame_scene(oracle, "parity")
    src, tgt = _lib.Cloud(ctx, scene), _lib.Cloud(ctx, f.model_points, f.normals)
    inits = np.stack([np.linalg.inv(T) for T in synth.batched_start_poses(5)])
    radii = np.full(5, 8.0)
    for limit in (50, 40, 70, 3):    # 70 + 2 crosses into the next capacity class: captured again
        T, fit, rmse, its = _lib.icp_batched_ex(ctx, src, tgt, radii, inits, max_iteration=limit)
        for b in range(5):
            one = _lib.icp(ctx, src, tgt, radii[b], inits[b], max_iteration=limit)
            assert np.array_equal(T[b], one["T"]) and fit[b] == one["fitness"] and rmse[b] == one["inlier_rmse"]
            assert its[b] == one["iters"] <= limit


def test_results_do_not_depend_on_a_handles_history(ctx, oracle):
    """The scene's spatial order -- the order of every float64 sum of a registration -- is built from the cloud
    alone.  Two fresh handles of the same data, first touched by registrations from very different start
    poses (and one of them by a batch), then give identical bits for the same registration."""
    from pedp_hip import _lib, synth

    f, scene = _frame_scene(oracle, "parity")
    tgt = _lib.Cloud(ctx, f.model_points, f.normals)
    init = f.icp_init()
    far = init.copy()
    far[:3, 3] += np.array([400.0, -250.0, 300.0])           # a first registration somewhere else entirely
    poses = np.stack([np.linalg.inv(T) for T in synth.batched_start_poses(4)])
    a, b, c = (_lib.Cloud(ctx, scene) for _ in range(3))
    _lib.icp(ctx, a, tgt, 10.0, far, max_iteration=2)
    _lib.icp_batched(ctx, b, tgt, 6.0, poses, max_iteration=3)
    res = [_lib.icp(ctx, h, tgt, 10.0, init, max_iteration=12, relative_fitness=-1, relative_rmse=-1, want_corr=True, want_trace=True)
           for h in (a, b, c)]
    for r in res[1:]:
        assert np.array_equal(r["T"], res[0]["T"]) and np.array_equal(r["trace"], res[0]["trace"])
        assert np.array_equal(r["corr"], res[0]["corr"]) and r["inlier_rmse"] == res[0]["inlier_rmse"]


_FINISH_PROBE = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/oracle")
from pedp_hip import _lib, synth
ctx = _lib.Context(0)
f = synth.Frame("parity")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
scene = f.scene(mesh.cast_rays(f.rays6, want_uv=False)["t_hit"])
src, tgt = _lib.Cloud(ctx, scene), _lib.Cloud(ctx, f.model_points, f.normals)
inits = np.stack([np.linalg.inv(T) for T in synth.batched_start_poses(9)])
one = _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), max_iteration=15, relative_fitness=-1, relative_rmse=-1, want_corr=True, want_trace=True)
p2p = _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), max_iteration=5, estimator=_lib.POINT_TO_POINT)
T, fit, rmse, its = _lib.icp_batched_ex(ctx, src, tgt, np.full(9, 7.0), inits, max_iteration=30)
np.savez(sys.argv[2], T=one["T"], trace=one["trace"], corr=one["corr"], pT=p2p["T"], bT=T, bfit=fit, brmse=rmse, bits=its)
"""


def test_in_launch_finish_equals_the_finish_kernel(tmp_path):
    """The pass is closed inside the launch by the workgroup that finishes last (partial sums handed over with
    write-through stores, a ticket and sc1 loads).  PEDP_ICP_UNFUSED_FINISH=1 closes it with a launch of its
    own behind a kernel boundary instead: every result must agree in every bit."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in ("0", "1"):
        out = str(tmp_path / f"finish{flag}.npz")
        env = dict(os.environ, PEDP_ICP_UNFUSED_FINISH=flag)
        p = subprocess.run([sys.executable, "-c", _FINISH_PROBE, root, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert p.returncode == 0, p.stdout.decode()
        outs.append(np.load(out))
    for k in outs[0].files:
        assert np.array_equal(outs[0][k], outs[1][k]), k


_PLAN_PROBE = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/oracle")
from pedp_hip import _lib, synth
ctx = _lib.Context(0)
f = synth.Frame("bench_100k")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
scene = f.scene(mesh.cast_rays(f.rays6, want_uv=False)["t_hit"])
src, tgt = _lib.Cloud(ctx, scene), _lib.Cloud(ctx, f.model_points, f.normals)
out = {}
for name, init, radius in (("a", f.icp_init(), 10.0), ("b", np.linalg.inv(synth.batched_start_poses(3)[2]), 10.0), ("c", f.icp_init(), 22.0)):
    r = _lib.icp(ctx, src, tgt, radius, init, max_iteration=20, relative_fitness=-1, relative_rmse=-1, want_corr=True, want_trace=True)
    out[name + "T"], out[name + "trace"], out[name + "corr"] = r["T"], r["trace"], r["corr"]
    out[name + "planned"] = np.int64(_lib.icp_last_planned_passes(ctx))
np.savez(sys.argv[2], **out)
"""


def test_visit_plan_changes_no_bit(tmp_path):
    """With more live chunks than CUs (the bench frame: 281 on 256) the first workgroup through a pass hands the
    lightest chunks to the workgroup positions that share a CU.  Scheduling only: the transformation, every pass of
    the trace and every correspondence agree bit for bit with PEDP_ICP_NO_VISIT_PLAN=1 -- and the plan was in force."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in ("0", "1"):
        out = str(tmp_path / f"plan{flag}.npz")
        env = dict(os.environ, PEDP_ICP_NO_VISIT_PLAN=flag)
        p = subprocess.run([sys.executable, "-c", _PLAN_PROBE, root, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert p.returncode == 0, p.stdout.decode()
        outs.append(np.load(out))
    for k in outs[0].files:
        if not k.endswith("planned"):
            assert np.array_equal(outs[0][k], outs[1][k]), k
    assert int(outs[1]["aplanned"]) == 0 and int(outs[1]["bplanned"]) == 0 and int(outs[1]["cplanned"]) == 0   # (c: a wider radius, more live chunks)
    import torch
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:   # 281 live chunks: a plan from the fourth steady pass on
        assert int(outs[0]["aplanned"]) >= 10, int(outs[0]["aplanned"])


def test_begin_end_equals_the_blocking_call(ctx, oracle):
    """pedp_icp_begin / pedp_icp_end: the registration enqueued whole, collected later -- the same bits as pedp_icp, with
    and without the early exit; a second registration on the context in between is refused."""
    from pedp_hip import _lib

    f, scene = _frame_scene(oracle, "parity")
    src, tgt = _lib.Cloud(ctx, scene), _lib.Cloud(ctx, f.model_points, f.normals)
    for kw in (dict(max_iteration=12, relative_fitness=-1, relative_rmse=-1), dict(max_iteration=30)):
        one = _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), want_corr=True, want_trace=True, **kw)
        _lib.icp_begin(ctx, src, tgt, 10.0, f.icp_init(), want_trace=True, **kw)
        with pytest.raises(_lib.PedpError):
            _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), **kw)
        two = _lib.icp_end(ctx, want_corr=True)
        assert np.array_equal(one["T"], two["T"]) and one["fitness"] == two["fitness"] and one["inlier_rmse"] == two["inlier_rmse"]
        assert one["iters"] == two["iters"] and np.array_equal(one["corr"], two["corr"]) and np.array_equal(one["trace"], two["trace"])
    with pytest.raises(_lib.PedpError):
        _lib.icp_end(ctx)


_DENSE_PROBE = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/oracle")
from pedp_hip import _lib, synth
ctx = _lib.Context(0)
f = synth.Frame("parity")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
scene = f.scene(mesh.cast_rays(f.rays6, want_uv=False)["t_hit"])
src, tgt = _lib.Cloud(ctx, scene), _lib.Cloud(ctx, f.model_points, f.normals)
idx, d2 = _lib.nn(ctx, src, tgt, f.icp_init())
_lib.icp_configure(ctx, exhaustive=True, timed_pass=1)
r = _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), max_iteration=8, relative_fitness=-1, relative_rmse=-1, want_corr=True, want_trace=True)
np.savez(sys.argv[2], idx=idx, d2=d2, T=r["T"], corr=r["corr"], trace=r["trace"], fb=np.int64(_lib.icp_last_stats(ctx)[2]))
"""


def test_dense_sweep_on_the_bf16_pipe_equals_the_f32_form(tmp_path):
    """The all-pairs sweep (pedp_nn, the exhaustive configuration, large radii) runs as one v_mfma_f32_16x16x32_bf16 per
    16 x 16 pairs over exact three-way bf16 pieces; PEDP_NN_F32=1 keeps the f32-input MFMA of rounds 1-3.  Both are
    filters in front of the float64 selection: nearest neighbours, squared distances, correspondences and every pass of the
    trace agree bit for bit (the bf16 form's wider error bound only sends more slots to the exact search)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in ("0", "1"):
        out = str(tmp_path / f"dense{flag}.npz")
        env = dict(os.environ, PEDP_NN_F32=flag)
        p = subprocess.run([sys.executable, "-c", _DENSE_PROBE, root, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert p.returncode == 0, p.stdout.decode()
        outs.append(np.load(out))
    for k in ("idx", "d2", "T", "corr", "trace"):
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert int(outs[0]["fb"]) >= int(outs[1]["fb"])


def test_bf16_contraction_error_stays_inside_the_bound(ctx):
    """The dense sweep's g = |t'|^2 - 2 s'.t' as ONE bf16 MFMA over exact three-way pieces: against float64 on the same
    float32 inputs the error is the pipe's fp32 accumulation alone.  The slot's bound charges 34 M 2^-23 for it (M = 2 |s'|_1
    max|t'|_1 + max|t'|^2); measured over 4 M pairs at the bench frame's scale it uses a small part of that."""
    from pedp_hip import _lib

    rng = np.random.default_rng(3)
    worst = 0.0
    for scale_s, scale_t in ((400.0, 120.0), (60.0, 60.0), (5.0, 150.0)):
        s = (rng.uniform(-1, 1, (2048, 3)) * scale_s).astype(np.float32)
        t = (rng.uniform(-1, 1, (2048, 3)) * scale_t).astype(np.float32)
        t2 = (t.astype(np.float64) ** 2).sum(1).astype(np.float32)
        src4 = np.column_stack([-2.0 * s, np.ones(len(s), np.float32)]).astype(np.float32)
        tgt4 = np.column_stack([t, t2]).astype(np.float32)
        g = _lib.debug_nn_bf16(ctx, src4, tgt4).astype(np.float64)
        ref = t2.astype(np.float64)[None, :] + src4[:, :3].astype(np.float64) @ t.astype(np.float64).T
        Tn, T2 = np.abs(t).sum(1).max(), t2.max()
        M = 2.0 * np.abs(s).sum(1) * Tn + T2                          # per scene row, like pack_store's Mi
        used = np.abs(g - ref) / (34.0 * M[:, None] * 2.0 ** -23)
        worst = max(worst, float(used.max()))
    assert worst < 0.25, worst
