"""GPU parity at BASELINE.json's full sizes.  The oracle's BVH gives the bit-exact reference
in about a second on the GPU box's host cores; on top of it, size-independent properties:
all sweep variants agree, ray order does not matter, ICP recovers the known pose."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(res, ref):
    assert np.array_equal(res["primitive_ids"], ref["primitive_ids"])
    assert np.array_equal(res["t_hit"].view(np.uint32), ref["t_hit"].view(np.uint32))


@pytest.fixture(scope="module")
def frame100k():
    from pedp_hip import synth

    return synth.Frame("bench_100k")


def test_bench_100k_rays_bit_exact_all_variants(ctx, oracle, frame100k):
    from pedp_hip import _lib

    f = frame100k
    ref = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)
    assert np.isfinite(ref["t_hit"]).sum() == 35863            # hit count of this frame (oracle)
    mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
    for variant in (4, 3, 1, 5):                                 # triangle-driven grid (the default), cone-culled, exhaustive on the matrix pipe, exhaustive packed fp32
        _lib.raycast_configure(ctx, 0, variant)
        try:
            got = mesh.cast_rays(f.rays6)                        # natural ray order, with uv
            ran = _lib.raycast_last_variant(ctx)
        finally:
            _lib.raycast_configure(ctx, 0, 0)
        assert ran[0] == variant and (variant != 4 or ran[1] == 0)   # the GRID answered, not the exhaustive kernel behind it
        _same(got, ref)
        assert np.array_equal(got["primitive_uvs"].view(np.uint32), ref["primitive_uvs"].view(np.uint32))
    auto = mesh.cast_rays(f.rays6)
    assert _lib.raycast_last_variant(ctx) == (4, 0)              # ... and it is what a plain call of this frame takes
    _same(auto, ref)
    # permutation invariance: shuffling the rays shuffles the answers, nothing else
    perm = np.random.default_rng(0).permutation(f.n_rays)
    shuffled = mesh.cast_rays(f.rays6[perm], want_uv=False)
    assert np.array_equal(shuffled["primitive_ids"], ref["primitive_ids"][perm])
    assert np.array_equal(shuffled["t_hit"].view(np.uint32), ref["t_hit"][perm].view(np.uint32))
    # general-origin path on the same frame: one origin nudged -> device falls back, same bits elsewhere
    nudged = f.rays6.copy()
    nudged[12345, 0] = 1e-3
    g = mesh.cast_rays(nudged, want_uv=False)
    keep = np.arange(f.n_rays) != 12345
    assert np.array_equal(g["primitive_ids"][keep], ref["primitive_ids"][keep])


def test_bench_100k_icp_matches_oracle_and_ground_truth(ctx, oracle, frame100k):
    from pedp_hip import _lib

    f = frame100k
    depth = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)["t_hit"]
    scene = f.scene(depth)
    src, tgt = _lib.Cloud(ctx, scene), _lib.Cloud(ctx, f.model_points, f.normals)
    # the bench's workload -- 20 iterations -- against the oracle's KD-tree run: every one of the 21 passes
    # (fitness equal, rmse and pose to rounding) and the final correspondence set
    res = _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), max_iteration=20, relative_fitness=-1, relative_rmse=-1,
                   want_corr=True, want_trace=True)
    ref = oracle.icp(scene, f.model_points, f.normals, 10.0, f.icp_init(), max_iter=20, rel_fitness=-1, rel_rmse=-1)
    assert res["trace"].shape == (21, 18) and ref["trace"].shape == (21, 18)
    assert np.array_equal(res["corr"], ref["corr"])
    assert res["fitness"] == ref["fitness"] and abs(res["inlier_rmse"] - ref["inlier_rmse"]) < 1e-9
    assert np.array_equal(res["trace"][:, 0], ref["trace"][:, 0])            # same inlier count in every pass
    assert np.abs(res["trace"][:, 1] - ref["trace"][:, 1]).max() < 1e-9
    assert np.abs(res["trace"][:, 2:] - ref["trace"][:, 2:]).max() < 1e-5    # the 1e-5 bar of north_star (observed ~1e-12)
    assert np.abs(res["T"] - ref["T"]).max() < 1e-5
    passes, pairs, fb = _lib.icp_last_stats(ctx)
    assert passes == 21 and 0 < pairs < 21 * len(scene) * len(f.model_points) * 0.01   # culling at work: under 1 % of all pairs
    # ... and they land on the ground-truth pose (0.5 mm depth noise)
    assert np.abs(np.linalg.inv(res["T"]) - f.T_gt).max() < 0.05
    # the correspondence sets of the early passes too (4 iterations: the passes with the large updates)
    r4 = _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), max_iteration=4, relative_fitness=-1, relative_rmse=-1, want_corr=True)
    o4 = oracle.icp(scene, f.model_points, f.normals, 10.0, f.icp_init(), max_iter=4, rel_fitness=-1, rel_rmse=-1)
    assert np.array_equal(r4["corr"], o4["corr"]) and r4["fitness"] == o4["fitness"]
    # exact NN over ALL pairs (no radius): indices and float64 distances equal the KD-tree oracle
    idx, d2 = _lib.nn(ctx, src, tgt, f.icp_init())
    ridx, rd2 = oracle.nn(oracle.transform(f.icp_init(), scene), f.model_points, kdtree=True)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)


def test_bench_1m_config_rays(ctx, oracle):
    """BASELINE config 4 at one GPU: 1M-triangle mesh, 1280x720 dense frame (921,600 rays)."""
    from pedp_hip import _lib, synth

    f = synth.Frame("bench_1m")
    assert f.n_tris == 1_000_000 and f.n_rays == 921_600
    ref = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)
    mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
    got = mesh.cast_rays(f.rays6)
    assert _lib.raycast_last_variant(ctx) == (4, 0)              # the triangle-driven grid answered all 921,600 rays itself
    _same(got, ref)
    assert np.array_equal(got["primitive_uvs"].view(np.uint32), ref["primitive_uvs"].view(np.uint32))
    assert np.isfinite(ref["t_hit"]).sum() > 50_000


def test_batched_256_pose_refine(ctx, oracle, frame100k):
    """BASELINE config 3: 256 start poses share one scene and one model (FoundationPose's
    render-and-compare sizing, estimater.py:104-122).  pedp_icp_batched runs them on concurrent
    streams; every pose must equal the one-by-one call bit for bit, a sample of them the oracle."""
    import time
    from pedp_hip import _lib, synth

    f = frame100k
    depth = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)["t_hit"]
    scene = f.scene(depth)
    src, tgt = _lib.Cloud(ctx, scene), _lib.Cloud(ctx, f.model_points, f.normals)
    inits = np.stack([np.linalg.inv(T) for T in synth.batched_start_poses(256)])
    t0 = time.perf_counter()
    T, fit, rmse = _lib.icp_batched(ctx, src, tgt, 10.0, inits, max_iteration=6)
    dt = time.perf_counter() - t0
    print(f"256-pose batch, 6 iterations each: {1e3 * dt:.1f} ms")
    assert T.shape == (256, 4, 4) and np.all(np.isfinite(T)) and np.all((fit > 0) & (fit <= 1))
    for b in (0, 1, 77, 128, 255):
        one = _lib.icp(ctx, src, tgt, 10.0, inits[b], max_iteration=6, relative_fitness=-1, relative_rmse=-1)
        assert np.array_equal(T[b], one["T"]) and fit[b] == one["fitness"] and rmse[b] == one["inlier_rmse"]
    for b in (3, 200):
        ref = oracle.icp(scene, f.model_points, f.normals, 10.0, inits[b], max_iter=6, rel_fitness=-1, rel_rmse=-1)
        assert fit[b] == ref["fitness"] and np.abs(T[b] - ref["T"]).max() < 1e-5


def test_bench_1m_config_icp(ctx, oracle):
    """The 1M-triangle configuration's clouds through ICP: 921,600 scene points against the
    500,000 model vertices, two iterations against the KD-tree oracle."""
    from pedp_hip import _lib, synth

    f = synth.Frame("bench_1m")
    depth = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)["t_hit"]
    scene = f.scene(depth)
    assert len(scene) == 921_600 and len(f.model_points) == 500_000
    src, tgt = _lib.Cloud(ctx, scene), _lib.Cloud(ctx, f.model_points, f.normals)
    res = _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), max_iteration=2, relative_fitness=-1, relative_rmse=-1, want_corr=True)
    ref = oracle.icp(scene, f.model_points, f.normals, 10.0, f.icp_init(), max_iter=2, rel_fitness=-1, rel_rmse=-1)
    assert np.array_equal(res["corr"], ref["corr"])
    assert res["fitness"] == ref["fitness"] and np.abs(res["T"] - ref["T"]).max() < 1e-5
