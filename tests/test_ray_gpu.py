"""GPU parity of the ray sweep against the oracle: t_hit / primitive_ids / uv bit-exact
(BASELINE.md s6).  Everything goes through the C ABI (pedp_raycast)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(res, ref):
    assert np.array_equal(res["primitive_ids"], ref["primitive_ids"])
    assert np.array_equal(res["t_hit"].view(np.uint32), ref["t_hit"].view(np.uint32))
    assert np.array_equal(res["primitive_uvs"].view(np.uint32), ref["primitive_uvs"].view(np.uint32))


@pytest.mark.parametrize("config", ["tiny", "parity"])
@pytest.mark.parametrize("variant,chunks", [(1, 0), (1, 8), (1, 64), (2, 0), (2, 16), (3, 0), (4, 0)])
def test_frame_bit_exact(ctx, oracle, config, variant, chunks):
    from pedp_hip import _lib, synth

    f = synth.Frame(config)
    ref = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)
    assert np.isfinite(ref["t_hit"]).sum() > 50
    _lib.raycast_configure(ctx, chunks, variant)
    try:
        mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
        _same(mesh.cast_rays(f.rays6), ref)
    finally:
        _lib.raycast_configure(ctx, 0, 0)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 255, 257, 1000])
def test_ragged_ray_counts(ctx, oracle, n):
    from pedp_hip import _lib, synth

    f = synth.Frame("tiny")
    rng = np.random.default_rng(n)
    sel = rng.choice(f.n_rays, size=n, replace=False) if n else np.zeros(0, int)
    rays = f.rays6[sel]
    mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
    res = mesh.cast_rays(rays)
    ref = oracle.raycast(f.verts_posed, f.tris, rays)
    _same(res, ref)


def test_general_origins_and_unnormalised_directions(ctx, oracle):
    """cast_rays takes arbitrary [o|d] rows: per-ray origins, directions used as given."""
    from pedp_hip import _lib, synth

    f = synth.Frame("tiny")
    rng = np.random.default_rng(5)
    n = 3000
    o = rng.normal(0, 150, size=(n, 3))
    tgt = f.verts_posed[rng.integers(0, len(f.verts_posed), n)] + rng.normal(0, 5, size=(n, 3))
    d = (tgt - o) * rng.uniform(0.01, 3.0, size=(n, 1))
    rays = np.hstack([o, d]).astype(np.float32)
    mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
    for variant in (1, 2, 3, 4):
        _lib.raycast_configure(ctx, 0, variant)
        try:
            _same(mesh.cast_rays(rays), oracle.raycast(f.verts_posed, f.tris, rays))
        finally:
            _lib.raycast_configure(ctx, 0, 0)


def test_known_answers_single_triangle_and_cube(ctx):
    from pedp_hip import _lib

    v = np.array([[0, 0, 5], [1, 0, 5], [0, 1, 5]], np.float32)
    t = np.array([[0, 1, 2]], np.uint32)
    mesh = _lib.Mesh(ctx, v, t)
    rays = np.array([
        [0.25, 0.25, 0, 0, 0, 1],    # interior: t = 5, u = v = 0.25
        [0.25, 0.25, 0, 0, 0, -1],   # behind the origin: miss
        [0.0, 0.0, 0, 0, 0, 1],      # vertex v0: inclusive
        [0.5, 0.5, 0, 0, 0, 1],      # on the hypotenuse: inclusive
        [0.6, 0.6, 0, 0, 0, 1],      # outside
        [0.25, 0.25, 10, 0, 0, -1],  # back face: no culling, t = 5
        [0.25, 0.25, 0, 0, 0, 2],    # unnormalised direction: t = 2.5
        [0.25, 0.25, 0, 1, 0, 0],    # parallel: det == 0 -> miss
    ], np.float32)
    r = mesh.cast_rays(rays)
    assert r["t_hit"].tolist() == [5.0, np.inf, 5.0, 5.0, np.inf, 5.0, 2.5, np.inf]
    assert r["primitive_ids"].tolist() == [0, 0xFFFFFFFF, 0, 0, 0xFFFFFFFF, 0, 0, 0xFFFFFFFF]
    assert r["primitive_uvs"][0].tolist() == [0.25, 0.25]
    # unit cube [0,1]^3, 12 triangles; a ray along +z through (0.3, 0.6) enters at z=0
    cv = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    ct = np.array([[a, b, c] for a, b, c, d in quads] + [[a, c, d] for a, b, c, d in quads], np.uint32)
    cube = _lib.Mesh(ctx, cv, ct)
    rr = cube.cast_rays(np.array([[0.3, 0.6, -2, 0, 0, 1], [0.3, 0.6, 0.5, 0, 0, 1], [3, 3, -2, 0, 0, 1]], np.float32))
    assert rr["t_hit"].tolist() == [2.0, 0.5, np.inf]


def test_tie_takes_lowest_triangle_index(ctx, oracle):
    """Two coincident triangles: equal t -> the smaller index wins, whatever the chunking."""
    from pedp_hip import _lib

    base = np.array([[0, 0, 5], [1, 0, 5], [0, 1, 5]], np.float32)
    far = base + np.float32([0, 0, 3])
    v = np.vstack([far, base, base])
    pad = 70  # push the duplicates into different unroll groups / lanes
    filler = np.tile(np.arange(3, dtype=np.uint32), (pad, 1))          # far triangle, many times
    t = np.vstack([filler, [[3, 4, 5]], filler, [[6, 7, 8]]]).astype(np.uint32)
    rays = np.tile(np.array([[0.2, 0.3, 0, 0, 0, 1]], np.float32), (130, 1))
    mesh = _lib.Mesh(ctx, v, t)
    for variant in (1, 2, 3, 4):
        _lib.raycast_configure(ctx, 8, variant)
        try:
            r = mesh.cast_rays(rays)
        finally:
            _lib.raycast_configure(ctx, 0, 0)
        assert (r["primitive_ids"] == pad).all() and (r["t_hit"] == 5.0).all()
        ref = oracle.raycast(v, t, rays)
        assert np.array_equal(r["primitive_ids"], ref["primitive_ids"])


def test_shared_and_mixed_origins_same_bits(ctx, oracle):
    """The shared-origin fast path (origin-dependent terms hoisted per triangle) and the
    general path give identical bits; one differing origin anywhere switches paths."""
    from pedp_hip import _lib, synth

    f = synth.Frame("tiny")
    rays = f.rays6.copy()
    rays[:, :3] = np.float32([3.25, -1.5, 0.75])          # shared, non-zero origin
    mixed = rays.copy()
    mixed[-1, 0] = np.nextafter(mixed[-1, 0], np.float32(10))   # one ulp off in the last ray
    mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
    ref_a, ref_b = oracle.raycast(f.verts_posed, f.tris, rays), oracle.raycast(f.verts_posed, f.tris, mixed)
    for variant in (1, 3, 4):
        _lib.raycast_configure(ctx, 0, variant)
        try:
            a, b = mesh.cast_rays(rays), mesh.cast_rays(mixed)
        finally:
            _lib.raycast_configure(ctx, 0, 0)
        _same(a, ref_a)
        _same(b, ref_b)
        assert np.array_equal(a["primitive_ids"][:-1], b["primitive_ids"][:-1])


def test_culling_is_conservative_on_hard_packets(ctx, oracle):
    """Culled sweep (variant 3) on ray sets that stress the cone logic: origin outside / in the
    hole / inside the tube, rays in all directions (wide packets), rays aimed exactly at
    vertices (edge hits), zero and NaN directions."""
    from pedp_hip import _lib, synth

    f = synth.Frame("parity")
    rng = np.random.default_rng(11)
    n = 20000
    mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
    centre = f.T_gt[:3, 3]
    for origin in (np.zeros(3), centre, centre + f.T_gt[:3, :3] @ [60.0, 0.0, 0.0]):
        d = rng.normal(size=(n, 3))
        d[:50] = 0.0
        d[50:60] = np.nan
        d[5000:10000] = f.verts_posed[rng.integers(0, len(f.verts_posed), 5000)] - origin
        rays = np.hstack([np.tile(origin, (n, 1)), d]).astype(np.float32)
        ref = oracle.raycast(f.verts_posed, f.tris, rays)
        assert np.isfinite(ref["t_hit"]).sum() > 1000
        for variant in (3, 4):     # 4: rays in all directions leave the grid's half space -> completed exhaustively
            _lib.raycast_configure(ctx, 0, variant)
            try:
                got = mesh.cast_rays(rays)
            finally:
                _lib.raycast_configure(ctx, 0, 0)
            _same(got, ref)


def test_bad_arguments_raise(ctx):
    from pedp_hip import _lib

    with pytest.raises(_lib.PedpError):
        _lib.Mesh(ctx, np.zeros((3, 3), np.float32), np.array([[0, 1, 3]], np.uint32))  # index >= V
    with pytest.raises(_lib.PedpError):
        _lib.raycast_configure(ctx, 12, 1)  # chunks must be a multiple of 8


def test_empty_mesh_all_miss(ctx):
    from pedp_hip import _lib

    mesh = _lib.Mesh(ctx, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32))
    r = mesh.cast_rays(np.array([[0, 0, 0, 0, 0, 1]] * 5, np.float32))
    assert np.isinf(r["t_hit"]).all() and (r["primitive_ids"] == 0xFFFFFFFF).all()


def test_kept_direction_order_never_changes_results(ctx, oracle):
    """The culled sweep keeps the direction order of the previous frame's rays while a sample of the
    directions is unchanged.  The order only shapes the packets (their cones come from the rays they
    actually hold), so frames that differ from the kept order's -- other rays of the same count,
    changes between the samples, a shuffled frame -- must still give the oracle's bits."""
    from pedp_hip import _lib, synth

    f = synth.Frame("parity")
    own = _lib.Context(0)
    mesh = _lib.Mesh(own, f.verts_posed, f.tris)
    _lib.raycast_configure(own, 0, 3)
    ref = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)

    def same(rays, want):
        got = mesh.cast_rays(rays)
        assert np.array_equal(got["primitive_ids"], want["primitive_ids"])
        assert np.array_equal(got["t_hit"].view(np.uint32), want["t_hit"].view(np.uint32))

    same(f.rays6, ref)
    same(f.rays6, ref)                                            # second frame: order kept
    # changes that the 4,096 samples cannot all see: every ray not sampled gets a new direction
    rng = np.random.default_rng(1)
    n = len(f.rays6)
    sampled = (np.arange(4096, dtype=np.int64) * n) // 4096
    moved = f.rays6.copy()
    other = np.setdiff1d(np.arange(n), sampled)
    d = moved[other, 3:] + rng.normal(0, 0.2, (len(other), 3)).astype(np.float32)
    moved[other, 3:] = d / np.linalg.norm(d, axis=1, keepdims=True)
    same(moved, oracle.raycast(f.verts_posed, f.tris, moved, bvh=True))      # stale order, kept: still exact
    # a shuffled frame (samples differ -> rebuilt) and back
    perm = rng.permutation(n)
    shuffled = f.rays6[perm]
    same(shuffled, {"primitive_ids": ref["primitive_ids"][perm], "t_hit": ref["t_hit"][perm]})
    same(f.rays6, ref)
    # another ray count invalidates the kept order
    same(f.rays6[: n - 777], {"primitive_ids": ref["primitive_ids"][: n - 777], "t_hit": ref["t_hit"][: n - 777]})
    mesh.close()
    own.close()


def _grid_cast(ctx, mesh, rays, want_status=None):
    from pedp_hip import _lib

    _lib.raycast_configure(ctx, 0, 4)
    try:
        got = mesh.cast_rays(rays)
        variant, status = _lib.raycast_last_variant(ctx)
    finally:
        _lib.raycast_configure(ctx, 0, 0)
    assert variant == 4
    if want_status is not None:
        assert status == want_status, f"grid status {status}, expected {want_status}"
    return got


def test_grid_sweep_answers_camera_frames_itself(ctx, oracle):
    """Variant 4 (triangle-driven sweep over a grid of directions) on pinhole frames: the grid -- not
    the exhaustive kernel behind it -- answers the cast (status 0), bit for bit the oracle's hits.
    Sub-sets of a frame's rays (what the fused projection casts: the hot pixels only) leave cells
    empty; a frame seen from inside the mesh has triangles behind and across the frame's plane."""
    from pedp_hip import _lib, synth

    for config in ("tiny", "parity"):
        f = synth.Frame(config)
        mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
        ref = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)
        _same(_grid_cast(ctx, mesh, f.rays6, 0), ref)
        rng = np.random.default_rng(3)
        sub = np.sort(rng.choice(f.n_rays, f.n_rays // 7, replace=False))
        _same(_grid_cast(ctx, mesh, f.rays6[sub], 0), oracle.raycast(f.verts_posed, f.tris, f.rays6[sub], bvh=True))
    # camera inside the tube, looking along it: triangles all around, behind and across the plane through the origin
    f = synth.Frame("parity")
    mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
    centre = f.T_gt[:3, 3]
    hits = 0
    R = f.T_gt[:3, :3]
    for origin in (centre, centre + R @ [60.0, 0.0, 0.0], centre + R @ [0.0, 0.0, 35.0], centre + R @ [45.0, 20.0, -10.0]):
        for turn in (np.eye(3), R, R[:, [2, 0, 1]]):      # the frame's rays as they are, and turned along the object's axes
            rays = f.rays6.copy()
            rays[:, :3] = origin.astype(np.float32)
            rays[:, 3:] = (f.rays6[:, 3:].astype(np.float64) @ turn.T).astype(np.float32)
            ref = oracle.raycast(f.verts_posed, f.tris, rays)
            hits += int(np.isfinite(ref["t_hit"]).sum())
            _same(_grid_cast(ctx, mesh, rays, 0), ref)
    assert hits > 20000


def test_grid_sweep_large_small_and_degenerate_triangles(ctx, oracle):
    """Triangles far larger than a cell (a two-triangle floor across the whole view: cut into wave
    items), slivers, zero-area triangles, triangles in a plane through the origin (edge-on from every
    ray), rays aimed exactly at vertices and along shared edges."""
    from pedp_hip import _lib, synth

    f = synth.Frame("parity")
    rng = np.random.default_rng(9)
    z = float(f.verts_posed[:, 2].max()) + 40.0
    floor_v = np.array([[-4000, -4000, z], [4000, -4000, z], [4000, 4000, z], [-4000, 4000, z]], np.float64)
    nv = len(f.verts_posed)
    extra_v = [floor_v]
    extra_t = [np.array([[0, 1, 2], [0, 2, 3]]) + nv]
    # slivers and zero-area triangles between existing vertices
    a = rng.integers(0, nv, 300)
    sl = np.stack([a, (a + 1) % nv, a], axis=1)                       # zero area
    sl2 = np.stack([a, (a + 1) % nv, (a + 2) % nv], axis=1)           # arbitrary thin ones
    # triangles in planes through the origin
    p = rng.normal(0, 1, (50, 3)); p /= np.linalg.norm(p, axis=1, keepdims=True)
    q = rng.normal(0, 1, (50, 3)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    through = np.stack([300.0 * p, 600.0 * p + 5.0 * q, 600.0 * p - 5.0 * q], axis=1).reshape(-1, 3)
    through[:, 2] = np.abs(through[:, 2])
    extra_v.append(through)
    extra_t.append(np.arange(150).reshape(50, 3) + nv + 4)
    verts = np.vstack([f.verts_posed] + extra_v)
    tris = np.vstack([f.tris, extra_t[0], sl, sl2, extra_t[1]]).astype(np.uint32)
    mesh = _lib.Mesh(ctx, verts, tris)
    rays = f.rays6.copy()
    at = rng.integers(0, nv, 4000)
    rays[:4000, 3:] = f.verts_posed[at].astype(np.float32)            # exactly at vertices (origin is 0)
    mid = 0.5 * (f.verts_posed[f.tris[:4000, 0]] + f.verts_posed[f.tris[:4000, 1]])
    rays[4000:8000, 3:] = mid.astype(np.float32)                      # along shared edges
    ref = oracle.raycast(verts, tris, rays)
    assert np.isfinite(ref["t_hit"]).mean() > 0.9                     # the floor catches what misses the object
    _same(_grid_cast(ctx, mesh, rays, 0), ref)


def test_grid_sweep_hands_over_what_it_cannot_answer(ctx, oracle):
    """Rays the grid cannot hold -- differing origins (status 1), directions behind the frame's plane or
    infinite (2), more than 64 rays in one cell (4) -- are completed by the exhaustive kernel in the
    same call, and the automatic choice then takes the cone culling for that ray count."""
    from pedp_hip import _lib, synth

    f = synth.Frame("parity")
    own = _lib.Context(0)
    mesh = _lib.Mesh(own, f.verts_posed, f.tris)
    n = 20000
    base = f.rays6[:: max(1, f.n_rays // n)][:n].copy()

    def check(rays, status):
        _same(_grid_cast(own, mesh, rays, status), oracle.raycast(f.verts_posed, f.tris, rays))

    check(base, 0)
    mixed = base.copy(); mixed[77, 1] += 0.5
    check(mixed, 1)
    back = base.copy(); back[5, 3:] = -back[5, 3:]
    check(back, 2)
    inf = base.copy(); inf[9, 3] = np.inf
    check(inf, 2)
    hit = int(np.flatnonzero(np.isfinite(oracle.raycast(f.verts_posed, f.tris, base)["t_hit"]))[0])
    crowd = base.copy(); crowd[100:400] = crowd[hit]                  # 300 rays in a cell that triangles visit
    check(crowd, 4)
    lonely = base.copy(); lonely[100:400] = lonely[0]                 # ... and in a corner cell nothing maps to: never walked
    assert not np.isfinite(oracle.raycast(f.verts_posed, f.tris, base[:1])["t_hit"][0])
    check(lonely, 0)
    odd = base.copy(); odd[:40, 3:] = 0.0; odd[40:60, 3:] = np.nan   # no direction: never a hit, not a reason to hand over
    check(odd, 0)
    # automatic choice: grid first, cone culling after a cast of this count that the grid handed over
    assert _lib.raycast_last_variant(own)[0] == 4
    _same(mesh.cast_rays(crowd), oracle.raycast(f.verts_posed, f.tris, crowd))
    assert _lib.raycast_last_variant(own) == (4, 4)
    _same(mesh.cast_rays(crowd), oracle.raycast(f.verts_posed, f.tris, crowd))
    assert _lib.raycast_last_variant(own)[0] == 3
    _same(mesh.cast_rays(base[:18000]), oracle.raycast(f.verts_posed, f.tris, base[:18000]))   # another count: the grid again
    assert _lib.raycast_last_variant(own) == (4, 0)


@pytest.mark.parametrize("seed", range(24))
def test_grid_sweep_fuzz_against_exhaustive(ctx, seed):
    """Random triangle soups (sizes over four decades, slivers, a few triangles through or behind the
    camera plane, some far larger than the view), random pinhole cameras: the triangle-driven sweep
    (variant 4) against the exhaustive sweep (variant 1) of the same library, every bit of t / ids / uv."""
    from pedp_hip import _lib

    rng = np.random.default_rng(1000 + seed)
    n_tri = int(rng.integers(3000, 30000))
    centre = rng.normal(0, 1, 3) * [200.0, 200.0, 0.0] + [0.0, 0.0, rng.uniform(300, 900)]
    c = centre + rng.normal(0, 1, (n_tri, 3)) * rng.uniform(30, 400)
    size = 10.0 ** rng.uniform(-1.5, 2.5, n_tri)                     # 0.03 .. 300 mm edges
    a = c + rng.normal(0, 1, (n_tri, 3)) * size[:, None]
    b = c + rng.normal(0, 1, (n_tri, 3)) * size[:, None]
    sl = rng.random(n_tri) < 0.05                                    # slivers: third corner almost on the first edge
    b[sl] = c[sl] + (a[sl] - c[sl]) * rng.uniform(0.2, 0.8, (int(sl.sum()), 1)) + rng.normal(0, 1e-4, (int(sl.sum()), 3))
    verts = np.stack([c, a, b], axis=1).reshape(-1, 3)
    k = max(1, n_tri // 200)                                         # some triangles through / behind the camera plane
    verts[: 3 * k] = rng.normal(0, 1, (3 * k, 3)) * [300.0, 300.0, 150.0]
    tris = np.arange(3 * n_tri, dtype=np.uint32).reshape(-1, 3)
    w, h = int(rng.integers(120, 320)), int(rng.integers(90, 240))
    fx = rng.uniform(0.6, 2.0) * w
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    d = np.stack([(u.ravel() - w / 2 + rng.uniform(-20, 20)) / fx, (v.ravel() - h / 2 + rng.uniform(-20, 20)) / fx, np.ones(w * h)], axis=1)
    tilt = np.linalg.qr(rng.normal(size=(3, 3)))[0] if seed % 3 == 2 else np.eye(3)   # a camera looking anywhere
    origin = rng.normal(0, 30, 3) if seed % 2 else np.zeros(3)
    if seed % 3 == 2:
        verts = verts @ tilt.T
    rays = np.hstack([np.tile(origin, (w * h, 1)), d @ tilt.T]).astype(np.float32)
    if seed % 4 == 3:
        rays = rays[rng.permutation(len(rays))[: len(rays) // 2]]    # an unordered sub-set
    mesh = _lib.Mesh(ctx, verts, tris)
    _lib.raycast_configure(ctx, 0, 1)
    try:
        ref = mesh.cast_rays(rays)
    finally:
        _lib.raycast_configure(ctx, 0, 0)
    got = _grid_cast(ctx, mesh, rays, 0)
    assert np.isfinite(ref["t_hit"]).mean() > 0.05
    _same(got, ref)


# ---- the margin of the triangle-driven ray stage, MEASURED against everything the oracle's test accepts
def _margin_slack(ctx, oracle, verts, tris, rays):
    """For every (ray, triangle) pair the oracle accepts: the ray's continuous cell coordinate lies inside the
    triangle's un-widened rectangle grown by HALF the margin the kernel applies (slack >= 2), and inside the
    cells the kernel visits.  Returns (largest used share of the margin, pairs checked, pairs with a bounded image)."""
    from pedp_hip import _lib

    mesh = _lib.Mesh(ctx, verts, tris)
    tri, ray, (GX, GY, status) = _lib.debug_rast_rects(ctx, mesh, rays)
    assert status == 0, f"the grid did not answer (status {status})"
    pairs = oracle.accepted_pairs(verts, tris, rays)
    i, f = pairs[:, 0], pairs[:, 1]
    t = tri[f]
    assert (t[:, 0] == 1).all(), "an accepted pair's triangle was dropped"
    assert (ray[i, 0] == 1).all()
    x, y = ray[i, 1].astype(np.float64), ray[i, 2].astype(np.float64)
    # the cells visited hold the ray's cell (bounded image or not)
    cx, cy = np.clip(np.floor(x), 0, GX - 1), np.clip(np.floor(y), 0, GY - 1)
    assert ((t[:, 8] <= cx) & (cx <= t[:, 9]) & (t[:, 10] <= cy) & (cy <= t[:, 11])).all()
    b = t[:, 1] == 0                                   # bounded image: rectangle + margin
    ex = np.maximum(np.maximum(t[b, 2] - x[b], x[b] - t[b, 3]), 0.0)
    ey = np.maximum(np.maximum(t[b, 4] - y[b], y[b] - t[b, 5]), 0.0)
    used = np.maximum(ex / t[b, 6], ey / t[b, 7]) if b.any() else np.zeros(0)
    worst = float(used.max()) if len(used) else 0.0
    assert worst <= 0.5, f"an accepted ray uses {worst:.3f} of the margin: less than a factor two is left"
    return worst, len(pairs), int(b.sum())


def _pinhole(rng, w, h, fov_scale, origin=None, tilt=None):
    fx = fov_scale * w
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    d = np.stack([(u.ravel() - w / 2 + 0.37) / fx, (v.ravel() - h / 2 - 0.21) / fx, np.ones(w * h)], axis=1)
    if tilt is not None:
        d = d @ tilt.T
    o = np.zeros(3) if origin is None else origin
    return np.hstack([np.tile(o, (w * h, 1)), d]).astype(np.float32)


def _soup_case(seed):
    """A fuzz soup (sizes over four decades, slivers, triangles across the camera plane) at a size the oracle can test
    pair by pair, and a pinhole camera's rays (every third one tilted, every second one off the origin)."""
    rng = np.random.default_rng(4000 + seed)
    n_tri = 700
    centre = rng.normal(0, 1, 3) * [150.0, 150.0, 0.0] + [0.0, 0.0, rng.uniform(300, 900)]
    c = centre + rng.normal(0, 1, (n_tri, 3)) * rng.uniform(30, 300)
    size = 10.0 ** rng.uniform(-1.5, 2.5, n_tri)
    a = c + rng.normal(0, 1, (n_tri, 3)) * size[:, None]
    b = c + rng.normal(0, 1, (n_tri, 3)) * size[:, None]
    sl = rng.random(n_tri) < 0.2                                     # slivers, the sharp corner anywhere
    b[sl] = c[sl] + (a[sl] - c[sl]) * rng.uniform(-0.2, 1.2, (int(sl.sum()), 1)) + rng.normal(0, 1e-4, (int(sl.sum()), 3))
    verts = np.stack([c, a, b], axis=1).reshape(-1, 3)
    verts[:12] = rng.normal(0, 1, (12, 3)) * [300.0, 300.0, 150.0]   # through / behind the camera plane
    tris = np.arange(3 * n_tri, dtype=np.uint32).reshape(-1, 3)
    tilt = np.linalg.qr(rng.normal(size=(3, 3)))[0] if seed % 3 == 2 else None
    origin = rng.normal(0, 30, 3) if seed % 2 else None
    if tilt is not None:
        verts = verts @ tilt.T
    rays = _pinhole(rng, 168, 110, rng.uniform(0.35, 1.6), origin, tilt)     # up to ~55 degrees off axis
    return verts.astype(np.float32), tris, rays


@pytest.mark.parametrize("seed", range(6))
def test_grid_margin_on_triangle_soups(ctx, oracle, seed):
    """The fuzz soups of the sweep test at a size the oracle can test pair by pair: every accepted pair is inside its
    rectangle with a factor >= 2 to spare."""
    verts, tris, rays = _soup_case(seed)
    worst, n_pairs, n_bounded = _margin_slack(ctx, oracle, verts, tris, rays)
    print(f"soup {seed}: {n_pairs} accepted pairs, {n_bounded} with a bounded image, largest used share of the margin {worst:.3f}")
    assert n_pairs > 300 and n_bounded > 0.3 * n_pairs    # (the triangles across the camera plane have no bounded image and meet many rays)


def _adversarial_cases():
    """Triangles seen almost edge-on (the origin a hair off their plane), needles whose sharp corner is NOT the record's
    first vertex, slivers of the size of the view, big triangles whose first vertex is far away while the rays meet them
    close by; three cameras, one wide-angle, one off the origin.  Yields (verts, tris, rays, fov)."""
    rng = np.random.default_rng(77)
    tri_list = []

    def add(p0, p1, p2):
        tri_list.append(np.array([p0, p1, p2], np.float64))

    for k in range(150):                                            # edge-on: planes through a point a hair off the origin
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        off = 10.0 ** rng.uniform(-7, -2) * rng.choice([-1, 1])
        base = np.array([rng.uniform(-200, 200), rng.uniform(-150, 150), rng.uniform(300, 700)])
        e = np.cross(n, rng.normal(size=3)); e /= np.linalg.norm(e)
        g = np.cross(n, e)
        q0 = base - n * (base @ n) + n * off * np.linalg.norm(base)         # the plane passes the origin at `off` radians
        s = rng.uniform(5, 200)
        add(q0, q0 + e * s, q0 + g * s * rng.uniform(0.2, 1.0))
    for k in range(150):                                            # needles: long first edge, tiny second, sharp corner at v1 / v2
        p0 = np.array([rng.uniform(-250, 250), rng.uniform(-180, 180), rng.uniform(250, 800)])
        d1 = rng.normal(size=3); d1 /= np.linalg.norm(d1)
        d2 = np.cross(d1, rng.normal(size=3)); d2 /= np.linalg.norm(d2)
        L, w = rng.uniform(50, 600), 10.0 ** rng.uniform(-4, -0.5)
        add(p0, p0 + d1 * L, p0 + d2 * w) if k % 2 else add(p0, p0 + d2 * w, p0 + d1 * L)
    for k in range(60):                                             # view-sized slivers across the frame
        z = rng.uniform(300, 600)
        y = rng.uniform(-200, 200)
        add([-600, y, z], [600, y + rng.uniform(-30, 30), z + rng.uniform(-50, 50)], [0, y + 10.0 ** rng.uniform(-4, -1), z])
    for k in range(60):                                             # floors: first vertex far away, hit close by
        h = rng.uniform(20, 150)
        far = np.array([rng.uniform(-3e4, 3e4), h, rng.uniform(2e4, 9e4)])
        add(far, [-800, h + rng.uniform(-5, 5), 60], [900, h + rng.uniform(-5, 5), 80])
    verts = np.concatenate(tri_list).astype(np.float32)
    tris = np.arange(len(verts), dtype=np.uint32).reshape(-1, 3)
    for fov, origin in ((0.9, None), (0.3, None), (0.45, rng.normal(0, 20, 3))):   # 0.3: +-59 degrees across
        yield verts, tris, _pinhole(rng, 168, 110, fov, origin), fov


def test_grid_margin_on_adversarial_triangles(ctx, oracle):
    """What the margin's formula is about (the sets of _adversarial_cases): every accepted pair inside with >= 2x."""
    for verts, tris, rays, fov in _adversarial_cases():
        worst, n_pairs, n_bounded = _margin_slack(ctx, oracle, verts, tris, rays)
        print(f"adversarial, fov scale {fov}: {n_pairs} accepted pairs, {n_bounded} bounded, largest used share of the margin {worst:.3f}")
        assert n_pairs > 300


def _filter_slack(ctx, oracle, verts, tris, rays):
    """The exhaustive sweep's matrix-pipe filter (variant 1, shared origin) against EVERY pair the oracle accepts: the
    pair's score is >= 0 (it reaches the exact test), and what the bf16 pipe made of the dot product alone lies less than
    HALF the slack below zero.  Returns (largest used share of the slack, accepted pairs, pairs that pass the filter)."""
    from pedp_hip import _lib

    mesh = _lib.Mesh(ctx, verts, tris)
    score, slack = _lib.debug_mfma_scores(ctx, mesh, rays)
    pairs = oracle.accepted_pairs(verts, tris, rays)
    s, sl = score[pairs[:, 0], pairs[:, 1]].astype(np.float64), slack[pairs[:, 0], pairs[:, 1]].astype(np.float64)
    assert (s >= 0).all() and not np.signbit(s).any(), f"the filter rejects {int(np.signbit(s).sum())} pair(s) the oracle accepts"
    real = np.isfinite(sl) & (sl < 1e20) & (sl > 0)                 # (records / rays sent straight to the exact test carry 1e30)
    used = np.maximum(sl[real] - s[real], 0.0) / sl[real]
    worst = float(used.max()) if real.any() else 0.0
    assert worst <= 0.5, f"an accepted pair uses {worst:.3f} of the slack: less than a factor two is left"
    return worst, len(pairs), int((~np.signbit(score)).sum())


@pytest.mark.parametrize("seed", range(6))
def test_mfma_filter_on_triangle_soups(ctx, oracle, seed):
    """Variant 1's bf16 MFMA filter on the soups: no accepted pair is rejected, half the slack is never used, and the
    pairs that pass are few (the exact test that follows them is the rare branch)."""
    verts, tris, rays = _soup_case(seed)
    if seed % 2:                                                    # (the filter serves rays of ONE origin: these soups' cameras sit off it)
        rays[:, :3] = rays[0, :3]
    worst, n_pairs, n_pass = _filter_slack(ctx, oracle, verts, tris, rays)
    print(f"soup {seed}: {n_pairs} accepted pairs, {n_pass} pass the filter ({n_pass / max(n_pairs, 1):.3f}x), largest used share of the slack {worst:.4f}")
    assert n_pairs > 300 and n_pass < 3 * n_pairs + 50000           # (triangles in a plane through the origin pass for every ray)


def test_mfma_filter_on_adversarial_triangles(ctx, oracle):
    for verts, tris, rays, fov in _adversarial_cases():
        worst, n_pairs, n_pass = _filter_slack(ctx, oracle, verts, tris, rays)
        print(f"adversarial, fov scale {fov}: {n_pairs} accepted pairs, {n_pass} pass the filter, largest used share of the slack {worst:.4f}")
        assert n_pairs > 300


def test_mfma_filter_sends_hostile_operands_to_the_exact_test(ctx, oracle):
    """Rays without a finite direction, astronomically large triangles, zero-area triangles, triangles in a plane through
    the origin: the pipe never sees operands whose products could overflow, the results are the oracle's."""
    from pedp_hip import _lib, synth

    f = synth.Frame("tiny")
    verts = f.verts_posed.copy()
    tris = f.tris.copy()
    verts = np.vstack([verts, np.array([[1e20, 0, 1e3], [0, 1e20, 1e3], [-1e20, -1e20, 1e3],      # a triangle the size of a galaxy
                                        [5, 5, 50], [5, 5, 50], [5, 5, 50],                        # zero area
                                        [0, 0, 0], [10, 0, 100], [0, 10, 100]], np.float32)])      # in a plane through the origin
    n0 = len(f.verts_posed)
    tris = np.vstack([tris, np.array([[n0, n0 + 1, n0 + 2], [n0 + 3, n0 + 4, n0 + 5], [n0 + 6, n0 + 7, n0 + 8]], tris.dtype)])
    rays = f.rays6.copy()
    rays[5, 3:] = np.nan
    rays[6, 3:] = (np.inf, 0, 1)
    rays[7, 3:] = 0.0
    rays[8, 3:] = (3e30, 1e30, 2e30)
    mesh = _lib.Mesh(ctx, verts, tris)
    _lib.raycast_configure(ctx, 0, 1)
    try:
        got = mesh.cast_rays(rays)
    finally:
        _lib.raycast_configure(ctx, 0, 0)
    ref = oracle.raycast(verts, tris, rays)
    assert np.array_equal(got["primitive_ids"], ref["primitive_ids"])
    assert np.array_equal(got["t_hit"].view(np.uint32), ref["t_hit"].view(np.uint32))


def test_grid_margin_degenerate_records_cost_nothing(ctx, oracle):
    """Zero-area triangles (m == 0 exactly: det is 0 for every ray) are dropped, not tested against every cell."""
    from pedp_hip import _lib, synth

    f = synth.Frame("parity")
    v = np.concatenate([f.verts_posed, np.repeat(f.verts_posed[:1], 3, 0)])
    n = len(f.verts_posed)
    dead = np.array([[n, n + 1, n + 2]] * 50 + [[5, 5, 9]] * 50, np.uint32)      # coincident corners, repeated index
    tris = np.concatenate([f.tris, dead])
    mesh = _lib.Mesh(ctx, v, tris)
    tri, ray, (GX, GY, status) = _lib.debug_rast_rects(ctx, mesh, f.rays6)
    assert status == 0 and (tri[len(f.tris):, 0] == 0).all()
    got = _grid_cast(ctx, mesh, f.rays6, 0)
    _same(got, oracle.raycast(v, tris, f.rays6, bvh=True))


def test_resident_ray_set_equals_the_per_call_cast(ctx, oracle):
    """pedp_rayset_*: a camera's rays resident, the grid's chains built once -- every cast against the set returns the
    oracle's bits, cast after cast (the keys are re-armed by the result kernel), against different meshes and poses,
    with and without uv, into host and device memory; rays the grid cannot serve are cast by the other variants."""
    torch = pytest.importorskip("torch")
    from pedp_hip import _lib, synth

    f = synth.Frame("parity")
    rs = _lib.RaySet(ctx, f.rays6)
    mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
    ref = oracle.raycast(f.verts_posed, f.tris, f.rays6, bvh=True)
    for rep in range(3):
        got = mesh.cast_rayset(rs, want_uv=(rep != 1))
        assert rs.last_variant() == (4, 0)
        assert np.array_equal(got["primitive_ids"], ref["primitive_ids"]) and np.array_equal(got["t_hit"].view(np.uint32), ref["t_hit"].view(np.uint32))
        if rep != 1:
            assert np.array_equal(got["primitive_uvs"].view(np.uint32), ref["primitive_uvs"].view(np.uint32))
    # another mesh (moved: other hits, other misses), then the first one again
    moved = (f.verts_posed + np.array([12.0, -7.0, 25.0], np.float32)).astype(np.float32)
    mesh2 = _lib.Mesh(ctx, moved, f.tris)
    ref2 = oracle.raycast(moved, f.tris, f.rays6, bvh=True)
    got2 = mesh2.cast_rayset(rs)
    assert np.array_equal(got2["primitive_ids"], ref2["primitive_ids"]) and np.array_equal(got2["t_hit"].view(np.uint32), ref2["t_hit"].view(np.uint32))
    got = mesh.cast_rayset(rs)
    assert np.array_equal(got["primitive_ids"], ref["primitive_ids"]) and (got2["primitive_ids"] != got["primitive_ids"]).any()
    # device in, device out
    d_rays = torch.from_numpy(f.rays6).cuda()
    rs_dev = _lib.RaySet(ctx, device_ptr=d_rays.data_ptr(), n=len(f.rays6))
    del d_rays                                               # (the set owns its copy)
    t = torch.empty(len(f.rays6), dtype=torch.float32, device="cuda")
    ids = torch.empty(len(f.rays6), dtype=torch.int32, device="cuda")
    mesh.cast_rayset_device(rs_dev, t.data_ptr(), ids.data_ptr())
    ctx.synchronize()
    assert np.array_equal(t.cpu().numpy().view(np.uint32), ref["t_hit"].view(np.uint32)) and np.array_equal(ids.cpu().numpy().view(np.uint32), ref["primitive_ids"])
    # rays the grid cannot serve: origins that differ -- still the oracle's bits, by another variant
    mixed = f.rays6.copy()
    mixed[::3, :3] += np.array([5.0, -3.0, 2.0], np.float32)
    rs_mixed = _lib.RaySet(ctx, mixed)
    refm = oracle.raycast(f.verts_posed, f.tris, mixed, bvh=True)
    gotm = mesh.cast_rayset(rs_mixed)
    assert rs_mixed.last_variant()[0] != 4
    assert np.array_equal(gotm["primitive_ids"], refm["primitive_ids"]) and np.array_equal(gotm["t_hit"].view(np.uint32), refm["t_hit"].view(np.uint32))
    # a posable mesh under changing poses, the set unchanged
    pm = _lib.Mesh(ctx, f.model_points, f.tris, posable=True)
    for k in range(3):
        T = f.T_gt.copy()
        T[:3, 3] += (3.0 * k, -2.0 * k, 5.0 * k)
        pm.set_pose(T)
        refp = oracle.raycast(oracle.pose_vertices(T, f.model_points), f.tris, f.rays6, bvh=True)
        gotp = pm.cast_rayset(rs, want_uv=False)
        assert np.array_equal(gotp["primitive_ids"], refp["primitive_ids"]) and np.array_equal(gotp["t_hit"].view(np.uint32), refp["t_hit"].view(np.uint32))
