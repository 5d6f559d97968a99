"""GPU parity of the depth pre-filters against the oracle (oracle/depth.c): erode
and both back-projections bit-exact (NaN positions included); the bilateral filter within 2e-6
relative (its exp comes from two different libm implementations, everything else is the same
float32 operation sequence).  All through the C ABI, host-memory and device-memory modes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BILATERAL_RTOL = 2e-6


def _same(a, b):
    assert a.shape == b.shape and a.dtype == b.dtype
    assert np.array_equal(a, b, equal_nan=True)


def _close(a, b):
    assert a.shape == b.shape
    assert np.array_equal(np.isnan(a), np.isnan(b))
    assert np.array_equal(a == 0, b == 0)
    m = ~np.isnan(a)
    assert np.all(np.abs(a[m] - b[m]) <= BILATERAL_RTOL * np.abs(b[m]))


def test_golden_fixture(ctx):
    from pedp_hip import compat

    g = np.load(os.path.join(GOLD, "g7_depth_filters.npz"))
    _same(compat.erode_depth(g["depth"]), g["erode"])
    _same(compat.erode_depth(g["depth"], 3, 0.002, 0.5, 1.0), g["erode_r3"])
    _close(compat.bilateral_filter_depth(g["depth"]), g["bilateral"])
    _close(compat.bilateral_filter_depth(g["depth"], 1, 1.0, 1.5, 0.02), g["bilateral_r1"])
    _same(compat.depth2xyzmap(g["depth"], g["K"]), g["xyz"])
    _same(compat.depth2xyzmap_batch(g["depths"], g["Ks"], 0.9), g["xyz_batch"])


@pytest.mark.parametrize("shape", [(1, 1), (3, 200), (17, 65), (64, 64), (145, 131), (576, 640)])
@pytest.mark.parametrize("radius", [0, 1, 2, 3, 4, 6])
def test_stencils_match_oracle(ctx, oracle, shape, radius):
    from pedp_hip import compat, synth

    if shape == (576, 640) and radius not in (2, 6):
        pytest.skip("full frame: default radius and the generic path only")
    d = synth.depth_image(*shape, seed=radius + shape[1])
    _same(compat.erode_depth(d, radius), oracle.erode_depth(d, radius))
    _same(compat.erode_depth(d, radius, 0.004, 0.3, 0.9), oracle.erode_depth(d, radius, 0.004, 0.3, 0.9))
    _close(compat.bilateral_filter_depth(d, radius), oracle.bilateral_filter_depth(d, radius))
    _close(compat.bilateral_filter_depth(d, radius, 0.8, 1.0, 0.005), oracle.bilateral_filter_depth(d, radius, 0.8, 1.0, 0.005))


@pytest.mark.parametrize("radius", [1, 2, 4])
def test_stencils_on_a_large_image(ctx, oracle, radius):
    """Images of a million pixels and more take the wide tiles (256 x 16, dealt to the XCDs in contiguous bands); sizes
    that are no multiple of the tile, invalid readings, NaN and infinities included."""
    from pedp_hip import compat, synth

    d = np.tile(synth.depth_image(576, 640, seed=11 + radius), (2, 2))[:1100, :1111].copy()
    assert d.size >= 1 << 20
    d[5, 7] = np.inf; d[700, 1110] = -np.inf; d[1099, 0] = np.nan; d[300:303, 255:258] = 0.0
    _same(compat.erode_depth(d, radius), oracle.erode_depth(d, radius))
    _same(compat.erode_depth(d, radius, 0.004, 0.3, 0.9), oracle.erode_depth(d, radius, 0.004, 0.3, 0.9))
    _close(compat.bilateral_filter_depth(d, radius), oracle.bilateral_filter_depth(d, radius))
    _close(compat.bilateral_filter_depth(d, radius, 0.8, 1.0, 0.005), oracle.bilateral_filter_depth(d, radius, 0.8, 1.0, 0.005))


@pytest.mark.parametrize("shape", [(1, 1 << 20), (1 << 20, 1), (3, 350000), (70000, 15)])
def test_band_walker_on_extreme_shapes(ctx, oracle, shape):
    """The large-image erode kernel on images that are one band wide, one block high, or thinner than its halo."""
    from pedp_hip import compat, synth

    base = synth.depth_image(576, 640, seed=5)
    d = np.resize(base, shape).astype(np.float32)
    _same(compat.erode_depth(d, 2), oracle.erode_depth(d, 2))
    _same(compat.erode_depth(d, 3, 0.004, 0.3, 0.9), oracle.erode_depth(d, 3, 0.004, 0.3, 0.9))


def test_back_projection_and_chain(ctx, oracle):
    from pedp_hip import compat, synth

    d = synth.depth_image(576, 640, seed=1)
    K = np.array([[504.0, 0, 319.5], [0, 503.0, 287.5], [0, 0, 1]])
    _same(compat.depth2xyzmap(d, K), oracle.depth2xyzmap(d, K))
    ds = np.stack([d, synth.depth_image(576, 640, seed=2), synth.depth_image(576, 640, seed=3, nan=False)])
    Ks = np.stack([K, K * 1.01, K * 0.97]).astype(np.float32)
    for zfar in (np.inf, 0.8):
        _same(compat.depth2xyzmap_batch(ds, Ks, zfar), oracle.depth2xyzmap_batch(ds, Ks, zfar))
    # estimater.py:255-259: erode -> bilateral -> batched back-projection
    e = compat.erode_depth(d, radius=2, device="cuda")
    b = compat.bilateral_filter_depth(e, radius=2, device="cuda")
    _close(b, oracle.bilateral_filter_depth(oracle.erode_depth(d)))
    one = compat.depth2xyzmap(d, K, uvs=np.array([[7.4, 11.6]]))       # pixel (7, 12) only
    assert np.count_nonzero(one.any(axis=2)) <= 1 and np.array_equal(one[12, 7], compat.depth2xyzmap(d, K)[12, 7])
    assert compat.erode_depth(np.zeros((0, 0), np.float32)).shape == (0, 0)


def test_device_tensors_stay_on_device(ctx, oracle):
    """torch tensors on the GPU go through data_ptr() in PEDP_DEVICE mode (estimater.py hands
    CUDA tensors to these functions)."""
    import torch
    from pedp_hip import compat, synth

    d = synth.depth_image(145, 131, seed=4)
    t = torch.from_numpy(d).cuda()
    e = compat.erode_depth(t, radius=2)
    assert e.is_cuda and e.dtype == torch.float32
    _same(e.cpu().numpy(), oracle.erode_depth(d))
    _close(compat.bilateral_filter_depth(e, radius=2).cpu().numpy(), oracle.bilateral_filter_depth(oracle.erode_depth(d)))
    K = np.array([[120.0, 0, 65.0], [0, 121.0, 72.0], [0, 0, 1]])
    x = compat.depth2xyzmap_batch(t[None], torch.as_tensor(K, dtype=torch.float32, device="cuda")[None], zfar=np.inf)
    assert x.is_cuda and tuple(x.shape) == (1, 145, 131, 3)
    _same(x.cpu().numpy(), oracle.depth2xyzmap_batch(d[None], K[None].astype(np.float32), np.inf))
    _same(compat.depth2xyzmap(t, K).cpu().numpy(), oracle.depth2xyzmap(d, K))


def test_depth2xyzmap_with_pixel_list(ctx, oracle):
    """depth2xyzmap(depth, K, uvs) (Utils.py:406-409): only the listed (u, v) pixels, rounded, are
    back-projected; the reference's numpy statement gives the expected map."""
    from pedp_hip import compat, synth

    d = synth.depth_image(40, 48, seed=2, nan=False)
    K = np.array([[50.0, 0, 23.5], [0, 52.0, 19.5], [0, 0, 1]])
    rng = np.random.default_rng(0)
    uvs = np.stack([rng.uniform(0, 47, 60), rng.uniform(0, 39, 60)], axis=1)
    got = compat.depth2xyzmap(d, K, uvs, ctx=ctx)
    us, vs = uvs.round().astype(int)[:, 0], uvs.round().astype(int)[:, 1]
    zs = d[vs, us]
    pts = np.stack(((us - K[0, 2]) * zs / K[0, 0], (vs - K[1, 2]) * zs / K[1, 1], zs), 1)
    ref = np.zeros((40, 48, 3), np.float32)
    ref[vs, us] = pts
    ref[d < 0.001] = 0
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_depth_to_scene_in_one_call_equals_the_three_calls(ctx, oracle):
    """pedp_depth_to_scene (a frame's depth entry: erode -> bilateral -> depth2xyzmap_batch -> the valid points in mm) against
    the single calls and torch's mask expression: the filtered image, the xyz map and the float64 points in every bit, from
    a host image and from a CUDA tensor, frame after frame (the point buffers are taken in turn), odd sizes, no valid pixel."""
    torch = pytest.importorskip("torch")
    from pedp_hip import compat, depth_filters, synth

    K = np.array([[504.0, 0, 319.5], [0, 504.0, 287.5], [0, 0, 1]], np.float32)
    buffers = {}
    keep = []
    for k, (h, w) in enumerate([(576, 640), (576, 640), (61, 83), (576, 640)]):
        depth = synth.depth_image(h, w, seed=10 + k, nan=False).astype(np.float32)
        src = torch.from_numpy(depth).cuda() if k == 1 else depth
        d, xyz, pts = depth_filters.depth_to_scene(src, K, buffers=buffers)
        e = compat.erode_depth(torch.from_numpy(depth).cuda(), radius=2, device="cuda")
        b = compat.bilateral_filter_depth(e, radius=2, device="cuda")
        x = compat.depth2xyzmap_batch(b[None], torch.from_numpy(K).cuda()[None], zfar=np.inf)[0]
        ref = x[x[..., 2] >= 0.001].double() * 1000.0
        assert torch.equal(d, b) and torch.equal(xyz, x)
        assert pts.dtype == torch.float64 and pts.shape == ref.shape and torch.equal(pts, ref) and len(pts) > 100
        keep.append((pts, ref.clone()))
    assert torch.equal(keep[2][0], keep[2][1])          # the cloud of the frame before the last is intact (two buffers in turn)
    d, xyz, pts = depth_filters.depth_to_scene(np.zeros((40, 48), np.float32), K, buffers=buffers)
    assert len(pts) == 0 and not xyz.any()
