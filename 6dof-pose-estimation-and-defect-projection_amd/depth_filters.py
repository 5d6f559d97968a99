"""Host mirror of the reference's depth pre-filters (Utils.py): same names, arguments and return
types; the pixels are processed by libpedp_hip.so instead of warp-lang's CUDA JIT.

    erode_depth             Utils.py:386-396  (kernel :356-383; callers estimater.py:171, :255)
    bilateral_filter_depth  Utils.py:347-357  (kernel :304-345; callers estimater.py:172, :256)
    depth2xyzmap            Utils.py:401-420  (callers run.py:89, estimater.py:175, :212)
    depth2xyzmap_batch      Utils.py:423-442  (caller estimater.py:259)

numpy in -> numpy out; a torch tensor on the GPU in -> a torch tensor on the same device out,
processed in place on the device through its data_ptr() (no host round trip).  `device` is
accepted for signature compatibility; the context's GPU is used.
"""
import ctypes as C

import numpy as np

from . import _lib


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _run(fn_name, depth, out_shape_tail, call, ctx=None):
    """Dispatch one image-shaped call on host (numpy) or device (torch) memory."""
    lib = _lib.load()
    if _is_torch(depth):
        import torch

        if not depth.is_cuda:
            return torch.from_numpy(_run(fn_name, depth.numpy(), out_shape_tail, call, ctx))
        d = depth.contiguous().to(torch.float32)
        out = torch.empty(tuple(d.shape) + out_shape_tail, dtype=torch.float32, device=d.device)
        cur = torch.cuda.current_stream(d.device)
        if ctx is None:
            ctx = _stream_context(d.device.index or 0, cur.cuda_stream)
        shared = ctx.stream_handle not in (None, 0) and ctx.stream_handle == cur.cuda_stream
        if not shared:
            cur.synchronize()  # the context runs on another stream: the input must be complete
        _lib.check(call(lib, ctx._h, C.c_void_p(d.data_ptr()), _lib.DEVICE, C.c_void_p(out.data_ptr()), d.shape), fn_name)
        if not shared:
            ctx.synchronize()  # ... and the output before torch touches it
        return out
    d = np.ascontiguousarray(depth, dtype=np.float32)
    ctx = ctx or _lib.default_context()
    out = np.empty(d.shape + out_shape_tail, np.float32)
    _lib.check(call(lib, ctx._h, _lib._ptr(d), _lib.HOST, _lib._ptr(out), d.shape), fn_name)
    return out


_stream_ctx = {}


def _stream_context(device, stream_handle):
    """A context per (device, torch stream).  torch's default stream has handle 0, which the C
    ABI reads as "own stream": such a context is not ordered with torch and _run synchronises
    around the call instead."""
    key = (device, stream_handle)
    if key not in _stream_ctx:
        _stream_ctx[key] = _lib.Context(device, stream=stream_handle or None)
    return _stream_ctx[key]


def _hw(shape):
    if len(shape) != 2:
        raise _lib.PedpError(f"expected an H x W depth image, got shape {tuple(shape)}")
    return int(shape[0]), int(shape[1])


def erode_depth(depth, radius=2, depth_diff_thres=0.001, ratio_thres=0.8, zfar=100, device="cuda", ctx=None):
    def call(lib, h, src, mem, dst, shape):
        H, W = _hw(shape)
        return lib.pedp_erode_depth(h, src, H, W, int(radius), float(depth_diff_thres), float(ratio_thres), float(zfar),
                                    mem, dst)
    return _run("pedp_erode_depth", depth, (), call, ctx)


def bilateral_filter_depth(depth, radius=2, zfar=100, sigmaD=2, sigmaR=100000, device="cuda", ctx=None):
    def call(lib, h, src, mem, dst, shape):
        H, W = _hw(shape)
        return lib.pedp_bilateral_filter_depth(h, src, H, W, int(radius), float(zfar), float(sigmaD), float(sigmaR),
                                               mem, dst)
    return _run("pedp_bilateral_filter_depth", depth, (), call, ctx)


def depth2xyzmap(depth, K, uvs=None, ctx=None):
    if uvs is not None:
        # Utils.py:406-409: only the listed pixels (u, v rounded) are back-projected, the rest of the
        # map stays zero.  A listed pixel's value is the full map's, so the full map comes from the
        # device and the listed pixels are picked out of it.
        full = depth2xyzmap(depth, K, None, ctx)
        on_device = _is_torch(full)
        arr = full.detach().cpu().numpy() if on_device else np.asarray(full)
        uv = np.asarray(uvs.detach().cpu().numpy() if _is_torch(uvs) else uvs).round().astype(int)
        out = np.zeros_like(arr)
        out[uv[:, 1], uv[:, 0]] = arr[uv[:, 1], uv[:, 0]]
        if on_device:
            import torch

            return torch.from_numpy(out).to(full.device)
        return out
    Kd = np.ascontiguousarray(np.asarray(K.detach().cpu().numpy() if _is_torch(K) else K), dtype=np.float64).reshape(3, 3)

    def call(lib, h, src, mem, dst, shape):
        H, W = _hw(shape)
        return lib.pedp_depth2xyzmap(h, src, H, W, _lib._ptr(Kd), mem, dst)
    return _run("pedp_depth2xyzmap", depth, (3,), call, ctx)


def depth2xyzmap_batch(depths, Ks, zfar, ctx=None):
    Kf = np.ascontiguousarray(np.asarray(Ks.detach().cpu().numpy() if _is_torch(Ks) else Ks), dtype=np.float32)

    def call(lib, h, src, mem, dst, shape):
        if len(shape) != 3:
            raise _lib.PedpError(f"expected B x H x W depths, got shape {tuple(shape)}")
        B, H, W = (int(v) for v in shape)
        if Kf.size != 9 * B:
            raise _lib.PedpError("Ks must hold one 3 x 3 matrix per image")
        return lib.pedp_depth2xyzmap_batch(h, src, B, H, W, _lib._ptr(Kf), float(zfar), mem, dst)
    return _run("pedp_depth2xyzmap_batch", depths, (3,), call, ctx)


def depth_to_scene(depth, K, erode_radius=2, bilateral_radius=2, depth_diff_thres=0.001, ratio_thres=0.8, erode_zfar=100,
                   bilateral_zfar=100, sigmaD=2, sigmaR=100000, xyz_zfar=np.inf, z_min=0.001, scale=1000.0, buffers=None, ctx=None):
    """A frame's depth entry in ONE library call (pedp_depth_to_scene): erode_depth -> bilateral_filter_depth ->
    depth2xyzmap_batch -> the points with z >= z_min as float64, scaled (estimater.py:255-259 and the scene cloud of run.py's
    loop; what `xyz[xyz[..., 2] >= z_min].double() * scale` gives after the three calls, bit for bit).  depth: H x W float32
    metres, a numpy array (uploaded by the library through its pinned staging) or a CUDA tensor.  Returns torch CUDA tensors
    (filtered depth H x W, xyz map H x W x 3, points n x 3 float64); `buffers` (a dict, optional) keeps them across frames."""
    import torch

    ctx = ctx or _lib.default_context()
    dev = torch.device(f"cuda:{ctx.device}")
    if _is_torch(depth):
        d_in = depth.to(dev, torch.float32).contiguous()
        torch.cuda.current_stream(dev).synchronize()            # the library reads it on its own stream
        H, W = _hw(d_in.shape)
        src, mem, keep = C.c_void_p(d_in.data_ptr()), _lib.DEVICE, d_in
    else:
        d_in = np.ascontiguousarray(depth, dtype=np.float32)
        H, W = _hw(d_in.shape)
        src, mem, keep = _lib._ptr(d_in), _lib.HOST, d_in
    buffers = {} if buffers is None else buffers
    if buffers.get("shape") != (H, W):
        buffers["shape"] = (H, W)
        buffers["depth"] = torch.empty((H, W), dtype=torch.float32, device=dev)
        buffers["xyz"] = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
        buffers["points"] = [torch.empty((H * W, 3), dtype=torch.float64, device=dev) for _ in range(2)]
        buffers["turn"] = 0
        torch.cuda.current_stream(dev).synchronize()
    buffers["turn"] ^= 1                                          # two point buffers in turn: a frame's cloud outlives the next call
    pts = buffers["points"][buffers["turn"]]
    Kf = np.asarray(K.detach().cpu().numpy() if _is_torch(K) else K, dtype=np.float32).reshape(9)
    prm = _lib.DepthEntryParams(int(erode_radius), float(depth_diff_thres), float(ratio_thres), float(erode_zfar), int(bilateral_radius),
                                float(bilateral_zfar), float(sigmaD), float(sigmaR), (C.c_float * 9)(*Kf.tolist()), float(xyz_zfar),
                                float(z_min), float(scale))
    n = C.c_int64()
    _lib.check(_lib.load().pedp_depth_to_scene(ctx._h, src, H, W, mem, C.byref(prm), C.c_void_p(buffers["depth"].data_ptr()),
                                               C.c_void_p(buffers["xyz"].data_ptr()), C.c_void_p(pts.data_ptr()), C.byref(n)),
               "pedp_depth_to_scene")
    del keep
    return buffers["depth"], buffers["xyz"], pts[: n.value]
