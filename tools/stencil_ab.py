"""A/B timing of the depth stencils at 4096^2 and 8192^2 (PEDP_LIB selects another build of the library):
python tools/stencil_ab.py [erode|bilateral]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pedp_hip import _lib, compat, synth
dev = torch.device("cuda:0"); stream = torch.cuda.Stream(device=dev)
ctx = _lib.Context(0, stream=stream.cuda_stream); torch.cuda.set_stream(stream)
base = synth.depth_image(576, 640, seed=0, nan=False)
which = sys.argv[1:] or ["erode", "bilateral"]
for (h, w) in [(4096, 4096), (8192, 8192)]:
    d = torch.from_numpy(np.tile(base, (-(-h // 576), -(-w // 640)))[:h, :w].copy()).to(dev)
    for name, fn in [("erode", lambda: compat.erode_depth(d, 2, ctx=ctx)), ("bilateral", lambda: compat.bilateral_filter_depth(d, 2, ctx=ctx))]:
        if name not in which:
            continue
        for _ in range(3): fn()
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(30): fn()
        ctx.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / 30
        print(os.environ.get("PEDP_LIB", "tree")[-20:], os.environ.get("PEDP_WALK_WGS", ""), h, name, f"{1e3*ms:.1f} us", flush=True)
