import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from pedp_hip import _lib, compat, synth
dev = torch.device("cuda:0"); stream = torch.cuda.Stream(device=dev)
ctx = _lib.Context(0, stream=stream.cuda_stream); torch.cuda.set_stream(stream)
base = synth.depth_image(576, 640, seed=0, nan=False)
h = w = 8192
full = np.tile(base, (-(-h // 576), -(-w // 640)))[:h, :w].copy()
variants = {"stress image (5% holes, 1% sub-threshold, 1% beyond zfar)": full,
            "holes only (beyond-zfar and sub-threshold readings -> 0)": np.where((full > 100) | (full < 0.001), 0, full).astype(np.float32),
            "no invalid readings at all": np.where((full > 100) | (full < 0.001), 0.7, full).astype(np.float32)}
for name, img in variants.items():
    d = torch.from_numpy(img).to(dev)
    for fn_name, fn in [("erode", lambda: compat.erode_depth(d, 2, ctx=ctx)), ("bilateral", lambda: compat.bilateral_filter_depth(d, 2, ctx=ctx))]:
        for _ in range(3): fn()
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        ctx.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / 20
        print(f"{name:62s} {fn_name:10s} {1e3*ms:8.1f} us  {8*h*w/ms/1e6/8000:.3f} of 8 TB/s", flush=True)
