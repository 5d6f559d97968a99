"""Timing of the depth pre-filters on device-resident images (torch tensors, PEDP_DEVICE mode):
python tools/depth_only.py.  GB/s = compulsory bytes (4 in + 4 out per pixel; 4 + 12 for the
back-projection) / wall time per call over 50 back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pedp_hip import _lib, compat, synth

dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)          # one explicit stream shared by torch and the library:
ctx = _lib.Context(0, stream=stream.cuda_stream)  # the calls below only enqueue, no host synchronisation
torch.cuda.set_stream(stream)
K = np.array([[504.0, 0, 319.5], [0, 504.0, 287.5], [0, 0, 1]])
for (h, w) in [(576, 640), (720, 1280), (8192, 8192)]:
    base = synth.depth_image(576, 640, seed=0, nan=False)
    d = torch.from_numpy(np.tile(base, (-(-h // 576), -(-w // 640)))[:h, :w].copy()).to(dev)
    n = h * w
    for name, fn, nbytes in [("erode_depth r=2", lambda: compat.erode_depth(d, 2, ctx=ctx), 8 * n),
                             ("bilateral_filter_depth r=2", lambda: compat.bilateral_filter_depth(d, 2, ctx=ctx), 8 * n),
                             ("depth2xyzmap", lambda: compat.depth2xyzmap(d, K, ctx=ctx), 16 * n)]:
        for _ in range(3):
            fn()
        import time
        reps = 50
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.synchronize()  # the context runs on its own stream: wall clock over back-to-back launches
        ms = 1e3 * (time.perf_counter() - t0) / reps
        print(f"{h}x{w} {name:28s} {1e3*ms:9.1f} us  {nbytes/ms/1e6:8.1f} GB/s  ({nbytes/ms/1e6/8000:.3f} of 8 TB/s)", flush=True)
