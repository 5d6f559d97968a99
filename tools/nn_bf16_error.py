"""How much of its error bound the dense sweep's bf16 contraction uses (against float64 on the same float32 inputs): python tools/nn_bf16_error.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib
ctx = _lib.Context(0)
rng = np.random.default_rng(3)
for scale_s, scale_t in ((400.0, 120.0), (60.0, 60.0), (5.0, 150.0), (1000.0, 300.0)):
    s = (rng.uniform(-1, 1, (4096, 3)) * scale_s).astype(np.float32)
    t = (rng.uniform(-1, 1, (4096, 3)) * scale_t).astype(np.float32)
    t2 = (t.astype(np.float64) ** 2).sum(1).astype(np.float32)
    src4 = np.column_stack([-2.0 * s, np.ones(len(s), np.float32)]).astype(np.float32)
    tgt4 = np.column_stack([t, t2]).astype(np.float32)
    g = _lib.debug_nn_bf16(ctx, src4, tgt4).astype(np.float64)
    ref = t2.astype(np.float64)[None, :] + src4[:, :3].astype(np.float64) @ t.astype(np.float64).T
    M = 2.0 * np.abs(s).sum(1) * np.abs(t).sum(1).max() + t2.max()
    err = np.abs(g - ref)
    ulp = M[:, None] * 2.0 ** -23
    print(f"|s| <= {scale_s:6.0f}, |t| <= {scale_t:5.0f}: {g.size:.2e} pairs, largest error {err.max():.3e} = {float((err / ulp).max()):.3f} M 2^-23 (bound 34), mean {float((err / ulp).mean()):.4f}")
