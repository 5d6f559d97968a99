"""Timing of the dense all-pairs NN sweep (pedp_nn) at bench size: python tools/nn_only.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth
ctx = _lib.Context(0)
f = synth.Frame("bench_100k")
rng = np.random.default_rng(0)
scene = rng.normal(0, 60, (368640, 3))          # timing only: any finite cloud
src = _lib.Cloud(ctx, scene); tgt = _lib.Cloud(ctx, f.model_points, f.normals)
for rep in range(4):
    _lib.nn(ctx, src, tgt, np.eye(4)); ms = _lib.nn_last_sweep_ms(ctx)
print(os.environ.get("PEDP_LIB", "default")[-20:], f"nn sweep {ms:.3f} ms  {8*368640*50000/ms/1e9:.1f} TFLOP/s", flush=True)
