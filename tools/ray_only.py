"""Timing of the culled ray stage: python tools/ray_only.py [config]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth
ctx = _lib.Context(0)
f = synth.Frame(sys.argv[1] if len(sys.argv) > 1 else "bench_100k")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
import time
for rep in range(4):
    t0 = time.perf_counter()
    res = mesh.cast_rays(f.rays6, want_uv=False)      # host arrays in and out: PCIe-inclusive
    dt = time.perf_counter() - t0
    print(f"sweep stage {_lib.raycast_last_sweep_ms(ctx):.3f} ms (HIP events); host-memory call {1e3 * dt:.3f} ms "
          f"= {f.n_rays / dt / 1e6:.1f} Mrays/s incl. {f.rays6.nbytes / 1e6:.1f} MB in, {8 * f.n_rays / 1e6:.1f} MB out", flush=True)
