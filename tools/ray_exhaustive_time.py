"""The exhaustive ray sweep (variant 1: every ray x every triangle) on the bench frame: kernel ms by the library's HIP events,
executed / algorithmic TFLOP/s: python tools/ray_exhaustive_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth
ctx = _lib.Context(0)
f = synth.Frame("bench_100k")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
_lib.raycast_configure(ctx, 0, 1)
ms = []
for _ in range(8):
    r = mesh.cast_rays(f.rays6, want_uv=False)
    ms.append(_lib.raycast_last_sweep_ms(ctx))
m = float(np.median(ms[2:]))
tests = f.n_rays * f.n_tris
print(f"{os.path.basename(os.environ.get('PEDP_LIB', 'libpedp_hip.so'))}: exhaustive shared-origin sweep {m:.3f} ms = {f.n_rays / m / 1e3:.1f} Mrays/s; "
      f"hits {int(np.isfinite(r['t_hit']).sum())}")
