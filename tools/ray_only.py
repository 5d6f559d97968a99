"""Timing of the culled ray stage: python tools/ray_only.py [config]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth
ctx = _lib.Context(0)
f = synth.Frame(sys.argv[1] if len(sys.argv) > 1 else "bench_100k")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
for rep in range(3):
    res = mesh.cast_rays(f.rays6, want_uv=False)
    print(_lib.raycast_last_sweep_ms(ctx), flush=True)
