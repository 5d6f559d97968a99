"""Small workload for rocprofv3 --pmc passes: the two roofline kernels of bench.py, twice each
(all-pairs NN sweep via pedp_nn, exhaustive ray sweep variant 1) plus the culled ray stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth
ctx = _lib.Context(0)
f = synth.Frame("bench_100k")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
depth = mesh.cast_rays(f.rays6, want_uv=False)["t_hit"]
src = _lib.Cloud(ctx, f.scene(depth)); tgt = _lib.Cloud(ctx, f.model_points, f.normals)
for _ in range(2):
    _lib.nn(ctx, src, tgt, f.icp_init())
_lib.raycast_configure(ctx, 0, 1)
for _ in range(2):
    mesh.cast_rays(f.rays6, want_uv=False)
_lib.raycast_configure(ctx, 0, 0)
mesh.cast_rays(f.rays6, want_uv=False)
print("done")
