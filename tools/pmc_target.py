"""Small workload for rocprofv3 --pmc passes: the roofline kernels of bench.py twice each
(all-pairs NN sweep via pedp_nn, exhaustive ray sweep variant 1), the culled ray stage, one
registration (culled NN kernels) and the depth pre-filters on a 4096 x 4096 image."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, compat, synth
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
for _ in range(2):
    mesh.cast_rays(f.rays6, want_uv=False)
_lib.icp(ctx, src, tgt, 10.0, f.icp_init(), max_iteration=20, relative_fitness=-1, relative_rmse=-1)   # 21 fused passes
big = np.tile(synth.depth_image(512, 512, seed=0, nan=False), (8, 8))
for _ in range(2):
    compat.erode_depth(big, 2, ctx=ctx)
    compat.depth2xyzmap(big, f.K, ctx=ctx)
print("done")
