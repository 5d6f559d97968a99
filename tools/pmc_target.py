"""Small workload for rocprofv3 --pmc passes (and a kernel trace): the roofline kernels of bench.py a few times each --
all-pairs NN sweep via pedp_nn, exhaustive ray sweep (variant 1: matrix pipe; variant 5: packed fp32), the triangle-driven ray stage (variant 4 per call, and against a resident ray set),
the cone-culled ray stage (variant 3), two registrations (21 launches of the pass kernel each), the depth pre-filters
on a 4096 x 4096 image."""
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
for variant, reps in ((1, 2), (5, 1), (3, 3), (4, 6)):     # 1: exhaustive on the matrix pipe, 5: round 3's packed fp32 loop
    _lib.raycast_configure(ctx, 0, variant)
    for _ in range(reps):
        mesh.cast_rays(f.rays6, want_uv=False)
_lib.raycast_configure(ctx, 0, 0)
rs = _lib.RaySet(ctx, f.rays6)                              # the resident form of the default ray stage: chains built once
for _ in range(6):
    mesh.cast_rayset(rs, want_uv=False)
for _ in range(2):
    _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), max_iteration=20, relative_fitness=-1, relative_rmse=-1)   # 21 fused passes
big = np.tile(synth.depth_image(512, 512, seed=0, nan=False), (8, 8))
for _ in range(2):
    compat.erode_depth(big, 2, ctx=ctx)
    compat.bilateral_filter_depth(big, 2, ctx=ctx)
    compat.depth2xyzmap(big, f.K, ctx=ctx)
print("done")
