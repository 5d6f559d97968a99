"""Bit-stability of the in-launch hand-off of the ICP pass (partial sums -> the workgroup that closes the pass):
    python tools/handoff_probe.py [reps]
Runs the bench-size registration `reps` times alone and as 32-pose batches (two workgroups per CU, uneven load) and
counts results that differ in any bit from the first run and from each other's single-call result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = _lib.Context(0)
f = synth.Frame("bench_100k")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
depth = mesh.cast_rays(f.rays6, want_uv=False)["t_hit"]
src = _lib.Cloud(ctx, f.scene(depth)); tgt = _lib.Cloud(ctx, f.model_points, f.normals)
rng = np.random.default_rng(0)
B = 32
inits = np.repeat(f.icp_init()[None], B, 0).copy()
inits[:, :3, 3] += rng.normal(0, 0.5, (B, 3))
kw = dict(max_iteration=20, relative_fitness=-1, relative_rmse=-1)
single = [_lib.icp(ctx, src, tgt, 10.0, inits[b], **kw) for b in range(B)]
bad_single = 0
for r in range(reps):
    b = r % B
    res = _lib.icp(ctx, src, tgt, 10.0, inits[b], **kw)
    if not (np.array_equal(res["T"], single[b]["T"]) and res["fitness"] == single[b]["fitness"] and res["inlier_rmse"] == single[b]["inlier_rmse"]):
        bad_single += 1
print(f"single registrations repeated {reps} times: {bad_single} differ from their first run", flush=True)
bad_batch = 0; worst = 0.0
for r in range(max(reps // 4, 1)):
    T, fit, rm = _lib.icp_batched(ctx, src, tgt, 10.0, inits, max_iteration=20)[:3]
    for b in range(B):
        if not (np.array_equal(T[b], single[b]["T"]) and fit[b] == single[b]["fitness"] and rm[b] == single[b]["inlier_rmse"]):
            bad_batch += 1
            worst = max(worst, float(np.abs(T[b] - single[b]["T"]).max()))
print(f"{max(reps // 4, 1)} batches of {B}: {bad_batch} poses differ from their single registration (largest |dT| {worst:.3e})", flush=True)
