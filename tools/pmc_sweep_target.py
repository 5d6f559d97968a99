"""rocprofv3 --pmc target: the exhaustive ray sweep of the bench frame, variants 1 (matrix pipe) and 5 (packed fp32), three casts each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pedp_hip import _lib, synth
ctx = _lib.Context(0)
f = synth.Frame("bench_100k")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
rays = torch.from_numpy(f.rays6).cuda(); t = torch.empty(f.n_rays, dtype=torch.float32, device="cuda"); ids = torch.empty(f.n_rays, dtype=torch.int32, device="cuda")
for variant in (1, 5):
    _lib.raycast_configure(ctx, 0, variant)
    for _ in range(3):
        mesh.cast_rays_device(rays.data_ptr(), f.n_rays, t.data_ptr(), ids.data_ptr(), 0)
    ctx.synchronize()
print("done")
