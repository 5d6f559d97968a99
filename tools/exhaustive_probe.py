"""The exhaustive ray sweep (every ray x every triangle) of the bench frame: variant 1 (matrix-pipe filter) against
variant 5 (round 3's packed fp32 loop), with the sweeps' chunk counts: python tools/exhaustive_probe.py [chunks ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pedp_hip import _lib, synth

ctx = _lib.default_context()
f = synth.Frame("bench_100k")
m = _lib.Mesh(ctx, f.verts_posed, f.tris)
rays = torch.from_numpy(f.rays6).cuda()
t = torch.empty(f.n_rays, dtype=torch.float32, device="cuda")
ids = torch.empty(f.n_rays, dtype=torch.int32, device="cuda")
ref = None
for variant, chunks in [(5, 0), (1, 0)] + [(1, int(c)) for c in sys.argv[1:]]:
    _lib.raycast_configure(ctx, chunks, variant)
    for rep in range(2):
        m.cast_rays_device(rays.data_ptr(), f.n_rays, t.data_ptr(), ids.data_ptr(), 0)
    ctx.synchronize()
    ts = []
    for rep in range(5):
        m.cast_rays_device(rays.data_ptr(), f.n_rays, t.data_ptr(), ids.data_ptr(), 0)
        ctx.synchronize()
        ts.append(_lib.raycast_last_sweep_ms(ctx))
    key = (t.clone(), ids.clone())
    same = ref is None or (torch.equal(key[0].view(torch.int32), ref[0].view(torch.int32)) and torch.equal(key[1], ref[1]))
    ref = ref or key
    ms = float(np.median(ts))
    tests = f.n_rays * f.n_tris
    print(f"variant {variant} chunks {chunks}: sweep {ms:.3f} ms = {f.n_rays / ms / 1e3:.1f} Mrays/s, "
          f"triangle-stream accounting {(-(-f.n_rays // 64)) * f.n_tris * 36 / ms / 1e6 / 8000:.3f} of 8 TB/s, same bits as variant 5: {same}", flush=True)
_lib.raycast_configure(ctx, 0, 0)
