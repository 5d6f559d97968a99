"""Which sweep variant answers the bench frames, and how long the ray stage takes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pedp_hip import _lib, synth

ctx = _lib.default_context()
for config in sys.argv[1:] or ["bench_100k", "bench_1m"]:
    f = synth.Frame(config)
    m = _lib.Mesh(ctx, f.verts_posed, f.tris)
    rays = torch.from_numpy(f.rays6).cuda()
    t = torch.empty(f.n_rays, dtype=torch.float32, device="cuda")
    ids = torch.empty(f.n_rays, dtype=torch.int32, device="cuda")
    for variant in (4, 3):
        _lib.raycast_configure(ctx, 0, variant)
        for rep in range(3):
            m.cast_rays_device(rays.data_ptr(), f.n_rays, t.data_ptr(), ids.data_ptr(), 0)
        ctx.synchronize()
        t0 = time.perf_counter()
        for rep in range(20):
            m.cast_rays_device(rays.data_ptr(), f.n_rays, t.data_ptr(), ids.data_ptr(), 0)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(config, "variant", variant, "last", _lib.raycast_last_variant(ctx), f"{dt * 1e3:.3f} ms per cast, sweep stage",
              f"{_lib.raycast_last_sweep_ms(ctx):.3f} ms, hits {int(torch.isfinite(t).sum())}", flush=True)
    _lib.raycast_configure(ctx, 0, 0)
    rs = _lib.RaySet(ctx, device_ptr=rays.data_ptr(), n=f.n_rays)          # the rays resident: chains built once
    for rep in range(3):
        m.cast_rayset_device(rs, t.data_ptr(), ids.data_ptr())
    ctx.synchronize()
    t0 = time.perf_counter()
    for rep in range(20):
        m.cast_rayset_device(rs, t.data_ptr(), ids.data_ptr())
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(config, "resident ray set, last", rs.last_variant(), f"{dt * 1e3:.3f} ms per cast, sweep stage",
          f"{_lib.raycast_last_sweep_ms(ctx):.3f} ms, hits {int(torch.isfinite(t).sum())}", flush=True)
