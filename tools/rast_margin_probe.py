"""Margins and visited cells of the triangle-driven ray stage on the bench frames: python tools/rast_margin_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth
ctx = _lib.Context(0)
for cfg in ("bench_100k", "bench_1m"):
    f = synth.Frame(cfg)
    mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
    tri, ray, (GX, GY, status) = _lib.debug_rast_rects(ctx, mesh, f.rays6)
    live = tri[:, 0] == 1
    full = live & (tri[:, 1] == 1)
    b = live & (tri[:, 1] == 0)
    cells = (tri[:, 9] - tri[:, 8] + 1) * (tri[:, 11] - tri[:, 10] + 1)
    w = tri[b, 3] - tri[b, 2]
    print(f"{cfg}: grid {GX}x{GY} status {status}; live {live.sum()} of {len(tri)}, every-cell {full.sum()}; rectangle width (cells) median {np.median(w):.2f}; "
          f"margin x median {np.median(tri[b,6]):.3f} p90 {np.percentile(tri[b,6],90):.3f} p99 {np.percentile(tri[b,6],99):.3f} max {tri[b,6].max():.2f}; "
          f"cells visited: sum {cells[live].sum():.3g} median {np.median(cells[b]):.0f} p99 {np.percentile(cells[b],99):.0f}; > 16 cells: {(cells[b] > 16).sum()}")
