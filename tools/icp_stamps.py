"""In-kernel phase timing of the fused ICP pass (diagnostic build with s_memtime stamps):
    python tools/icp_stamps.py [max_iteration]
Builds libpedp_hip_stamps.so with -DPEDP_ICP_STAMPS=1 (the product library carries no stamps), runs one
registration at bench size and prints, for the LAST pass, the cycles between phase boundaries
(s_memtime ticks at 100 MHz -> microseconds)."""
import ctypes as C
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "6dof-pose-estimation-and-defect-projection_amd")
spec = importlib.util.spec_from_file_location("pedp_build", os.path.join(PKG, "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
out = os.path.join(PKG, "libpedp_hip_stamps.so")
b.build(force=not os.path.exists(out), verbose=False, extra_flags=["-DPEDP_ICP_STAMPS=1"], out=out)
os.environ["PEDP_LIB"] = out
import numpy as np
from pedp_hip import _lib, synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = _lib.Context(0)
f = synth.Frame("bench_100k")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
depth = mesh.cast_rays(f.rays6, want_uv=False)["t_hit"]
src = _lib.Cloud(ctx, f.scene(depth)); tgt = _lib.Cloud(ctx, f.model_points, f.normals)
for _ in range(3):
    _lib.icp(ctx, src, tgt, 10.0, f.icp_init(), max_iteration=iters, relative_fitness=-1, relative_rmse=-1)
lib = C.CDLL(out)
buf = np.zeros((3, 4096, 8), np.int64)
assert lib.pedp_debug_icp_stamps(C.c_void_p(buf.ctypes.data)) == 0
cal = buf[2][1]
mhz = (cal[2] - cal[0]) / max(cal[3] - cal[1], 1) * 100.0   # s_memtime ticks per s_memrealtime tick (100 MHz)
print(f"s_memtime runs at {mhz:.0f} MHz (finish kernel: {(cal[3]-cal[1])/100:.2f} us)")
tick = 1.0 / mhz
blk = buf[1][buf[1][:, 6] > 0]
blk = blk[blk[:, 0] > blk[:, 0].max() - 400 * mhz]   # the last pass only
if len(blk):
    d = np.diff(blk[:, :7], axis=1) * tick
    names = ["transform+pack", "cull", "sweep", "select", "fallback", "accumulate"]
    print(f"block: {len(blk)} workgroups; span {(blk[:,6].max() - blk[:,0].min()) * tick:.2f} us")
    for k, n in enumerate(names):
        print(f"   {n:12s} median {np.median(d[:,k]):7.2f}  p90 {np.percentile(d[:,k],90):7.2f}  max {d[:,k].max():7.2f} us")
    info = blk[:, 7]
    tiles, nsurv, nfb, ncand = info >> 32, (info >> 24) & 0xFF, (info >> 16) & 0xFF, info & 0xFFFF
    tot = (blk[:, 6] - blk[:, 0]) * tick
    worst = np.argsort(-tot)[:5]
    for w in worst:
        print(f"   slowest: total {tot[w]:.2f} us tiles {tiles[w]} words {nsurv[w]} fallback slots {nfb[w]} candidates {ncand[w]}  phases {np.round(d[w],2)}")
    print(f"   tiles/block median {np.median(tiles)} max {tiles.max()}; fallback slots/block max {nfb.max()} sum {nfb.sum()}")
sv = buf[2][2]
print("finish, last solving pass: LDLT %.2f  sincos+T %.2f  T update, motion, stores %.2f us" % tuple(np.diff(sv[:4]) * tick))
fin = buf[2][0]
print("finish: reduce %.2f solve %.2f tail %.2f us" % tuple(np.diff(fin[:4]) * tick))
