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
blk = buf[1][buf[1][:, 5] > 0]
blk = blk[blk[:, 0] > blk[:, 0].max() - 100 * mhz]   # the last pass only (stamps of earlier passes are overwritten per workgroup)
if len(blk):
    d = np.diff(blk[:, :6], axis=1) * tick
    names = ["transform", "cull+sweep", "select", "exact search", "sums+store"]
    print(f"pass kernel, wave 0 of each workgroup: {len(blk)} workgroups; span {(blk[:,5].max() - blk[:,0].min()) * tick:.2f} us")
    for k, n in enumerate(names):
        print(f"   {n:12s} median {np.median(d[:,k]):7.2f}  p90 {np.percentile(d[:,k],90):7.2f}  max {d[:,k].max():7.2f} us")
    info = blk[:, 7]
    tiles, nfb, nsl = info >> 32, (info >> 16) & 0xFFFF, info & 0xFFFF
    tot = (blk[:, 5] - blk[:, 0]) * tick
    worst = np.argsort(-tot)[:5]
    for w in worst:
        print(f"   slowest: total {tot[w]:.2f} us  wave-0 tiles {tiles[w]} exact searches {nfb[w]} slots {nsl[w]}  phases {np.round(d[w],2)}")
    print(f"   wave-0 tiles median {np.median(tiles)} max {tiles.max()}")
    last = blk[blk[:, 6] > 0]
    if len(last):
        fin0 = buf[2][0]
        print(f"   last arriver: ticket taken {(last[0,6] - blk[:,0].min()) * tick:.2f} us after the first workgroup started; finish body {(fin0[3]-fin0[0])*tick:.2f} us")
sv = buf[2][2]
print("finish, last solving pass: LDLT %.2f  sincos+T %.2f  T update, motion, stores %.2f us" % tuple(np.diff(sv[:4]) * tick))
fin = buf[2][0]
f5 = buf[2][3]
if f5[6] > 0:
    print("finish of pass 5 (us): state read + partial loads %.2f | barrier %.2f | 32-range sum + barrier %.2f | fitness, trace, criteria %.2f | solve + sincos %.2f | pose, bound, state %.2f" % tuple(np.diff(f5[:7]) * tick))
# device-wide timeline (s_memrealtime, 10-ns ticks) of every pass: first entry -> last chunk done -> pass closed -> next pass's first entry
rt = np.zeros((32, 512, 8), np.int64)
if hasattr(lib, "pedp_debug_icp_rt") and lib.pedp_debug_icp_rt(C.c_void_p(rt.ctypes.data)) == 0:
    prev_close = None
    for p in range(min(iters + 1, 32)):
        r = rt[p]
        on = r[:, 0] > 0
        if not on.any():
            continue
        e0 = r[on, 0].min()
        last = r[on, 4].max()
        us = lambda t: (t - e0) / 100.0
        line = (f"pass {p:2d}: {on.sum():3d} workgroups; entry spread {us(r[on,0].max()):5.2f}; state read (median) {np.median(us(r[on,1])):5.2f}; "
                f"chunks done median {np.median(us(r[on,2])):6.2f} max {us(r[on,2].max()):6.2f}; ticket back max {us(r[on,3].max()):6.2f}; closed {us(last):6.2f}")
        if prev_close is not None:
            line += f"; boundary (closed -> next entry) {(e0 - prev_close) / 100.0:5.2f}"
        prev_close = last
        print(line)
# per-wave view of the last pass
wvb = np.zeros((512, 8, 16), np.int64)
if hasattr(lib, "pedp_debug_icp_wave") and lib.pedp_debug_icp_wave(C.c_void_p(wvb.ctypes.data)) == 0:
    w = wvb.reshape(-1, 16)
    last_pass = int(w[:, 9].max())
    w = w[(w[:, 4] > 0) & (w[:, 2] > 0) & (w[:, 9] == last_pass)]   # s_memtime is per XCD: select by pass, not by time
    ph = np.diff(w[:, :5], axis=1) * tick
    words, batches, wide, slots, tiles = w[:, 5] >> 32, (w[:, 5] >> 16) & 0xFFFF, ((w[:, 5] >> 8) & 0xFF) - 1, w[:, 5] & 0xFF, w[:, 6]
    print(f"waves with slots in the last pass: {len(w)}; wide {int(wide.sum())}")
    for k, n in enumerate(["slots ready", "cull+sweep", "select", "sums"]):
        print(f"   {n:12s} median {np.median(ph[:,k]):6.2f} p90 {np.percentile(ph[:,k],90):6.2f} max {ph[:,k].max():6.2f}")
    print(f"   inside cull+sweep: cover {np.median((w[:,7]-w[:,1])*tick):5.2f} (p90 {np.percentile((w[:,7]-w[:,1])*tick,90):5.2f}) | word + tile tests, list {np.median((w[:,8]-w[:,7])*tick):5.2f} (p90 {np.percentile((w[:,8]-w[:,7])*tick,90):5.2f}) | sweep {np.median((w[:,2]-w[:,8])*tick):5.2f} (p90 {np.percentile((w[:,2]-w[:,8])*tick,90):5.2f})")
    print(f"   words median {np.median(words)} p90 {np.percentile(words,90)} max {words.max()}; batches median {np.median(batches)} max {batches.max()}; tiles median {np.median(tiles)} p90 {np.percentile(tiles,90)} max {tiles.max()}")
    cs = ph[:, 1]
    for b in sorted(set(batches.tolist())):
        m = batches == b
        print(f"   batches {b}: {m.sum():4d} waves, cull+sweep median {np.median(cs[m]):6.2f} max {cs[m].max():6.2f}, tiles median {np.median(tiles[m])}, wide {int(wide[m].sum())}")
    worst = np.argsort(-cs)[:8]
    for i in worst:
        print(f"   slowest cull+sweep {cs[i]:6.2f}: words {words[i]} batches {batches[i]} tiles {tiles[i]} wide {wide[i]} slots {slots[i]}")
    # per workgroup: how much of its time is the spread of the sweep between its waves
    g = wvb.copy()
    ok = (g[:, :, 4] > 0) & (g[:, :, 2] > 0) & (g[:, :, 9] == last_pass)
    rows = []
    for b in range(g.shape[0]):
        m = ok[b]
        if m.sum() < 2:
            continue
        t = g[b][m]
        total = (t[:, 4] - t[:, 0]) * tick
        sweep = (t[:, 2] - t[:, 8]) * tick
        lists = (t[:, 8] - t[:, 1]) * tick
        rows.append((total.max(), total.mean(), sweep.max(), sweep.mean(), (total - sweep).max() + sweep.mean(), lists.max(), lists.mean(), m.sum()))
    if rows:
        r = np.array(rows)
        print(f"workgroups with per-wave stamps: {len(r)}; waves each median {np.median(r[:,7]):.0f}")
        print(f"   slowest wave of a workgroup: median {np.median(r[:,0]):.2f} max {r[:,0].max():.2f} | mean wave: median {np.median(r[:,1]):.2f} max {r[:,1].max():.2f}")
        print(f"   sweep: slowest wave median {np.median(r[:,2]):.2f} max {r[:,2].max():.2f} | mean median {np.median(r[:,3]):.2f} max {r[:,3].max():.2f}")
        print(f"   if the sweep were shared evenly inside a workgroup: slowest wave median {np.median(r[:,4]):.2f} max {r[:,4].max():.2f}")
        print(f"   cover + lists: slowest wave median {np.median(r[:,5]):.2f} max {r[:,5].max():.2f} | mean median {np.median(r[:,6]):.2f}")

    # the slowest workgroups of the last pass on the device-wide clock: where their slowest wave spent its time
    t0 = g[:, :, 10][ok].min()
    done = np.where(ok, g[:, :, 11], 0).max(axis=1)
    order = np.argsort(-done)[:10]
    print("slowest workgroups (us after the first wave's start on the 100-MHz clock; phases of the wave that finished last):")
    for b in order:
        if done[b] == 0:
            continue
        m = ok[b]
        t = g[b][m]
        k = int(np.argmax(t[:, 11]))
        q = t[k]
        info = q[5]
        print(f"   wg {b:3d}: done {(done[b]-t0)/100:6.2f}, started {(t[:,10].min()-t0)/100:5.2f}; its last wave: slots ready {(q[1]-q[0])*tick:5.2f} cover {(q[7]-q[1])*tick:5.2f} "
              f"lists {(q[8]-q[7])*tick:5.2f} sweep {(q[2]-q[8])*tick:5.2f} select {(q[3]-q[2])*tick:5.2f} sums {(q[4]-q[3])*tick:5.2f} | words {info>>32} batches {(info>>16)&0xFFFF} "
              f"tiles {q[6]} slots {info&0xFF}; waves' ends spread {(t[:,11].max()-t[:,11].min())/100:5.2f}")
    ends = (done[done > 0] - t0) / 100
    print(f"   workgroup end times: median {np.median(ends):.2f} p90 {np.percentile(ends,90):.2f} p99 {np.percentile(ends,99):.2f} max {ends.max():.2f}")
    tl = (g[:, :, 6][ok])
    print(f"   tiles per wave: median {np.median(tl)} p90 {np.percentile(tl,90)} p99 {np.percentile(tl,99)} max {tl.max()}")

    # which workgroups shared a CU (HW_ID bits 15:8 = CU / SH / SE, XCC_ID): times of lone and of co-resident workgroups
    hw = g[:, 0, 12]
    cu_key = ((hw >> 32) & 0xF) * 256 + ((hw & 0xFFFFFFFF) >> 8 & 0xFF)
    live = ok.any(axis=1)
    import collections
    groups = collections.defaultdict(list)
    for b in np.nonzero(live)[0]:
        groups[int(cu_key[b])].append(int(b))
    dur = np.where(ok, g[:, :, 11] - g[:, :, 10], 0).max(axis=1) / 100.0
    lone = [dur[v[0]] for v in groups.values() if len(v) == 1]
    shared = [dur[b] for v in groups.values() if len(v) > 1 for b in v]
    print(f"CUs in use {len(groups)}; workgroups alone on a CU {len(lone)}: duration median {np.median(lone):.2f} p90 {np.percentile(lone,90):.2f} max {np.max(lone):.2f} us")
    if shared:
        print(f"   workgroups sharing a CU {len(shared)} (on {sum(1 for v in groups.values() if len(v) > 1)} CUs, up to {max(len(v) for v in groups.values())} each): duration median {np.median(shared):.2f} p90 {np.percentile(shared,90):.2f} max {np.max(shared):.2f} us")
    pairs = sorted(tuple(v) for v in groups.values() if len(v) > 1)
    print("   co-resident workgroups:", pairs[:40])
    xcc = ((hw >> 32) & 0xF)
    print("   XCC of workgroups 0..15:", [int(x) for x in xcc[:16]], " CU field of 0..7:", [int((h & 0xFFFFFFFF) >> 8 & 0xFF) for h in hw[:8]], " of 256..263:", [int((h & 0xFFFFFFFF) >> 8 & 0xFF) for h in hw[256:264]])
    # work of lone and of sharing workgroups: slots and swept tiles per workgroup
    slots_wg = np.where(ok, g[:, :, 5] & 0xFF, 0).sum(axis=1)
    tiles_wg = np.where(ok, g[:, :, 6], 0).sum(axis=1)
    lone_b = [v[0] for v in groups.values() if len(v) == 1]
    shared_b = [b for v in groups.values() if len(v) > 1 for b in v]
    for name, bs in (("alone", lone_b), ("sharing", shared_b)):
        if bs:
            bs = np.array(bs)
            print(f"   {name:8s}: slots per workgroup median {np.median(slots_wg[bs]):.0f} (min {slots_wg[bs].min()}, max {slots_wg[bs].max()}); tiles swept median {np.median(tiles_wg[bs]):.0f} (max {tiles_wg[bs].max()}); duration median {np.median(dur[bs]):.2f}")
    order_d = np.argsort(dur[np.array(lone_b)])
    lb = np.array(lone_b)[order_d]
    print("   lone workgroups by duration, deciles of (duration, slots, tiles):", [(round(float(dur[b]), 1), int(slots_wg[b]), int(tiles_wg[b])) for b in lb[:: max(len(lb) // 10, 1)]])
