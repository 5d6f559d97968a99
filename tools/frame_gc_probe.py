"""Frame times of the chain over a few hundred frames, with and without FrameChain's gc.freeze(): python tools/frame_gc_probe.py [frames]
Prints mean / median / p99 / max frame time and the collector's runs (generation, duration) inside the loop."""
import gc, os, sys, logging, queue, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import _lib, synth, viewer_wire
from pedp_hip import frame_chain as fc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
f = synth.Frame("bench_100k")
m = _lib.Mesh(_lib.default_context(), f.verts_posed, f.tris)
t_hit = m.cast_rays(f.rays6, want_uv=False)["t_hit"]
viewer_wire.attach_queues(viewer_wire.LatestQueue())
root = logging.getLogger(); sink = logging.StreamHandler(open(os.devnull, "w")); sink.setFormatter(logging.Formatter("[%(funcName)s()] %(message)s")); root.addHandler(sink); root.setLevel(logging.INFO)
runs = []
gc.callbacks.append(lambda phase, info: runs.append((phase, info["generation"], time.perf_counter())))
for freeze in (False, True):
    gc.unfreeze()
    orig = fc.FrameChain.__init__
    def init(self, *a, **k):
        k["freeze_gc"] = freeze
        return orig(self, *a, **k)
    fc.FrameChain.__init__ = init
    chain, depth_m, heat, init_pose = fc.bench_frame_setup(f, t_hit)
    fc.FrameChain.__init__ = orig
    for k in range(5):
        chain.process(depth_m, init_pose(), heat, seed=0)
    del runs[:]
    ts = []
    for k in range(N):
        t0 = time.perf_counter()
        chain.process(depth_m, init_pose(), heat, seed=k)
        ts.append(1e3 * (time.perf_counter() - t0))
    ts = np.array(ts)
    st = [r for r in runs if r[0] == "start"]; en = [r for r in runs if r[0] == "stop"]
    per_gen = {g: [round(1e3 * (e[2] - s[2]), 2) for s, e in zip(st, en) if s[1] == g] for g in (0, 1, 2)}
    print(f"gc.freeze() {'on ' if freeze else 'off'}: {N} frames, mean {ts.mean():.3f} median {np.median(ts):.3f} p99 {np.percentile(ts, 99):.3f} max {ts.max():.3f} ms; "
          f"collections gen0 {len(per_gen[0])} gen1 {len(per_gen[1])} gen2 {len(per_gen[2])}; gen2 durations ms {per_gen[2]}; gen1 max {max(per_gen[1] or [0])}")
