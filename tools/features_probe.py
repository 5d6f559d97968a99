"""Times of the feature kernels on the bench model (50 k points): python tools/features_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pedp_hip import cloud_ops, synth

f = synth.Frame("bench_100k")
p, n = f.model_points, f.normals
for rep in range(3):
    t0 = time.perf_counter(); F = cloud_ops.compute_fpfh(p, n, 15.0, 100); t1 = time.perf_counter()
    idx = cloud_ops.match_features(F[:6000], F); t2 = time.perf_counter()
    print("fpfh 50k pts %.2f ms, match 6k x 50k %.2f ms (self matches: %d of 6000)" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), int((idx == np.arange(6000)).sum())), flush=True)
