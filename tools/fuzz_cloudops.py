"""Randomised parity of DBSCAN labels and kNN mean distances against the oracle (clouds with clumps, duplicates, lines, sheets,
outliers; eps from far below to far above the point spacing): python tools/fuzz_cloudops.py [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from pedp_hip import cloud_ops
import pedp_oracle as oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(2024)
bad = 0
for case in range(n_cases):
    parts = []
    for _ in range(rng.integers(1, 6)):
        kind = rng.integers(0, 5)
        n = int(rng.integers(5, int(os.environ.get("FUZZ_NMAX", "1500"))))
        c = rng.uniform(-50, 50, 3)
        if kind == 0:
            parts.append(c + rng.normal(0, rng.uniform(0.2, 6.0), (n, 3)))                       # blob
        elif kind == 1:
            parts.append(c + np.outer(rng.uniform(-30, 30, n), rng.normal(size=3)) + rng.normal(0, 0.05, (n, 3)))   # line
        elif kind == 2:
            parts.append(c + np.column_stack([rng.uniform(-20, 20, n), rng.uniform(-20, 20, n), rng.normal(0, 0.1, n)]))  # sheet
        elif kind == 3:
            parts.append(np.repeat(c[None], n, axis=0) + rng.normal(0, 1e-9, (n, 3)))             # near-duplicates
        else:
            parts.append(rng.uniform(-80, 80, (max(n // 20, 3), 3)))                              # scattered outliers
    pts = np.vstack(parts)
    if rng.random() < 0.3:
        pts = np.vstack([pts, pts[rng.integers(0, len(pts), 40)]])                                # exact duplicates
    pts = pts[rng.permutation(len(pts))]
    eps = float(10 ** rng.uniform(-1.0, 1.3))
    mp = int(rng.integers(1, 15))
    got, ref = cloud_ops.cluster_dbscan(pts, eps, mp), oracle.cluster_dbscan(pts, eps, mp)
    k = int(rng.integers(1, 100))
    ka, kr = cloud_ops.knn_mean_distance(pts, k), oracle.knn_mean_distance(pts, k)
    ok = np.array_equal(got, ref) and np.array_equal(ka, kr)
    bad += not ok
    if not ok:
        print(f"case {case}: N {len(pts)} eps {eps:.3f} min_points {mp} k {k}: labels equal {np.array_equal(got, ref)}, knn equal {np.array_equal(ka, kr)}")
print(f"{n_cases} cases, {bad} mismatches")
