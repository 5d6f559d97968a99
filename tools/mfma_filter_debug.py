"""Debug of the matrix-pipe filter: its scores on the tiny frame against a float64 statement of the three dot products."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from pedp_hip import _lib, synth
import pedp_oracle as oracle

ctx = _lib.default_context()
f = synth.Frame("tiny")
mesh = _lib.Mesh(ctx, f.verts_posed, f.tris)
rays = f.rays6
score, slack = _lib.debug_mfma_scores(ctx, mesh, rays)
v = f.verts_posed.astype(np.float64)
v0, v1, v2 = v[f.tris[:, 0]], v[f.tris[:, 1]], v[f.tris[:, 2]]
e1, e2 = v1 - v0, v2 - v0
m = np.cross(e2, e1)
o = rays[0, :3].astype(np.float64)
s = o - v0
a, b = np.cross(e2, s), np.cross(s, e1)
tn = -(s * m).sum(1)
sg = np.where(tn < 0, -1.0, 1.0)[:, None]
A, B, Cc = sg * a, sg * b, sg * (m - a - b)
d = rays[:, 3:].astype(np.float64)
ua, ub, uc = d @ A.T, d @ B.T, d @ Cc.T
exp = np.minimum(np.minimum(ua, ub), uc)
raw = score.astype(np.float64) - slack
print("shape", score.shape, "finite", np.isfinite(score).mean(), "pass share", (score >= 0).mean(), "expected pass share", (exp >= 0).mean())
err = np.abs(raw - exp) / (np.abs(d).sum(1)[:, None] * np.maximum(np.maximum(np.abs(A).max(1), np.abs(B).max(1)), np.abs(Cc).max(1))[None, :])
print("max relative error of score - slack against the float64 min edge:", err.max(), "median", np.median(err))
i, j = np.unravel_index(np.argmax(err), err.shape)
print("worst pair", i, j, "raw", raw[i, j], "expected", exp[i, j], "ua ub uc", ua[i, j], ub[i, j], uc[i, j], "slack", slack[i, j])
print("corr with ua, ub, uc:", [float(np.corrcoef(raw.ravel()[:50000], x.ravel()[:50000])[0, 1]) for x in (ua, ub, uc)])
pairs = oracle.accepted_pairs(f.verts_posed, f.tris, rays)
print("accepted pairs", len(pairs), "rejected by the filter", int((score[pairs[:, 0], pairs[:, 1]] < 0).sum()))
print("slack/|d|1 per triangle (first 5):", (slack / np.abs(d).sum(1)[:, None])[0, :5], "expected w*max|x| ~", 7.9e-6 * np.abs(A).max(1)[:5])
