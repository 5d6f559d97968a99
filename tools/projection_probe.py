"""The frame chain's projection stage piece by piece (run.py:109-118): posed mesh for the viewer, pose of the resident
model, heat-map projection with the heat map as float64 / float32 host array and as a CUDA tensor:
python tools/projection_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pedp_hip import _lib, compat, synth
from pedp_hip.ray_projection import FrameProjector, _jet_lut

f = synth.Frame("bench_100k")
ctx = _lib.default_context()
intr = compat.PinholeCameraIntrinsic(f.width, f.height, intrinsic_matrix=f.K)
c2d = np.eye(4); c2d[:3, 3] = (2.0, -1.0, 0.5)
model = compat.TriangleMesh(f.model_points, f.tris)
proj = FrameProjector(model, intr, c2d)
heat = np.zeros((f.height, f.width)); heat[200:380, 220:420] = np.linspace(0.76, 1.0, 200)[None, :]
pose = f.T_gt

def med(fn, n=30):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); ctx.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    return float(np.median(ts[5:])), r

print("posed_mesh (device vertices + download)  %.3f ms" % med(lambda: proj.posed_mesh(pose, model))[0])
print("transform_object (host)                  %.3f ms" % med(lambda: compat.transform_object(model, pose))[0])
print("set_pose                                 %.3f ms" % med(lambda: proj.mesh.set_pose(pose))[0])
lut = _jet_lut()
for name, h in (("float64 host", heat), ("float32 host", heat.astype(np.float32)), ("float64 cuda", torch.from_numpy(heat).cuda()),
                ("float32 cuda", torch.from_numpy(heat.astype(np.float32)).cuda())):
    t, r = med(lambda: proj.mesh.project_heatmap(h, f.K, 0.75, jet_lut=lut, post=c2d))
    print(f"project_heatmap {name:14s}           {t:.3f} ms  ({len(r['points'])} hits)")
    t, r = med(lambda: proj.project(pose, h, 0.75, into=c2d))
    print(f"FrameProjector.project {name:14s}    {t:.3f} ms")
t, _ = med(lambda: (proj.posed_mesh(pose, model), proj.project(pose, heat, 0.75, into=c2d)))
print("whole stage (float64 host)               %.3f ms" % t)
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    proj.posed_mesh(pose, model); proj.project(pose, heat, 0.75, into=c2d)
pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue())
