"""INTEGRATION.md's Level-2 ctypes stub is executed as written (the fenced block is read from the
file), with the variables a maintainer would have at those call sites, and its results are checked
against the oracle: the document cannot drift from the ABI."""
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_level2_stub_runs_as_documented(oracle):
    from pedp_hip import _lib, synth
    from pedp_hip.compat import PinholeCameraIntrinsic, PointCloud, TriangleMesh

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, re.S)
    stub = next(b for b in blocks if "pedp_ctx_create" in b)
    stub = stub.replace('C.CDLL("libpedp_hip.so")', f'C.CDLL({_lib.LIB_PATH!r})')
    f = synth.Frame("tiny")
    depth_hits = oracle.raycast(f.verts_posed, f.tris, f.rays6)["t_hit"]
    scene = f.scene(depth_hits)
    hm = np.zeros((f.height, f.width)); hm[10:30, 12:36] = 1.0
    c2d = np.eye(4); c2d[:3, 3] = (3.0, -1.0, 0.5)
    env = {
        "mesh": TriangleMesh(f.verts_posed.astype(np.float64), f.tris),
        "ray_tensor": f.rays6,
        "source": PointCloud(scene), "target": PointCloud(f.model_points, normals=f.normals),
        "params": {"distance_threshold": 10.0}, "transformation": f.icp_init(),
        "model_mesh": TriangleMesh(f.model_points, f.tris),
        "color_intrinsics": PinholeCameraIntrinsic(f.width, f.height, intrinsic_matrix=f.K),
        "color_to_depth": c2d, "pose": f.T_gt, "heatmap": hm,
        "depth": synth.depth_image(40, 48, seed=1), "H": 40, "W": 48,
    }
    import torch  # noqa: F401  (one HIP runtime in the process, see INTEGRATION.md "Loading order")
    exec(compile(stub, "INTEGRATION.md:level2", "exec"), env)
    # ray cast of the stub == oracle
    ref = oracle.raycast(f.verts_posed, f.tris, f.rays6)
    assert np.array_equal(env["ids"], ref["primitive_ids"]) and np.array_equal(env["valid_hits"], np.isfinite(ref["t_hit"]))
    # ICP of the stub == oracle (default criteria)
    ricp = oracle.icp(scene, f.model_points, f.normals, 10.0, f.icp_init())
    assert env["rc"] == 0 and env["fit"].value == ricp["fitness"] and np.abs(env["T"] - ricp["T"]).max() < 1e-5
    # fused projection of the stub == oracle on the posed vertices
    v32 = oracle.pose_vertices(np.linalg.inv(c2d) @ f.T_gt, f.model_points)
    rp = oracle.project_heatmap(v32, f.tris, hm, f.K, 0.75)
    assert len(rp["points"]) > 10 and np.abs(env["intersection_points"] - rp["points"]).max() < 1e-9
    # depth filter of the stub == oracle
    assert np.array_equal(env["out"], oracle.erode_depth(env["depth"]), equal_nan=True)
