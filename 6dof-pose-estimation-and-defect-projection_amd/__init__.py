"""pedp_hip -- MI355X-native ICP refinement and ray-mesh defect projection.

The directory name follows the repository convention
(`6dof-pose-estimation-and-defect-projection_amd`); it is not a valid Python identifier,
so import it as `pedp_hip` (the small alias package at the repository root).

Layout
  csrc/            HIP kernels + the C ABI (include/pedp.h) -> libpedp_hip.so
  _lib.py          ctypes binding; raises if the library or the GPU is missing
  geometry.py      PointCloud / TriangleMesh / RegistrationResult / PinholeCameraIntrinsic
                   holders with the attribute names the reference uses (Open3D duck types)
  registration.py  registration_icp, estimation/criteria classes, get_rotation_matrix_from_xyz
  icp_refine.py    host mirror of src/pose_estimation.py's refinement functions
  ray_projection.py host mirror of src/defect_projection.py's projection functions
  compat.py        `from pedp_hip.compat import *` in place of `from src import *` (run.py:3)
  dist.py          one-process-per-GPU sharding over torch.distributed (RCCL)
  synth.py         deterministic synthetic inputs of the benchmark configurations
"""
from . import _lib
from ._lib import PedpError, LIB_PATH

__all__ = ["_lib", "PedpError", "LIB_PATH"]
