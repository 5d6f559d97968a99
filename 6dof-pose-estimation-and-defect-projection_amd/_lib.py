"""ctypes binding of libpedp_hip.so (C ABI: include/pedp.h).

The library is the product: if it is missing, cannot be loaded, or no GPU answers,
every entry point raises -- there is no CPU fallback in this package.
"""
import ctypes as C
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PEDP_LIB", os.path.join(_HERE, "libpedp_hip.so"))

HOST, DEVICE = 0, 1
POINT_TO_PLANE, POINT_TO_POINT = 0, 1


class PreprocessParams(C.Structure):
    """pedp_preprocess_params (include/pedp.h)."""
    _fields_ = [("voxel_size", C.c_double), ("plane_distance", C.c_double), ("plane_iterations", C.c_int32),
                ("first_frame", C.c_int32), ("seed", C.c_uint64), ("normal_radius", C.c_double), ("normal_max_nn", C.c_int32),
                ("cluster_min_points", C.c_int32), ("cluster_eps", C.c_double), ("outlier_neighbors", C.c_int32),
                ("flags", C.c_int32), ("outlier_std_ratio", C.c_double), ("average_normal_voxel", C.c_double)]

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)


class IcpParams(C.Structure):
    _fields_ = [
        ("max_correspondence_distance", C.c_double),
        ("estimator", C.c_int),
        ("max_iteration", C.c_int),
        ("relative_fitness", C.c_double),
        ("relative_rmse", C.c_double),
        ("allreduce", ALLREDUCE_FN),
        ("allreduce_user", C.c_void_p),
        ("n_source_global", C.c_int64),
        ("use_comm", C.c_int),
    ]


# name -> (restype, argtypes); every symbol include/pedp.h declares
_P = C.POINTER
PROTOTYPES = {
    "pedp_version": (C.c_int, []),
    "pedp_last_error": (C.c_char_p, []),
    "pedp_device_count": (C.c_int, [_P(C.c_int)]),
    "pedp_ctx_create": (C.c_int, [C.c_int, C.c_void_p, _P(C.c_void_p)]),
    "pedp_ctx_destroy": (None, [C.c_void_p]),
    "pedp_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "pedp_mesh_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, _P(C.c_void_p)]),
    "pedp_mesh_destroy": (None, [C.c_void_p]),
    "pedp_mesh_size": (C.c_int, [C.c_void_p, _P(C.c_int64), _P(C.c_int64)]),
    "pedp_raycast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pedp_mesh_create_posable": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, _P(C.c_void_p)]),
    "pedp_mesh_set_pose": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pedp_project_heatmap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int,
                                       C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _P(C.c_int64),
                                       _P(C.c_int64)]),
    "pedp_depth_to_scene": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      _P(C.c_int64)]),
    "pedp_host_alloc": (C.c_int, [C.c_size_t, _P(C.c_void_p)]),
    "pedp_host_free": (None, [C.c_void_p]),
    "pedp_rayset_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, _P(C.c_void_p)]),
    "pedp_rayset_destroy": (None, [C.c_void_p]),
    "pedp_raycast_rayset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pedp_rayset_last_variant": (C.c_int, [C.c_void_p, _P(C.c_int), _P(C.c_int)]),
    "pedp_debug_mfma_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "pedp_mesh_posed_vertices": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pedp_project_heatmap_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int,
                                          C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _P(C.c_int64),
                                          _P(C.c_int64), C.c_void_p]),
    "pedp_erode_depth": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                   C.c_int, C.c_void_p]),
    "pedp_bilateral_filter_depth": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                              C.c_float, C.c_int, C.c_void_p]),
    "pedp_depth2xyzmap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "pedp_depth2xyzmap_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float,
                                          C.c_int, C.c_void_p]),
    "pedp_voxel_down_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p,
                                         C.c_int64, _P(C.c_int64)]),
    "pedp_voxel_down_sample_device_in": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_int64,
                                                   _P(C.c_int64)]),
    "pedp_preprocess_source": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                         _P(C.c_int64), _P(C.c_int64), _P(C.c_int)]),
    "pedp_preprocess_source_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                            _P(C.c_int64), _P(C.c_int64), _P(C.c_int), C.c_void_p]),
    "pedp_transform_points": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "pedp_cluster_dbscan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_void_p]),
    "pedp_knn_mean_distance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "pedp_estimate_normals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_void_p, C.c_void_p]),
    "pedp_ransac_hypotheses": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_int,
                                         C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "pedp_fpfh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_void_p]),
    "pedp_feature_match": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "pedp_segment_plane": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_uint64, C.c_void_p,
                                     C.c_void_p, _P(C.c_int64)]),
    "pedp_raycast_configure": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "pedp_raycast_last_sweep_ms": (C.c_int, [C.c_void_p, _P(C.c_float)]),
    "pedp_raycast_last_variant": (C.c_int, [C.c_void_p, _P(C.c_int), _P(C.c_int)]),
    "pedp_debug_rast_rects": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pedp_cloud_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, _P(C.c_void_p)]),
    "pedp_cloud_create_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, _P(C.c_void_p)]),
    "pedp_cloud_destroy": (None, [C.c_void_p]),
    "pedp_cloud_size": (C.c_int, [C.c_void_p, _P(C.c_int64), _P(C.c_int)]),
    "pedp_icp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _P(IcpParams), C.c_void_p, C.c_void_p,
                           _P(C.c_double), _P(C.c_double), _P(C.c_int32), C.c_void_p, C.c_void_p]),
    "pedp_icp_batched": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _P(IcpParams), C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "pedp_icp_batched_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _P(IcpParams), C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pedp_nn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pedp_nn_last_sweep_ms": (C.c_int, [C.c_void_p, _P(C.c_float)]),
    "pedp_icp_last_stats": (C.c_int, [C.c_void_p, _P(C.c_int64), _P(C.c_int64), _P(C.c_int64)]),
    "pedp_icp_last_planned_passes": (C.c_int, [C.c_void_p, _P(C.c_int64)]),
    "pedp_debug_nn_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "pedp_icp_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _P(IcpParams), C.c_void_p, C.c_int]),
    "pedp_icp_end": (C.c_int, [C.c_void_p, C.c_void_p, _P(C.c_double), _P(C.c_double), _P(C.c_int32), C.c_void_p, C.c_void_p]),
    "pedp_icp_configure": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "pedp_comm_unique_id": (C.c_int, [C.c_void_p]),
    "pedp_comm_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "pedp_comm_destroy": (C.c_int, [C.c_void_p]),
    "pedp_comm_size": (C.c_int, [C.c_void_p, _P(C.c_int), _P(C.c_int)]),
    "pedp_comm_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "pedp_comm_allreduce_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pedp_cluster_poses": (C.c_int, [C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                     C.c_void_p, _P(C.c_int)]),
}

_lib = None
_lock = threading.Lock()


class PedpError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and bind every prototype.  Raises if it is absent."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise PedpError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` or "
                f"`python 6dof-pose-estimation-and-defect-projection_amd/build.py` "
                "(this package has no CPU fallback)")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as
        # the system one this library links to).  If the system copy is loaded first, a later
        # `import torch` finds "No HIP GPUs"; loading torch first makes both share torch's copy.
        # torch is optional for the library itself -- skip quietly when it is not installed.
        if "torch" not in sys.modules and os.environ.get("PEDP_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI and the header disagree
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(status, what=""):
    if status != 0:
        msg = load().pedp_last_error()
        raise PedpError(f"{what} failed (status {status}): {msg.decode() if msg else ''}")


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class _Lease:
    """A page-locked block on loan from the pool: numpy arrays over it keep it alive through `.base`; when the last of them
    is gone the block goes back to the pool."""

    def __init__(self, ptr, cap):
        self.ptr, self.cap = ptr, cap
        self.__array_interface__ = {"shape": (cap,), "typestr": "|u1", "data": (ptr, False), "version": 3}

    def __del__(self):
        try:
            _host_pool_give(self.ptr, self.cap)
        except Exception:
            pass


_HOST_POOL = {}        # capacity (a power of two) -> page-locked blocks at rest
_HOST_POOL_KEEP = 6    # blocks kept per capacity; the rest are freed


def _host_pool_give(ptr, cap):
    lst = _HOST_POOL.setdefault(cap, [])
    if len(lst) < _HOST_POOL_KEEP:
        lst.append(ptr)
    else:
        load().pedp_host_free(C.c_void_p(ptr))


def host_array(shape, dtype):
    """A result array in page-locked memory from a pool (pedp_host_alloc): downloads reach it without a staging copy, and
    its pages do not fault in anew every frame the way a fresh np.empty's do.  An ordinary (writeable) numpy array to its
    holder; the block returns to the pool when the array and every view of it are gone."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    if n < (64 << 10):                       # small results: the allocator's own free lists serve these without faults
        return np.empty(shape, dtype)
    cap = 1 << (n - 1).bit_length()
    lst = _HOST_POOL.get(cap)
    if lst:
        ptr = lst.pop()
    else:
        p = C.c_void_p()
        check(load().pedp_host_alloc(cap, C.byref(p)), "pedp_host_alloc")
        ptr = p.value
    raw = np.asarray(_Lease(ptr, cap))
    return raw[:n].view(dtype).reshape(shape)


def device_count():
    n = C.c_int(0)
    check(load().pedp_device_count(C.byref(n)), "pedp_device_count")
    return n.value


class Context:
    """One GPU + one HIP stream.  `stream` may be a raw hipStream_t (int) of a torch.cuda.Stream
    (`.cuda_stream`), so torch work and torch.distributed collectives issued on that stream order
    with the library's kernels.  None / 0 (torch's DEFAULT stream has handle 0) makes the library
    create its own non-blocking stream, which is not ordered with torch: synchronise explicitly."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        self.device = device
        self.stream_handle = int(stream) if stream else None  # None: the library created its own stream
        lib = load()
        check(lib.pedp_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(self._h)),
              "pedp_ctx_create")

    def synchronize(self):
        check(load().pedp_ctx_synchronize(self._h), "pedp_ctx_synchronize")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            load().pedp_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=0):
    """Process-wide context per device (own stream)."""
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = _default_ctx[device] = Context(device)
    return ctx


class Pinhole(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("width", C.c_int32), ("height", C.c_int32)]


class DepthEntryParams(C.Structure):
    """pedp_depth_entry_params (include/pedp.h)."""
    _fields_ = [("erode_radius", C.c_int32), ("erode_diff", C.c_float), ("erode_ratio", C.c_float), ("erode_zfar", C.c_float),
                ("bilateral_radius", C.c_int32), ("bilateral_zfar", C.c_float), ("sigmaD", C.c_float), ("sigmaR", C.c_float),
                ("K", C.c_float * 9), ("xyz_zfar", C.c_float), ("z_min", C.c_float), ("scale", C.c_double)]


class ProjectOpts(C.Structure):
    """pedp_project_opts (include/pedp.h)."""
    _fields_ = [("heat_f32", C.c_int32), ("heat_mem", C.c_int32), ("jet_lut", C.c_void_p), ("colors", C.c_void_p),
                ("post", C.c_void_p)]


class Mesh:
    """Device-resident triangle records of one posed mesh (RaycastingScene stand-in).
    posable=True keeps the model-frame float64 vertices on the device: set_pose(T) then stands for
    deepcopy + mesh.transform(T) + from_legacy (float64 transform, float32 cast) without a
    round trip through the host."""

    def __init__(self, ctx, vertices, triangles, posable=False):
        self.ctx = ctx
        v = np.ascontiguousarray(vertices, dtype=np.float64 if posable else np.float32).reshape(-1, 3)
        t = np.asarray(triangles)
        if t.size and (t.min() < 0):
            raise PedpError("negative vertex index")
        t = np.ascontiguousarray(t, dtype=np.uint32).reshape(-1, 3)
        self.V, self.F = len(v), len(t)
        self.posable = bool(posable)
        self._hit_cap = 0        # project_heatmap: room kept for the hits (last call's count and a margin)
        self._h = C.c_void_p()
        if posable:
            check(load().pedp_mesh_create_posable(ctx._h, _ptr(v), self.V, _ptr(t), self.F, C.byref(self._h)),
                  "pedp_mesh_create_posable")
        else:
            check(load().pedp_mesh_create(ctx._h, _ptr(v), self.V, _ptr(t), self.F, C.byref(self._h)),
                  "pedp_mesh_create")

    def set_pose(self, T=None):
        """Move the resident model-frame vertices by the 4x4 float64 T (None = identity)."""
        M = None if T is None else np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
        check(load().pedp_mesh_set_pose(self._h, _ptr(M)), "pedp_mesh_set_pose")
        return self

    def posed_vertices(self, T):
        """V x 3 float64 vertices of the resident model moved by T (pedp_mesh_posed_vertices); the mesh's pose stays."""
        if not self.posable:
            raise PedpError("posed_vertices needs a posable mesh")
        M = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
        out = host_array((self.V, 3), np.float64)
        check(load().pedp_mesh_posed_vertices(self._h, _ptr(M), HOST, _ptr(out)), "pedp_mesh_posed_vertices")
        return out

    def project_heatmap(self, heatmap, intrinsic_matrix, threshold=0.5, origin=(0.0, 0.0, 0.0), jet_lut=None, post=None):
        """heatmap_to_points + compute_rays + intersect_rays_with_mesh fused on the device (pedp_project_heatmap_ex).
        heatmap: H x W float64 or float32, a numpy array or a CUDA tensor (taken where and as it is).  jet_lut (256 x 3):
        also return the hits' jet colours (create_intersection_pcd); post (4 x 4): the hit points moved by it on the device.
        Returns dict(points M x 3 f64, intensities M f64, pixels M x 2 int32 (x, y), primitive_ids M u32, n_rays
        [, colors M x 3 f64])."""
        opts = ProjectOpts(0, HOST, None, None, None)
        keep = []
        if type(heatmap).__module__.startswith("torch"):
            import torch

            if not heatmap.is_cuda or heatmap.device.index != self.ctx.device:
                raise PedpError(f"heat map tensor must be on cuda:{self.ctx.device}")
            h = heatmap if heatmap.dtype in (torch.float32, torch.float64) else heatmap.double()
            h = h.contiguous()
            torch.cuda.current_stream(h.device).synchronize()       # the library reads it on its own stream
            opts.heat_f32, opts.heat_mem, h_ptr, shape = int(h.dtype == torch.float32), DEVICE, C.c_void_p(h.data_ptr()), tuple(h.shape)
            keep.append(h)
        else:
            h = np.asarray(heatmap)
            h = np.ascontiguousarray(h, dtype=np.float32 if h.dtype == np.float32 else np.float64)
            opts.heat_f32, h_ptr, shape = int(h.dtype == np.float32), _ptr(h), h.shape
        if len(shape) != 2:
            raise PedpError("heat map must be 2-D")
        K = np.asarray(intrinsic_matrix, dtype=np.float64)
        cam = Pinhole(K[0, 0], K[1, 1], K[0, 2], K[1, 2], shape[1], shape[0])
        o = np.ascontiguousarray(origin, dtype=np.float64).reshape(3)
        cap = self._hit_cap if self._hit_cap > 0 else max(shape[0] * shape[1], 1)   # room for the hits: every pixel the first
        # time, then the last call's count and a margin, grown on demand
        while True:
            pts = host_array((cap, 3), np.float64)
            inten = host_array((cap,), np.float64)
            pix = host_array((cap, 2), np.int32)
            prim = host_array((cap,), np.uint32)
            cols = None
            if jet_lut is not None:
                lut = np.ascontiguousarray(jet_lut, dtype=np.float64).reshape(256, 3)
                cols = host_array((cap, 3), np.float64)
                opts.jet_lut, opts.colors = lut.ctypes.data, cols.ctypes.data
                keep.append(lut)
            if post is not None:
                pT = np.ascontiguousarray(post, dtype=np.float64).reshape(16)
                opts.post = pT.ctypes.data
                keep.append(pT)
            n_rays, n_hits = C.c_int64(), C.c_int64()
            rc = load().pedp_project_heatmap_ex(self.ctx._h, self._h, C.byref(cam), h_ptr, float(threshold), _ptr(o), HOST,
                                                cap, _ptr(pts), _ptr(inten), _ptr(pix), _ptr(prim), C.byref(n_rays),
                                                C.byref(n_hits), C.byref(opts))
            if rc != 0 and n_hits.value > cap:   # more hits than room (the count is reported): once more with room for them
                cap = n_hits.value
                continue
            check(rc, "pedp_project_heatmap")
            break
        m = n_hits.value
        self._hit_cap = m + m // 4 + 1024
        out = {"points": pts[:m], "intensities": inten[:m], "pixels": pix[:m], "primitive_ids": prim[:m], "n_rays": n_rays.value}
        if cols is not None:
            out["colors"] = cols[:m]
        return out

    def cast_rays(self, rays6, want_uv=True):
        """rays6: N x 6 float32 host array.  Returns dict like RaycastingScene.cast_rays:
        t_hit (f32, inf on miss), primitive_ids (u32, 0xFFFFFFFF on miss), primitive_uvs."""
        r = np.ascontiguousarray(rays6, dtype=np.float32).reshape(-1, 6)
        n = len(r)
        t = np.empty(n, np.float32)
        ids = np.empty(n, np.uint32)
        uv = np.empty((n, 2), np.float32) if want_uv else None
        check(load().pedp_raycast(self.ctx._h, self._h, _ptr(r), n, HOST, _ptr(t), _ptr(ids), _ptr(uv)),
              "pedp_raycast")
        out = {"t_hit": t, "primitive_ids": ids}
        if want_uv:
            out["primitive_uvs"] = uv
        return out

    def cast_rayset(self, rayset, want_uv=True):
        """Closest hits of a resident ray set (RaySet): host arrays like cast_rays."""
        n = rayset.N
        t = np.empty(n, np.float32)
        ids = np.empty(n, np.uint32)
        uv = np.empty((n, 2), np.float32) if want_uv else None
        check(load().pedp_raycast_rayset(self.ctx._h, self._h, rayset._h, HOST, _ptr(t), _ptr(ids), _ptr(uv)), "pedp_raycast_rayset")
        out = {"t_hit": t, "primitive_ids": ids}
        if want_uv:
            out["primitive_uvs"] = uv
        return out

    def cast_rayset_device(self, rayset, t_ptr, id_ptr, uv_ptr=None):
        """The same into device memory (torch tensors' data_ptr()); asynchronous on the stream."""
        check(load().pedp_raycast_rayset(self.ctx._h, self._h, rayset._h, DEVICE, C.c_void_p(t_ptr), C.c_void_p(id_ptr),
                                         C.c_void_p(uv_ptr) if uv_ptr else None), "pedp_raycast_rayset")

    def cast_rays_device(self, rays_ptr, n, t_ptr, id_ptr, uv_ptr=None):
        """Device-pointer variant (torch tensors' data_ptr()); asynchronous on the stream."""
        check(load().pedp_raycast(self.ctx._h, self._h, C.c_void_p(rays_ptr), int(n), DEVICE, C.c_void_p(t_ptr),
                                  C.c_void_p(id_ptr), C.c_void_p(uv_ptr) if uv_ptr else None), "pedp_raycast")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            load().pedp_mesh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RaySet:
    """A camera's rays as a resident object (pedp_rayset_*): the N x 6 float32 rows are copied once and the grid of the
    triangle-driven ray stage is built once; casts against it are the triangle kernels and the result kernel only."""

    def __init__(self, ctx, rays6=None, device_ptr=None, n=None):
        self.ctx = ctx
        self._h = C.c_void_p()
        if device_ptr is not None:
            self.N = int(n)
            check(load().pedp_rayset_create(ctx._h, C.c_void_p(device_ptr), self.N, DEVICE, C.byref(self._h)), "pedp_rayset_create")
        else:
            r = np.ascontiguousarray(rays6, dtype=np.float32).reshape(-1, 6)
            self.N = len(r)
            check(load().pedp_rayset_create(ctx._h, _ptr(r), self.N, HOST, C.byref(self._h)), "pedp_rayset_create")

    def last_variant(self):
        v, st = C.c_int(), C.c_int()
        check(load().pedp_rayset_last_variant(self._h, C.byref(v), C.byref(st)), "pedp_rayset_last_variant")
        return v.value, st.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            load().pedp_rayset_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Cloud:
    """Device-resident float64 point cloud (+ normals)."""

    def __init__(self, ctx, points, normals=None):
        self.ctx = ctx
        p = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        nrm = None
        if normals is not None and len(normals):
            nrm = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
            if len(nrm) != len(p):
                raise PedpError("normals and points differ in length")
        self.N = len(p)
        self.has_normals = nrm is not None
        self._h = C.c_void_p()
        check(load().pedp_cloud_create(ctx._h, _ptr(p), _ptr(nrm), self.N, C.byref(self._h)), "pedp_cloud_create")

    @classmethod
    def from_device(cls, ctx, points_ptr, n, normals_ptr=None):
        """Cloud from device memory: N x 3 float64 at `points_ptr` (e.g. tensor.data_ptr() of a
        contiguous torch.float64 CUDA tensor) on the context's GPU; the data are copied, and the copy is
        complete when the call returns (the tensor may be freed or overwritten at once).  The caller
        makes sure the producing work is complete or ordered on the context's stream."""
        self = cls.__new__(cls)
        self.ctx, self.N, self.has_normals = ctx, int(n), normals_ptr is not None
        self._h = C.c_void_p()
        check(load().pedp_cloud_create_device(ctx._h, C.c_void_p(points_ptr), C.c_void_p(normals_ptr) if normals_ptr else None,
                                              self.N, C.byref(self._h)), "pedp_cloud_create_device")
        return self

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            load().pedp_cloud_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def raycast_configure(ctx, tri_chunks=0, variant=0):
    check(load().pedp_raycast_configure(ctx._h, int(tri_chunks), int(variant)), "pedp_raycast_configure")


def debug_rast_rects(ctx, mesh, rays6):
    """pedp_debug_rast_rects: (tri [F, 12], ray [N, 3], (GX, GY, grid_status)) of a variant-4 cast of these rays."""
    r = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6)
    tri = np.zeros((mesh.F, 12), np.float32)
    ray = np.zeros((len(r), 3), np.float32)
    grid = np.zeros(3, np.int32)
    check(load().pedp_debug_rast_rects(ctx._h, mesh._h, _ptr(r), len(r), _ptr(tri), _ptr(ray), _ptr(grid)), "pedp_debug_rast_rects")
    return tri, ray, tuple(int(x) for x in grid)


def raycast_last_sweep_ms(ctx):
    ms = C.c_float(0)
    check(load().pedp_raycast_last_sweep_ms(ctx._h, C.byref(ms)), "pedp_raycast_last_sweep_ms")
    return ms.value


def ransac_hypotheses(ctx, src, tgt, corr, seed, itr0, count, edge_similarity, max_distance, normal_angle):
    """(accepted[count] bool, T[count, 4, 4]) of RANSAC draws itr0 .. itr0 + count - 1 (pedp_ransac_hypotheses)."""
    c = np.ascontiguousarray(corr, np.int32)
    ok = np.empty(int(count), np.uint8)
    T = np.empty((int(count), 4, 4), np.float64)
    check(load().pedp_ransac_hypotheses(ctx._h, src._h, tgt._h, _ptr(c), C.c_uint64(int(seed)), int(itr0), int(count),
                                        float(edge_similarity), float(max_distance), float(normal_angle), _ptr(ok), _ptr(T)),
          "pedp_ransac_hypotheses")
    return ok.astype(bool), T


def raycast_last_variant(ctx):
    """(variant the last cast ran, grid status of a variant-4 cast: 0 = the grid answered it)."""
    v, g = C.c_int(0), C.c_int(0)
    check(load().pedp_raycast_last_variant(ctx._h, C.byref(v), C.byref(g)), "pedp_raycast_last_variant")
    return v.value, g.value


def nn_last_sweep_ms(ctx):
    ms = C.c_float(0)
    check(load().pedp_nn_last_sweep_ms(ctx._h, C.byref(ms)), "pedp_nn_last_sweep_ms")
    return ms.value


def icp_configure(ctx, exhaustive=False, timed_pass=-1):
    """Measurement knobs of pedp_icp on this context: exhaustive = all-pairs sweep in every pass
    (same results), timed_pass = the pass whose sweep kernel gets HIP events (nn_last_sweep_ms)."""
    check(load().pedp_icp_configure(ctx._h, int(bool(exhaustive)), int(timed_pass)), "pedp_icp_configure")


COMM_ID_BYTES = 128


def comm_unique_id():
    """128-byte RCCL unique id (rank 0 makes it; every rank passes the same bytes to comm_create)."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    check(load().pedp_comm_unique_id(buf), "pedp_comm_unique_id")
    return bytes(buf)


def comm_create(ctx, unique_id, nranks, rank):
    """Give the context its own RCCL communicator (collective: every rank calls it)."""
    if len(unique_id) != COMM_ID_BYTES:
        raise PedpError("comm_create: the unique id must be 128 bytes")
    buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
    check(load().pedp_comm_create(ctx._h, buf, int(nranks), int(rank)), "pedp_comm_create")


def comm_destroy(ctx):
    check(load().pedp_comm_destroy(ctx._h), "pedp_comm_destroy")


def comm_size(ctx):
    n, r = C.c_int(1), C.c_int(0)
    check(load().pedp_comm_size(ctx._h, C.byref(n), C.byref(r)), "pedp_comm_size")
    return n.value, r.value


def comm_allgather(ctx, send_ptr, recv_ptr, bytes_per_rank):
    """Device pointers; enqueued on the context's stream."""
    check(load().pedp_comm_allgather(ctx._h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), int(bytes_per_rank)),
          "pedp_comm_allgather")


def comm_allreduce_f64(ctx, ptr, n):
    check(load().pedp_comm_allreduce_f64(ctx._h, C.c_void_p(ptr), int(n)), "pedp_comm_allreduce_f64")


def icp_last_stats(ctx):
    """(passes, pairs swept by the MFMA kernel, points sent to the exact fallback) of the last icp()."""
    a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    check(load().pedp_icp_last_stats(ctx._h, C.byref(a), C.byref(b), C.byref(c)), "pedp_icp_last_stats")
    return a.value, b.value, c.value


def icp_last_planned_passes(ctx):
    """Passes of the last single icp() that ran under a visit plan (more live chunks than CUs; scheduling only)."""
    a = C.c_int64(0)
    check(load().pedp_icp_last_planned_passes(ctx._h, C.byref(a)), "pedp_icp_last_planned_passes")
    return a.value


def icp(ctx, source, target, max_correspondence_distance, init, estimator=POINT_TO_PLANE, max_iteration=30,
        relative_fitness=1e-6, relative_rmse=1e-6, want_corr=False, want_trace=False, allreduce=None,
        n_source_global=0, use_comm=False):
    """Raw pedp_icp call on Cloud handles.  Returns dict(T, fitness, inlier_rmse, iters[, corr, trace])."""
    prm = IcpParams()
    prm.max_correspondence_distance = float(max_correspondence_distance)
    prm.estimator = int(estimator)
    prm.max_iteration = int(max_iteration)
    prm.relative_fitness = float(relative_fitness)
    prm.relative_rmse = float(relative_rmse)
    cb = None
    if allreduce is not None:
        def _hook(user, dev_ptr, n, stream):
            try:
                allreduce(dev_ptr, n, stream)
                return 0
            except Exception:  # an exception must not unwind through C
                import traceback
                traceback.print_exc()
                return 1
        cb = ALLREDUCE_FN(_hook)
        prm.allreduce = cb
    else:
        prm.allreduce = C.cast(None, ALLREDUCE_FN)
    prm.allreduce_user = None
    prm.n_source_global = int(n_source_global)
    prm.use_comm = int(bool(use_comm))
    T0 = np.ascontiguousarray(init, dtype=np.float64).reshape(4, 4)
    T = np.empty((4, 4), np.float64)
    fit, rmse, it = C.c_double(0), C.c_double(0), C.c_int32(0)
    corr = np.empty(source.N, np.int32) if want_corr else None
    trace = np.zeros((max_iteration + 1, 18), np.float64) if want_trace else None
    check(load().pedp_icp(ctx._h, source._h, target._h, C.byref(prm), _ptr(T0), _ptr(T), C.byref(fit),
                          C.byref(rmse), C.byref(it), _ptr(corr), _ptr(trace)), "pedp_icp")
    out = {"T": T, "fitness": fit.value, "inlier_rmse": rmse.value, "iters": it.value}
    if want_corr:
        out["corr"] = corr
    if want_trace:
        out["trace"] = trace[: it.value + 1]
    return out


def debug_nn_bf16(ctx, src4, tgt4):
    """g of every (scene row, model row) pair as the dense sweep's bf16 MFMA produces it (pedp_debug_nn_bf16)."""
    a = np.ascontiguousarray(src4, np.float32).reshape(-1, 4)
    b = np.ascontiguousarray(tgt4, np.float32).reshape(-1, 4)
    g = np.empty((len(a), len(b)), np.float32)
    check(load().pedp_debug_nn_bf16(ctx._h, _ptr(a), len(a), _ptr(b), len(b), _ptr(g)), "pedp_debug_nn_bf16")
    return g


def icp_begin(ctx, source, target, max_correspondence_distance, init, estimator=POINT_TO_PLANE, max_iteration=30,
              relative_fitness=1e-6, relative_rmse=1e-6, want_trace=False, n_source_global=0, use_comm=False):
    """pedp_icp_begin: every pass of the registration is enqueued, the call returns at once; `icp_end(ctx, ...)` waits and
    returns pedp_icp's result.  The context is the registration's until then."""
    prm = IcpParams()
    prm.max_correspondence_distance = float(max_correspondence_distance)
    prm.estimator = int(estimator)
    prm.max_iteration = int(max_iteration)
    prm.relative_fitness = float(relative_fitness)
    prm.relative_rmse = float(relative_rmse)
    prm.allreduce = C.cast(None, ALLREDUCE_FN)
    prm.allreduce_user = None
    prm.n_source_global = int(n_source_global)
    prm.use_comm = int(bool(use_comm))
    T0 = np.ascontiguousarray(init, dtype=np.float64).reshape(4, 4)
    check(load().pedp_icp_begin(ctx._h, source._h, target._h, C.byref(prm), _ptr(T0), 1 if want_trace else 0), "pedp_icp_begin")
    ctx._icp_pending = (source, target, int(max_iteration), bool(want_trace))   # (the handles stay alive until the end)


def icp_end(ctx, want_corr=False):
    """pedp_icp_end: dict(T, fitness, inlier_rmse, iters[, corr, trace]) of the registration icp_begin started."""
    pending = getattr(ctx, "_icp_pending", None)
    if pending is None:
        raise PedpError("icp_end: no registration is pending on this context")
    source, _, max_iteration, want_trace = pending
    ctx._icp_pending = None
    T = np.empty((4, 4), np.float64)
    fit, rmse, it = C.c_double(0), C.c_double(0), C.c_int32(0)
    corr = np.empty(source.N, np.int32) if want_corr else None
    trace = np.zeros((max_iteration + 1, 18), np.float64) if want_trace else None
    check(load().pedp_icp_end(ctx._h, _ptr(T), C.byref(fit), C.byref(rmse), C.byref(it), _ptr(corr), _ptr(trace)), "pedp_icp_end")
    out = {"T": T, "fitness": fit.value, "inlier_rmse": rmse.value, "iters": it.value}
    if want_corr:
        out["corr"] = corr
    if want_trace:
        out["trace"] = trace[: it.value + 1]
    return out


def icp_batched(ctx, source, target, max_correspondence_distance, inits, estimator=POINT_TO_PLANE, max_iteration=30):
    prm = IcpParams()
    prm.max_correspondence_distance = float(max_correspondence_distance)
    prm.estimator = int(estimator)
    prm.max_iteration = int(max_iteration)
    prm.relative_fitness = -1.0
    prm.relative_rmse = -1.0
    prm.allreduce = C.cast(None, ALLREDUCE_FN)
    prm.allreduce_user = None
    prm.n_source_global = 0
    prm.use_comm = 0
    I = np.ascontiguousarray(inits, dtype=np.float64).reshape(-1, 16)
    B = len(I)
    T = np.empty((B, 4, 4), np.float64)
    fit = np.empty(B, np.float64)
    rmse = np.empty(B, np.float64)
    check(load().pedp_icp_batched(ctx._h, source._h, target._h, C.byref(prm), _ptr(I), B, _ptr(T), _ptr(fit),
                                  _ptr(rmse)), "pedp_icp_batched")
    return T, fit, rmse


def icp_batched_ex(ctx, source, target, radii, inits, estimator=POINT_TO_PLANE, max_iteration=30,
                   relative_fitness=1e-6, relative_rmse=1e-6):
    """B registrations with their own radius each (and the given criteria, a scalar or one per pose),
    each stopping by its own criteria.  Returns (T [B,4,4], fitness [B], rmse [B], iterations [B])."""
    I = np.ascontiguousarray(inits, dtype=np.float64).reshape(-1, 16)
    B = len(I)
    rad = np.broadcast_to(np.asarray(radii, dtype=np.float64), (B,))
    rf = np.broadcast_to(np.asarray(relative_fitness, dtype=np.float64), (B,))
    rr = np.broadcast_to(np.asarray(relative_rmse, dtype=np.float64), (B,))
    prms = (IcpParams * max(B, 1))()
    for b in range(B):
        q = prms[b]
        q.max_correspondence_distance = float(rad[b])
        q.estimator = int(estimator)
        q.max_iteration = int(max_iteration)
        q.relative_fitness = float(rf[b])
        q.relative_rmse = float(rr[b])
        q.allreduce = C.cast(None, ALLREDUCE_FN)
        q.allreduce_user = None
        q.n_source_global = 0
        q.use_comm = 0
    T = np.empty((B, 4, 4), np.float64)
    fit, rmse = np.empty(B, np.float64), np.empty(B, np.float64)
    its = np.empty(B, np.int32)
    check(load().pedp_icp_batched_ex(ctx._h, source._h, target._h, prms, _ptr(I), B, _ptr(T), _ptr(fit), _ptr(rmse),
                                     _ptr(its)), "pedp_icp_batched_ex")
    return T, fit, rmse, its


def nn(ctx, source, target, T=None):
    """Exact nearest neighbour of every (transformed) source point: (idx int32, d2 float64)."""
    M = np.ascontiguousarray(np.eye(4) if T is None else T, dtype=np.float64)
    idx = np.empty(source.N, np.int32)
    d2 = np.empty(source.N, np.float64)
    check(load().pedp_nn(ctx._h, source._h, target._h, _ptr(M), _ptr(idx), _ptr(d2)), "pedp_nn")
    return idx, d2


def cluster_poses(angle_diff, dist_diff, poses_in, symmetry_tfs):
    """mycpp.cluster_poses stand-in (estimater.py:118): returns the kept 4x4 float32 poses."""
    p = np.ascontiguousarray(poses_in, dtype=np.float32).reshape(-1, 16)
    s = np.ascontiguousarray(symmetry_tfs, dtype=np.float32).reshape(-1, 16)
    keep = np.empty(max(len(p), 1), np.int32)
    nk = C.c_int(0)
    check(load().pedp_cluster_poses(float(angle_diff), float(dist_diff), _ptr(p), len(p), _ptr(s), len(s),
                                    _ptr(keep), C.byref(nk)), "pedp_cluster_poses")
    return keep[: nk.value].copy()


def transform_points(T, points, rotate_only=False):
    """pedp_transform_points: rows of `points` (N x 3 float64) moved by the 4x4 `T` (rotation only for normals), the
    oracle's operation order; a fresh array.  Host arithmetic of the library (no GPU, no context)."""
    p = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    M = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    out = np.empty_like(p)
    check(load().pedp_transform_points(_ptr(M), _ptr(p), len(p), 1 if rotate_only else 0, _ptr(out)), "pedp_transform_points")
    return out


def debug_mfma_scores(ctx, mesh, rays6):
    """pedp_debug_mfma_scores: (score, slack), N x F float32 each, of the exhaustive sweep's matrix-pipe filter."""
    r = np.ascontiguousarray(rays6, dtype=np.float32).reshape(-1, 6)
    score = np.empty((len(r), mesh.F), np.float32)
    slack = np.empty((len(r), mesh.F), np.float32)
    check(load().pedp_debug_mfma_scores(ctx._h, mesh._h, _ptr(r), len(r), _ptr(score), _ptr(slack)), "pedp_debug_mfma_scores")
    return score, slack
