"""Plain holders with the attribute surface the reference uses on Open3D objects along the
hot path.  Open3D is not importable on the GPU box, so the
drop-in functions work on these -- and on real Open3D objects too, because every access
goes through `np.asarray(obj.points)`-style duck typing.

    o3d.geometry.PointCloud        .points .normals .colors .has_normals() .has_colors()
                                   .transform(T) .paint_uniform_color(rgb)
                                   (src/pose_estimation.py:162-169, :406-409, :769-770)
    o3d.geometry.TriangleMesh      .vertices .triangles .transform(T)
                                   (src/defect_projection.py:240-245, :549-550)
    o3d.camera.PinholeCameraIntrinsic   .intrinsic_matrix (src/defect_projection.py:209-212)
    o3d.pipelines.registration.RegistrationResult   .transformation .fitness .inlier_rmse
                                   .correspondence_set (src/pose_estimation.py:566-569, :617-620)
"""
import copy

import numpy as np


def _arr(x, cols=3, dtype=np.float64):
    a = np.asarray(x, dtype=dtype)
    return a.reshape(-1, cols) if a.size else np.zeros((0, cols), dtype)


def _on_device(x):
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)


def _frozen(a):
    """True when nobody can write to the array's memory through numpy: it (or the array it views) owns the data and is
    read-only.  Such arrays are passed between holders without a copy."""
    if a.flags.writeable:
        return False
    base = a if a.flags.owndata else a.base
    return isinstance(base, np.ndarray) and base.flags.owndata and not base.flags.writeable


def _own(x, cols=3, dtype=np.float64, adopt=False):
    """The holder's own, immutable array of `x` (N x cols): frozen arrays are taken as they are; `adopt` freezes the
    caller's fresh array in place (the caller gives it up); everything else is copied, like Open3D's Vector3dVector(array)."""
    a = np.asarray(x, dtype=dtype)
    fresh = a is not x and a.base is None                       # np.asarray had to convert: nobody else holds the result
    a = a.reshape(-1, cols) if a.size else np.zeros((0, cols), dtype)
    if _frozen(a):
        return a
    base = a if a.flags.owndata else a.base
    if not ((adopt or fresh) and isinstance(base, np.ndarray) and base.flags.owndata and a.flags.c_contiguous):
        a = np.array(a, dtype=dtype, order="C")
        base = a
    base.setflags(write=False)
    a.setflags(write=False)
    return a


def rigid(points, T, rotate_only=False):
    """Rows moved by the 4x4 T (normals: rotation only) -- pedp_transform_points, the oracle's operation order."""
    from . import _lib

    if len(points) == 0:
        return np.zeros((0, 3))
    return _lib.transform_points(T, points, rotate_only)


class PointCloud:
    """Open3D-shaped holder: `points` / `normals` / `colors` are N x 3 float64 numpy arrays.

    The holder OWNS its arrays, like an Open3D cloud owns its vectors: the constructor and the setters copy what they are
    given (Vector3dVector(array) does), and what the accessors hand out is read-only -- an in-place edit
    (`np.asarray(pcd.points)[mask] = ...`, which Open3D lets through) raises instead of leaving a stale device copy
    behind; assign a new array (`pcd.points = edited`).  That makes the kept device copy of a holder exact: it is valid as
    long as the holder's version counter, bumped by every setter and transform, has not moved (pedp_hip.registration.upload).
    `PointCloud.adopt(points, normals)` takes fresh arrays without a copy (the caller gives them up);
    `PointCloud.borrowed(points)` wraps a large array the caller keeps -- a frame's full scene cloud in a pinned buffer --
    without a copy and WITHOUT a kept device copy (the caller may rewrite the buffer for the next frame).

    `points` may also be given as a float64 N x 3 torch tensor on the GPU (a scene back-projected there,
    estimater.py hands CUDA tensors around): the holder keeps the tensor, `voxel_down_sample` works from it,
    and the numpy array is only made if somebody reads `points`.

    A holder made by `moved_copy` (the moved model `refine_pose_with_icp` returns and run.py:99 throws away) forms its
    points and normals when they are first read: until then it keeps the source's (immutable) arrays and the 4x4."""

    def __init__(self, points=None, normals=None, colors=None, _adopt=False):
        self._dev_points = None
        self._moved = None            # (points, normals, T): a moved copy that nobody has read yet
        self._version = 0             # bumped whenever points or normals change: keys the kept device copy
        self._borrowed = False
        if _on_device(points):
            self._dev_points, self._points = points.reshape(-1, 3), None
        else:
            self._points = _own([] if points is None else points, adopt=_adopt)
        self._normals = _own([] if normals is None else normals, adopt=_adopt)
        self._uniform = None          # paint_uniform_color: one colour for every point, written out when read
        self._colors = _own([] if colors is None else colors, adopt=_adopt)

    @classmethod
    def adopt(cls, points, normals=None, colors=None):
        """A holder over fresh arrays the caller hands over for good (results of the library's operations): no copy."""
        return cls(points, normals, colors, _adopt=True)

    @classmethod
    def borrowed(cls, points, device=None):
        """A holder over an array the caller keeps and may rewrite after the holder has served (a frame's scene cloud in a
        pinned buffer): no copy, no kept device copy.  `device`: the float64 N x 3 CUDA tensor the array was downloaded
        from, if the caller still has it -- the GPU operations then read it instead of uploading the same bytes again
        (valid as long as the holder's points are not assigned or transformed, which drop it)."""
        out = cls()
        a = np.asarray(points, dtype=np.float64)
        out._points = a.reshape(-1, 3) if a.size else np.zeros((0, 3))
        out._borrowed = True
        if device is not None:
            if not _on_device(device) or device.numel() != out._points.size:
                raise ValueError("PointCloud.borrowed: `device` must be the CUDA tensor the points were downloaded from")
            out._dev_points = device.reshape(-1, 3)
        return out

    @classmethod
    def moved_copy(cls, source, T):
        """What copy.deepcopy(source).transform(T) gives (same arithmetic, pose_estimation.py:815-816), formed on first read."""
        out = cls(colors=None if source._uniform is not None else source._colors)
        if source._uniform is not None:
            out.paint_uniform_color(source._uniform)
        out._points = None
        pts, nrm = source.points, source.normals
        if source._borrowed:
            pts = np.array(pts)
        out._moved = (pts, nrm, np.array(T, dtype=np.float64))
        return out

    def _form(self):
        if self._moved is not None:
            pts, nrm, T = self._moved
            self._moved = None
            self._points = _own(rigid(pts, T), adopt=True)
            self._normals = _own(rigid(nrm, T, rotate_only=True) if len(nrm) else [], adopt=True)

    @property
    def points(self):
        self._form()
        if self._points is None:
            self._points = _own(self._dev_points.detach().cpu().numpy(), adopt=True)
        return self._points

    @points.setter
    def points(self, value):
        self._form()
        self._points, self._dev_points, self._borrowed = _own(value), None, False
        self._version += 1

    @property
    def normals(self):
        self._form()
        return self._normals

    @normals.setter
    def normals(self, value):
        self._form()
        self._normals = _own(value)
        self._version += 1

    def _count(self):
        if self._moved is not None:
            return len(self._moved[0])
        return len(self._dev_points) if self._points is None else len(self._points)

    @property
    def colors(self):
        if self._uniform is not None:
            self._colors = _own(np.tile(self._uniform, (self._count(), 1)), adopt=True)
            self._uniform = None
        return self._colors

    @colors.setter
    def colors(self, value):
        self._colors = _own(value)
        self._uniform = None

    def has_normals(self):
        n = len(self._moved[1]) if self._moved is not None else len(self._normals)
        return n == self._count() and self._count() > 0

    def has_colors(self):
        if self._uniform is not None:
            return self._count() > 0
        return len(self._colors) == self._count() and self._count() > 0

    def has_points(self):
        return self._count() > 0

    def transform(self, T):
        """In place, float64, like Open3D: points by the full 4x4, normals by the rotation (pedp_transform_points)."""
        T = np.asarray(T, dtype=np.float64)
        pts, nrm = self.points, self.normals
        self._points, self._dev_points, self._borrowed = _own(rigid(pts, T), adopt=True), None, False
        if len(nrm):
            self._normals = _own(rigid(nrm, T, rotate_only=True), adopt=True)
        self._version += 1
        return self

    def paint_uniform_color(self, rgb):
        """One colour for every point (pose_estimation.py:769-770 paints both clouds every frame); the N x 3
        array is written out when `colors` is read."""
        self._colors = np.zeros((0, 3))
        self._uniform = np.asarray(rgb, np.float64).reshape(3).copy()
        return self

    def __len__(self):
        return self._count()

    def __deepcopy__(self, memo):
        out = PointCloud(self.points, self.normals, None if self._uniform is not None else self._colors)   # (immutable arrays: shared)
        if self._borrowed:
            out._points = _own(np.array(self._points), adopt=True)
        if self._uniform is not None:
            out.paint_uniform_color(self._uniform)
        return out

    # ---- the Open3D methods preprocess_source calls (src/pose_estimation.py:186-268); GPU work in
    # pedp_hip.cloud_ops.  Colours are not carried through voxel_down_sample (nothing downstream
    # reads them).
    def select_by_index(self, indices, invert=False):
        idx = np.asarray(indices, dtype=np.int64).reshape(-1)
        if invert:
            mask = np.ones(len(self.points), bool)
            mask[idx] = False
            idx = np.nonzero(mask)[0]
        out = PointCloud.adopt(np.asarray(self.points)[idx], self.normals[idx] if self.has_normals() else None,
                               self._colors[idx] if self._uniform is None and self.has_colors() else None)
        if self._uniform is not None:
            out.paint_uniform_color(self._uniform)
        return out

    def voxel_down_sample(self, voxel_size):
        from . import cloud_ops

        if self._dev_points is not None and not self.has_normals():      # device-resident scene (or its twin on the device): the grid is built from it
            pts, nrm = cloud_ops.voxel_down_sample(self._dev_points, voxel_size)
        else:
            pts, nrm = cloud_ops.voxel_down_sample(self.points, voxel_size, self.normals if self.has_normals() else None)
        return PointCloud.adopt(pts, nrm)

    def segment_plane(self, distance_threshold, ransac_n, num_iterations, probability=0.99999999):
        from . import cloud_ops

        plane, inliers = cloud_ops.segment_plane(self.points, distance_threshold, ransac_n, num_iterations)
        return plane, inliers  # int32 array (Open3D: list of int); select_by_index takes either

    def cluster_dbscan(self, eps, min_points, print_progress=False):
        from . import cloud_ops

        return cloud_ops.cluster_dbscan(self.points, eps, min_points)

    def estimate_normals(self, search_param=None, fast_normal_computation=True):
        """In place, like Open3D; search_param is a KDTreeSearchParamHybrid (radius, max_nn)."""
        from . import cloud_ops

        sp = search_param or KDTreeSearchParamHybrid(radius=0.1, max_nn=30)
        self._normals = _own(cloud_ops.estimate_normals(self.points, sp.radius, sp.max_nn,
                                                        self.normals if self.has_normals() else None), adopt=True)
        self._version += 1
        return self

    def remove_statistical_outlier(self, nb_neighbors, std_ratio, print_progress=False):
        from . import cloud_ops

        keep = cloud_ops.remove_statistical_outlier(self.points, nb_neighbors, std_ratio)
        return self.select_by_index(keep), keep


class KDTreeSearchParamHybrid:
    """o3d.geometry.KDTreeSearchParamHybrid(radius, max_nn) (src/pose_estimation.py:304-305)."""

    def __init__(self, radius, max_nn):
        self.radius, self.max_nn = float(radius), int(max_nn)


class TriangleMesh:
    def __init__(self, vertices=None, triangles=None):
        self.vertices = _arr([] if vertices is None else vertices)
        self.triangles = _arr([] if triangles is None else triangles, 3, np.int32)
        self.vertex_normals = np.zeros((0, 3))
        self.triangle_normals = np.zeros((0, 3))

    def transform(self, T):
        T = np.asarray(T, dtype=np.float64)
        self.vertices = rigid(self.vertices, T)
        if len(self.vertex_normals):
            self.vertex_normals = rigid(self.vertex_normals, T, rotate_only=True)
        if len(self.triangle_normals):
            self.triangle_normals = rigid(self.triangle_normals, T, rotate_only=True)
        return self

    def has_triangle_normals(self):
        return len(self.triangle_normals) == len(self.triangles) and len(self.triangles) > 0

    def has_vertex_normals(self):
        return len(self.vertex_normals) == len(self.vertices) and len(self.vertices) > 0

    def compute_triangle_normals(self):
        v, t = np.asarray(self.vertices, np.float64), np.asarray(self.triangles)
        n = np.cross(v[t[:, 1]] - v[t[:, 0]], v[t[:, 2]] - v[t[:, 0]])
        ln = np.linalg.norm(n, axis=1, keepdims=True)
        self.triangle_normals = n / np.where(ln > 0, ln, 1.0)
        return self

    def compute_vertex_normals(self):
        v, t = np.asarray(self.vertices, np.float64), np.asarray(self.triangles)
        fn = np.cross(v[t[:, 1]] - v[t[:, 0]], v[t[:, 2]] - v[t[:, 0]])
        acc = np.zeros_like(v)
        for k in range(3):
            np.add.at(acc, t[:, k], fn)
        ln = np.linalg.norm(acc, axis=1, keepdims=True)
        self.vertex_normals = acc / np.where(ln > 0, ln, 1.0)
        return self

    def __deepcopy__(self, memo):
        m = TriangleMesh(np.array(self.vertices), np.array(self.triangles))
        m.vertex_normals = np.array(self.vertex_normals)
        m.triangle_normals = np.array(self.triangle_normals)
        return m


class LineSet:
    """Debug rays returned by ray_tracing when nothing was hit (defect_projection.py:296-317)."""

    def __init__(self):
        self.points = np.zeros((0, 3))
        self.lines = np.zeros((0, 2), np.int32)
        self.colors = np.zeros((0, 3))

    def paint_uniform_color(self, rgb):
        self.colors = np.tile(np.asarray(rgb, np.float64), (len(self.lines), 1))
        return self

    def transform(self, T):
        T = np.asarray(T, dtype=np.float64)
        self.points = rigid(self.points, T)
        return self


class PinholeCameraIntrinsic:
    def __init__(self, width=0, height=0, fx=1.0, fy=1.0, cx=0.0, cy=0.0, intrinsic_matrix=None):
        self.width, self.height = int(width), int(height)
        if intrinsic_matrix is not None:
            self.intrinsic_matrix = np.array(intrinsic_matrix, dtype=np.float64)
        else:
            self.intrinsic_matrix = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float64)


class RegistrationResult:
    def __init__(self, transformation=None):
        self.transformation = np.eye(4) if transformation is None else np.array(transformation, dtype=np.float64)
        self.fitness = 0.0
        self.inlier_rmse = 0.0
        self.correspondence_set = np.zeros((0, 2), np.int32)

    def __repr__(self):
        return (f"RegistrationResult with fitness={self.fitness:e}, inlier_rmse={self.inlier_rmse:e}, "
                f"and correspondence_set size of {len(self.correspondence_set)}")


def points_of(obj):
    """float64 N x 3 view of a PointCloud-like object (ours, Open3D's, or a bare array)."""
    if hasattr(obj, "points"):
        return _arr(obj.points)
    return _arr(obj)


def normals_of(obj):
    if hasattr(obj, "has_normals"):
        return _arr(obj.normals) if obj.has_normals() else None
    n = getattr(obj, "normals", None)
    return _arr(n) if n is not None and len(n) else None


def clone(obj):
    return copy.deepcopy(obj)


def as_holder(obj):
    """The object itself if it is our PointCloud, otherwise a PointCloud holding its points / normals /
    colours (an Open3D cloud handed to the preprocessing chain would otherwise run Open3D's own CPU
    methods of the same names)."""
    if obj is None or isinstance(obj, PointCloud):
        return obj
    colors = np.asarray(obj.colors) if hasattr(obj, "has_colors") and obj.has_colors() else None
    return PointCloud(points_of(obj), normals_of(obj), colors)
