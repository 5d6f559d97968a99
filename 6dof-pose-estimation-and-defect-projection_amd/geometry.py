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


class PointCloud:
    """Open3D-shaped holder: `points` / `normals` / `colors` are N x 3 float64 numpy arrays.

    `points` may also be given as a float64 N x 3 torch tensor on the GPU (a scene back-projected there,
    estimater.py hands CUDA tensors around): the holder keeps the tensor, `voxel_down_sample` works from it,
    and the numpy array is only made if somebody reads `points`.

    A holder made by `moved_copy` (the moved model `refine_pose_with_icp` returns and run.py:99 throws away) forms its
    points and normals when they are first read: until then it keeps the source's arrays as they were handed over and the
    4x4.  Our holders replace their arrays when they change, they do not write into them; a caller that edits the
    source's arrays in place before reading the copy should read the copy first."""

    def __init__(self, points=None, normals=None, colors=None):
        self._dev_points = None
        self._moved = None            # (points, normals, T): a moved copy that nobody has read yet
        if _on_device(points):
            self._dev_points, self._points = points.reshape(-1, 3), None
        else:
            self._points = _arr([] if points is None else points)
        self._normals = _arr([] if normals is None else normals)
        self._uniform = None          # paint_uniform_color: one colour for every point, written out when read
        self.colors = _arr([] if colors is None else colors)

    @classmethod
    def moved_copy(cls, source, T):
        """What copy.deepcopy(source).transform(T) gives (same arithmetic, pose_estimation.py:815-816), formed on first read."""
        out = cls(colors=None if source._uniform is not None else np.array(source._colors))
        if source._uniform is not None:
            out.paint_uniform_color(source._uniform)
        out._points = None
        out._moved = (np.asarray(source.points, np.float64), np.asarray(source.normals, np.float64), np.array(T, dtype=np.float64))
        return out

    def _form(self):
        if self._moved is not None:
            pts, nrm, T = self._moved
            self._moved = None
            self._points = pts @ T[:3, :3].T + T[:3, 3]
            self._normals = nrm @ T[:3, :3].T if len(nrm) else _arr([])

    @property
    def points(self):
        self._form()
        if self._points is None:
            self._points = _arr(self._dev_points.detach().cpu().numpy())
        return self._points

    @points.setter
    def points(self, value):
        self._form()
        self._points, self._dev_points = value, None

    @property
    def normals(self):
        self._form()
        return self._normals

    @normals.setter
    def normals(self, value):
        self._form()
        self._normals = value

    def _count(self):
        if self._moved is not None:
            return len(self._moved[0])
        return len(self._dev_points) if self._points is None else len(self._points)

    @property
    def colors(self):
        if self._uniform is not None:
            self._colors = np.tile(self._uniform, (self._count(), 1))
            self._uniform = None
        return self._colors

    @colors.setter
    def colors(self, value):
        self._colors = value
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
        """In place, float64, like Open3D: points by the full 4x4, normals by the rotation."""
        T = np.asarray(T, dtype=np.float64)
        self.points = np.asarray(self.points, np.float64) @ T[:3, :3].T + T[:3, 3]
        if len(self.normals):
            self.normals = np.asarray(self.normals, np.float64) @ T[:3, :3].T
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
        out = PointCloud(np.array(self.points), np.array(self.normals), None if self._uniform is not None else np.array(self._colors))
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
        out = PointCloud(np.asarray(self.points)[idx], self.normals[idx] if self.has_normals() else None,
                         self._colors[idx] if self._uniform is None and self.has_colors() else None)
        if self._uniform is not None:
            out.paint_uniform_color(self._uniform)
        return out

    def voxel_down_sample(self, voxel_size):
        from . import cloud_ops

        if self._dev_points is not None and self._points is None and not self.has_normals():      # device-resident scene: the grid is built from it
            pts, nrm = cloud_ops.voxel_down_sample(self._dev_points, voxel_size)
        else:
            pts, nrm = cloud_ops.voxel_down_sample(self.points, voxel_size, self.normals if self.has_normals() else None)
        return PointCloud(pts, nrm)

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
        self.normals = cloud_ops.estimate_normals(self.points, sp.radius, sp.max_nn,
                                                  self.normals if self.has_normals() else None)
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
        self.vertices = np.asarray(self.vertices, np.float64) @ T[:3, :3].T + T[:3, 3]
        if len(self.vertex_normals):
            self.vertex_normals = self.vertex_normals @ T[:3, :3].T
        if len(self.triangle_normals):
            self.triangle_normals = self.triangle_normals @ T[:3, :3].T
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
        self.points = self.points @ T[:3, :3].T + T[:3, 3]
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
