"""Host mirror of the reference's 2D -> 3D defect projection (src/defect_projection.py): same
names, positional signatures and return values; the ray / mesh intersection runs on the GPU
through libpedp_hip.so instead of Open3D's RaycastingScene (Embree).

    heatmap_to_points          defect_projection.py:165-179
    compute_rays               :196-223
    intersect_rays_with_mesh   :225-266
    create_intersection_pcd    :268-294
    project_debug_rays         :296-317
    load_extrinsics            :65-94
    ray_tracing                :527-563   (caller run.py:113, :187)
"""
import json
import logging

import numpy as np

from . import _lib
from .geometry import KDTreeSearchParamHybrid, LineSet, PointCloud, clone


def heatmap_to_points(heatmap, threshold=0.5):
    """Pixels above `threshold` as (x, y, intensity) tuples in row-major order (y outer):
    this order is the ray order of everything downstream."""
    rows, cols = np.nonzero(heatmap > threshold)
    return list(zip(cols, rows, heatmap[rows, cols]))


def compute_rays(points, intrinsic):
    """Unit viewing directions (float64) of the given pixels through a pinhole camera:
    normalise((x - cx) / fx, (y - cy) / fy, 1).  The reference loops in Python per pixel
    (defect_projection.py:216-222); this is the same arithmetic on whole arrays."""
    K = np.asarray(intrinsic.intrinsic_matrix, dtype=np.float64)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    if len(points) == 0:
        return np.array([]), np.array([])
    px = np.array([p[0] for p in points], dtype=np.float64)
    py = np.array([p[1] for p in points], dtype=np.float64)
    intensities = np.array([p[2] for p in points])
    d = np.stack([(px - cx) / fx, (py - cy) / fy, np.ones_like(px)], axis=1)
    d /= np.sqrt(np.einsum("ij,ij->i", d, d))[:, None]
    return d, intensities


def _device_mesh(mesh, ctx):
    """from_legacy: float32 vertices, uint32 triangles (defect_projection.py:245)."""
    if isinstance(mesh, _lib.Mesh):
        return mesh
    return _lib.Mesh(ctx, np.asarray(mesh.vertices, dtype=np.float64).astype(np.float32),
                     np.asarray(mesh.triangles))


def cast_rays(mesh, rays6, ctx=None):
    """RaycastingScene.cast_rays stand-in: dict with t_hit, primitive_ids, primitive_uvs."""
    ctx = ctx or (mesh.ctx if isinstance(mesh, _lib.Mesh) else _lib.default_context())
    return _device_mesh(mesh, ctx).cast_rays(rays6)


def intersect_rays_with_mesh(mesh, rays, origin, intensities, ctx=None, details=None):
    """Closest hit of every ray with the mesh; returns (hit points M x 3 float64, intensities
    of the rays that hit), ray order preserved.  As in the reference the [origin | direction]
    rows are cast to float32 for the intersection (:251) while the hit points are formed in
    float64 from the float64 directions and the float32 distance (:261-263).
    `details`, if a dict, receives t_hit / primitive_ids / primitive_uvs / valid mask (the
    reference discards them; north_star asks for the triangle indices)."""
    if hasattr(mesh, "has_triangle_normals") and not mesh.has_triangle_normals():  # :240-243, in place
        mesh.compute_triangle_normals()
    if hasattr(mesh, "has_vertex_normals") and not mesh.has_vertex_normals():
        mesh.compute_vertex_normals()
    rays = np.asarray(rays, dtype=np.float64).reshape(-1, 3)
    n = len(rays)
    origins = np.tile(np.asarray(origin), (n, 1))
    rays6 = np.hstack((origins, rays)).astype(np.float32)
    hit = cast_rays(mesh, rays6, ctx)
    t = hit["t_hit"]
    valid = t != np.inf
    points = origins[valid] + rays[valid] * t[valid, np.newaxis]
    if details is not None:
        details.update(hit)
        details["valid"] = valid
    return points, np.asarray(intensities)[valid]


# matplotlib's 'jet' segment data (x, y) per channel, restated so the GPU box needs no
# matplotlib: the reference maps normalised intensities through cm.get_cmap('jet') (:290).
_JET = {
    "r": [(0.0, 0.0), (0.35, 0.0), (0.66, 1.0), (0.89, 1.0), (1.0, 0.5)],
    "g": [(0.0, 0.0), (0.125, 0.0), (0.375, 1.0), (0.64, 1.0), (0.91, 0.0), (1.0, 0.0)],
    "b": [(0.0, 0.5), (0.11, 1.0), (0.34, 1.0), (0.65, 0.0), (1.0, 0.0)],
}
_JET_LUT = None


def _jet_lut():
    global _JET_LUT
    if _JET_LUT is None:
        grid = np.linspace(0.0, 1.0, 256)
        _JET_LUT = np.stack([np.interp(grid, *zip(*_JET[c])) for c in "rgb"], axis=1)
    return _JET_LUT


def jet(values):
    """RGB of matplotlib's 256-entry 'jet' lookup: index floor(v * 256) clipped to 0..255;
    NaN maps to black (matplotlib's 'bad' colour is transparent black)."""
    v = np.asarray(values, dtype=np.float64)
    if v.ndim == 0:                        # a scalar: one colour
        return jet(v.reshape(1))[0]
    x = v * 256.0
    bad = x != x
    if bad.any():
        x[bad] = 0.0
    np.clip(x, 0.0, 255.0, out=x)          # below 0 -> entry 0, 1 and above -> entry 255 (truncation does the floor)
    rgb = np.take(_jet_lut(), x.astype(np.intp), axis=0)
    if bad.any():
        rgb[bad] = 0.0
    return rgb


def create_intersection_pcd(intersections, intensities):
    """Hit points coloured by min-max normalised intensity through 'jet'
    (defect_projection.py:268-294).  Equal intensities divide by zero in the reference too
    (NaN -> black)."""
    pcd = PointCloud.adopt(intersections) if isinstance(intersections, np.ndarray) and intersections.flags.owndata else PointCloud(intersections)
    intensities = np.asarray(intensities, dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        scaled = (intensities - np.min(intensities)) / (np.max(intensities) - np.min(intensities))
    pcd.colors = jet(scaled)
    return pcd


def project_debug_rays(rays, origin):
    """Red 1000-unit line segments along the rays, returned when nothing was hit (:296-317)."""
    logging.info("No intersections found.")
    rays = np.asarray(rays, dtype=np.float64).reshape(-1, 3)
    ls = LineSet()
    ls.points = np.vstack((np.tile(origin, (len(rays), 1)), origin + rays * 1000))
    ls.lines = np.array([[i, i + len(rays)] for i in range(len(rays))], dtype=np.int32).reshape(-1, 2)
    ls.paint_uniform_color([1, 0, 0])
    return ls


def load_extrinsics(file_path):
    """color_to_depth / depth_to_color 4x4 from {file_path}/configs/camera_extrinsics.json
    (defect_projection.py:65-94)."""
    with open(f"{file_path}/configs/camera_extrinsics.json", "r") as fh:
        data = json.load(fh)

    def _mat(entry):
        T = np.eye(4)
        T[:3, :3] = np.array(entry["rotation_matrix"])
        T[:3, 3] = np.array(entry["translation_vector"][0])
        return T

    return _mat(data["color_to_depth"]), _mat(data["depth_to_color"])


def ray_tracing(data_dir, target_mesh, heatmap, color_intrinsics, heatmap_threshold=0.5):
    """Project the heat map's hot pixels onto the posed mesh (defect_projection.py:527-563):
    rays start at the colour camera's origin, the mesh (given in the depth-camera frame) is
    copied and moved into the colour-camera frame with inv(color_to_depth), and the closest
    hits come back as a jet-coloured cloud -- or red debug rays if nothing was hit.
    Returns (cloud or LineSet, moved mesh copy).

    Pixel selection, ray generation, the sweep and the hit filter run as one device call
    (pedp_project_heatmap); the step-by-step functions above give the same arrays."""
    origin = np.array([0, 0, 0])
    color_to_depth, _ = load_extrinsics(data_dir)
    mesh_in_color = clone(target_mesh)
    mesh_in_color.transform(np.linalg.inv(color_to_depth))
    # side effect of intersect_rays_with_mesh on the returned copy (defect_projection.py:240-243)
    if hasattr(mesh_in_color, "has_triangle_normals") and not mesh_in_color.has_triangle_normals():
        mesh_in_color.compute_triangle_normals()
    if hasattr(mesh_in_color, "has_vertex_normals") and not mesh_in_color.has_vertex_normals():
        mesh_in_color.compute_vertex_normals()
    dev = _device_mesh(mesh_in_color, _lib.default_context())
    out = dev.project_heatmap(heatmap, color_intrinsics.intrinsic_matrix, heatmap_threshold, origin)
    if len(out["points"]) > 0:
        return create_intersection_pcd(out["points"], out["intensities"]), mesh_in_color
    rays, _ = compute_rays(heatmap_to_points(heatmap, heatmap_threshold), color_intrinsics)
    return project_debug_rays(rays, origin), mesh_in_color


class FrameProjector:
    """Camera-rate form of run.py:95-119 / :176-200 for a fixed model: the model-frame mesh is
    uploaded once, every frame only sends a 4x4 pose and the heat map.

        proj = FrameProjector(model_mesh, color_intrinsics, color_to_depth)
        cloud = proj.project(pose_model_to_depth, heatmap, 0.75)

    The pose chain is the reference's: the mesh is moved to the depth-camera frame by the refined
    pose (transform_object, pose_estimation.py:406-409) and then into the colour-camera frame by
    inv(color_to_depth) (defect_projection.py:549-550).  Open3D applies the two transforms one
    after the other on float64 vertices; here their float64 product is applied once on the device,
    which moves a vertex by at most a few ulp before the float32 cast."""

    def __init__(self, model_mesh, color_intrinsics, color_to_depth=None, ctx=None):
        self.ctx = ctx or _lib.default_context()
        self.K = np.array(color_intrinsics.intrinsic_matrix, dtype=np.float64)
        self.depth_to_color = np.eye(4) if color_to_depth is None else np.linalg.inv(np.asarray(color_to_depth, np.float64))
        self.mesh = _lib.Mesh(self.ctx, np.asarray(model_mesh.vertices, np.float64), np.asarray(model_mesh.triangles),
                              posable=True)

    def project(self, pose, heatmap, heatmap_threshold=0.5, details=None, into=None):
        """pose: model -> depth-camera 4x4.  heatmap: H x W float64 / float32, a numpy array or a CUDA tensor (a
        detector's output, or a map kept on the device between detections: no upload).  into (4 x 4, optional): the hit
        cloud is moved by it on the device -- run.py:118 / :200 `cloud.transform(reader.color_to_depth)` in the same
        call.  Returns the jet-coloured hit cloud (colours from the device, create_intersection_pcd's arithmetic), or
        None when no ray hits; `details` (dict) receives pixels / primitive_ids / n_rays."""
        self.mesh.set_pose(self.depth_to_color @ np.asarray(pose, dtype=np.float64))
        out = self.mesh.project_heatmap(heatmap, self.K, heatmap_threshold, jet_lut=_jet_lut(), post=into)
        if details is not None:
            details.update(out)
        if len(out["points"]) == 0:
            return None
        return PointCloud.adopt(out["points"], colors=out["colors"])

    def posed_mesh(self, pose, template):
        """transform_object(reader.target_mesh, pose) (run.py:109-110, :179-181) from the resident model: the float64
        posed vertices come from the device (pedp_mesh_posed_vertices: the arithmetic pedp_mesh_set_pose starts from),
        the triangles are the template's (read-only, shared: the copy the viewer gets is never written to)."""
        from .geometry import TriangleMesh

        tris = np.asarray(template.triangles)
        tris = tris.view()
        tris.setflags(write=False)
        mesh = TriangleMesh(self.mesh.posed_vertices(pose))
        mesh.triangles = tris
        if len(template.vertex_normals) or len(template.triangle_normals):
            from .geometry import rigid
            mesh.vertex_normals = rigid(template.vertex_normals, pose, rotate_only=True) if len(template.vertex_normals) else np.zeros((0, 3))
            mesh.triangle_normals = rigid(template.triangle_normals, pose, rotate_only=True) if len(template.triangle_normals) else np.zeros((0, 3))
        return mesh


# ---------------------------------------------------------------- depth-based projection
# The reference's alternative to ray casting (defect_projection.py:359-460): back-project the hot
# pixels through the depth image and snap them onto the model cloud.  The snap is one exact
# nearest-neighbour pass of the ICP stage (pedp_nn).

def heatmap_to_point3d(heatmap, depth_image, intrinsic, threshold=0.1):
    """Rows [x, y, z, intensity] of the pixels whose max-normalised heat exceeds `threshold` and
    whose depth is positive, row-major like the reference's double loop (:359-397):
    x = (u - cx) * depth / fx, y = (v - cy) * depth / fy, z = 0.98 * depth."""
    heat = np.asarray(heatmap)
    depth = np.asarray(depth_image)
    K = np.asarray(intrinsic.intrinsic_matrix, dtype=np.float64)
    h = min(heat.shape[0], depth.shape[0])
    w = min(heat.shape[1], depth.shape[1])
    inten = heat[:h, :w] / np.max(heat)
    d = depth[:h, :w]
    vs, us = np.nonzero((inten > threshold) & (d > 0))
    dz = d[vs, us]
    x3d = (us - K[0, 2]) * dz / K[0, 0]
    y3d = (vs - K[1, 2]) * dz / K[1, 1]
    if len(vs) == 0:
        return np.array([])
    return np.stack([x3d, y3d, dz * 0.98, inten[vs, us]], axis=1)


def pcd_from_point3d(points_3D):
    if len(points_3D) == 0:
        raise ValueError("No valid 3D points found.")
    return PointCloud(np.array(points_3D)[:, :3])


def calc_coordinates(depth_image, points, intrinsic):
    """3-D coordinates of the given (x, y) pixels from the depth image; zero depth is skipped
    (:462-494)."""
    K = np.asarray(intrinsic.intrinsic_matrix, dtype=np.float64)
    out = []
    for x, y in points:
        depth = depth_image[y, x]
        if depth == 0:
            logging.info(f"Depth is zero at coordinates x = {x}, y = {y}. Skipping this point.")
            continue
        out.append([(x - K[0, 2]) * depth / K[0, 0], (y - K[1, 2]) * depth / K[1, 1], depth])
    return np.array(out, dtype=np.float64)


def align_to_surface(defect_points, target_pcd, offset=0.1, ctx=None):
    """Snap every defect point [x, y, z, ...] to its nearest model point and lift it by `offset`
    along that point's normal (:413-460).  Returns (offset_points, aligned_points).  Like the
    reference (:428-433) a model without normals gets them estimated first, in place, with
    KDTreeSearchParamHybrid(radius=0.1, max_nn=30) (pedp_estimate_normals)."""
    from .geometry import as_holder, normals_of, points_of

    if len(defect_points) == 0:
        return np.array([]), np.array([])
    pts = np.asarray(defect_points, dtype=np.float64).reshape(len(defect_points), -1)
    model, normals = points_of(target_pcd), normals_of(target_pcd)
    if normals is None:
        if isinstance(target_pcd, PointCloud):
            target_pcd.estimate_normals(search_param=KDTreeSearchParamHybrid(radius=0.1, max_nn=30))   # on the GPU, in place
            normals = normals_of(target_pcd)
        else:   # an Open3D cloud: normals on the GPU from its points, written back like the reference's in-place call
            holder = as_holder(target_pcd)
            holder.estimate_normals(search_param=KDTreeSearchParamHybrid(radius=0.1, max_nn=30))
            normals = np.asarray(holder.normals)
            try:
                import open3d as o3d

                target_pcd.normals = o3d.utility.Vector3dVector(normals)
            except ImportError:
                pass
    ctx = ctx or _lib.default_context()
    idx, _ = _lib.nn(ctx, _lib.Cloud(ctx, pts[:, :3]), _lib.Cloud(ctx, model))
    aligned = model[idx]
    return aligned + normals[idx] * offset, aligned
