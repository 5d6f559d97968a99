"""Deterministic synthetic inputs of the benchmark and parity configurations
(SURVEY.md s8d / BASELINE.md s3): the "bumpy torus" mesh, the Azure-Kinect NFOV
unbinned camera (640x576, datareader.py:278), the ground-truth pose, one ray per
pixel, and the scene cloud back-projected from a rendered depth frame.

Pure numpy; no GPU, no oracle.  The depth frame itself is rendered by whoever calls
`scene_from_depth` (the HIP ray caster in bench.py / smoke, the oracle in CPU tests).
"""
import numpy as np

CONFIGS = {
    # name: (W, H, image width, image height, fx=fy, cx, cy)
    "bench_100k": (250, 200, 640, 576, 504.0, 319.5, 287.5),
    "bench_1m": (1000, 500, 1280, 720, 610.0, 639.5, 359.5),
    "parity": (50, 40, 160, 144, 126.0, 79.5, 71.5),
    "tiny": (16, 12, 48, 40, 38.0, 23.5, 19.5),
}


def bumpy_torus(W, H):
    """Vertices (W*H x 3, float32 values stored as float32), triangles (2*W*H x 3 uint32),
    vertex normals (float64, area-weighted face normals, normalised)."""
    i = np.arange(W)[:, None]
    j = np.arange(H)[None, :]
    u = 2.0 * np.pi * i / W
    v = 2.0 * np.pi * j / H
    r = 25.0 * (1.0 + 0.15 * np.sin(5.0 * u) * np.cos(3.0 * v))
    R = 60.0
    x = (R + r * np.cos(v)) * np.cos(u)
    y = (R + r * np.cos(v)) * np.sin(u)
    z = r * np.sin(v) + 0.0 * u
    verts = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(np.float32)

    def vid(a, b):
        return ((a % W) * H + (b % H)).astype(np.uint32)

    ii, jj = np.meshgrid(np.arange(W), np.arange(H), indexing="ij")
    a, b, c, d = vid(ii, jj), vid(ii + 1, jj), vid(ii + 1, jj + 1), vid(ii, jj + 1)
    tris = np.stack([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], axis=2).reshape(-1, 3)
    tris = np.ascontiguousarray(tris, dtype=np.uint32)

    vd = verts.astype(np.float64)
    fn = np.cross(vd[tris[:, 1]] - vd[tris[:, 0]], vd[tris[:, 2]] - vd[tris[:, 0]])  # 2*area*n
    normals = np.zeros_like(vd)
    for k in range(3):
        np.add.at(normals, tris[:, k], fn)
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    return verts, tris, normals


def rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float64)


def axis_angle(axis, angle):
    k = np.asarray(axis, dtype=np.float64)
    k = k / np.linalg.norm(k)
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1.0 - np.cos(angle)) * (K @ K)


def gt_pose():
    """model -> camera, millimetres."""
    T = np.eye(4)
    T[:3, :3] = rot_x(0.4) @ rot_y(-0.7) @ rot_z(0.25)
    T[:3, 3] = [0.0, 0.0, 350.0]
    return T


def start_pose():
    """GT perturbed by 3 degrees about (1,1,0)/sqrt(2) and (+2,-1.5,+3) mm (model -> camera)."""
    D = np.eye(4)
    D[:3, :3] = axis_angle([1.0, 1.0, 0.0], np.deg2rad(3.0))
    D[:3, 3] = [2.0, -1.5, 3.0]
    return D @ gt_pose()


def batched_start_poses(n=256, seed=1):
    """n start poses: start_pose() composed with rotations of <= 5 degrees (default_rng(seed))."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, 4, 4))
    base = start_pose()
    for b in range(n):
        axis = rng.normal(size=3)
        ang = np.deg2rad(5.0) * rng.uniform(0.0, 1.0)
        D = np.eye(4)
        D[:3, :3] = axis_angle(axis, ang)
        out[b] = D @ base
    return out


def pixel_rays(width, height, f, cx, cy, mask=None):
    """Unit directions of one ray per pixel, row-major (y outer) like heatmap_to_points /
    compute_rays (src/defect_projection.py:176-178, :216-222): float64."""
    ys, xs = np.mgrid[0:height, 0:width]
    if mask is not None:
        ys, xs = ys[mask], xs[mask]
    xn = (xs.reshape(-1) - cx) / f
    yn = (ys.reshape(-1) - cy) / f
    d = np.stack([xn, yn, np.ones_like(xn)], axis=1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return d


def rays6_from_dirs(dirs, origin=(0.0, 0.0, 0.0)):
    """[o|d] rows cast to float32 exactly like intersect_rays_with_mesh builds its tensor
    (src/defect_projection.py:247-251)."""
    o = np.tile(np.asarray(origin, dtype=np.float64), (len(dirs), 1))
    return np.ascontiguousarray(np.hstack([o, dirs]).astype(np.float32))


def posed_vertices(verts, T):
    """TriangleMesh.transform in float64, then the float32 cast of from_legacy."""
    v = verts.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    return np.ascontiguousarray(v.astype(np.float32))


def scene_from_depth(t_hit, dirs, noise_sigma=0.5, back_plane=600.0, seed=0):
    """Scene cloud of the frame: z = t_hit * d_z (back plane where the ray missed), plus
    N(0, sigma) depth noise from default_rng(seed), back-projected through every pixel."""
    t = np.asarray(t_hit, dtype=np.float64)
    dz = dirs[:, 2]
    z = np.where(np.isfinite(t), t * dz, back_plane)
    rng = np.random.default_rng(seed)
    z = z + rng.normal(0.0, noise_sigma, size=z.shape)
    pts = dirs / dz[:, None] * z[:, None]
    return np.ascontiguousarray(pts)


def depth_image(height, width, seed=0, nan=True):
    """Sensor-like depth in metres (the unit of the reference's depth filters, zfar = 100,
    thresholds 1 mm): a slanted wavy surface with a step edge, 1 mm noise, holes (0), sub-
    threshold values, readings beyond zfar and, optionally, a few NaN."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    d = 0.6 + 0.0008 * x + 0.0004 * y + 0.01 * np.sin(x / 7.0) * np.cos(y / 5.0)
    d[:, width // 2:] += 0.15                                   # step edge
    d += rng.normal(0, 0.001, d.shape)
    u = rng.uniform(size=d.shape)
    d[u < 0.05] = 0.0                                           # holes
    d[(u >= 0.05) & (u < 0.06)] = 0.0005                        # below the validity threshold
    d[(u >= 0.06) & (u < 0.07)] = 150.0                         # beyond zfar
    blob = (x - width * 0.3) ** 2 + (y - height * 0.6) ** 2 < (min(height, width) * 0.12) ** 2
    d[blob] = 0.0                                               # a larger hole
    if nan:
        d[(u >= 0.07) & (u < 0.072)] = np.nan
    return d.astype(np.float32)


class Frame:
    """Everything one benchmark / parity frame needs, built lazily around a ray caster."""

    def __init__(self, config="bench_100k"):
        self.config = config
        W, H, self.width, self.height, self.f, self.cx, self.cy = CONFIGS[config]
        self.verts, self.tris, self.normals = bumpy_torus(W, H)
        self.T_gt = gt_pose()
        self.T_start = start_pose()
        self.dirs = pixel_rays(self.width, self.height, self.f, self.cx, self.cy)
        self.rays6 = rays6_from_dirs(self.dirs)
        self.verts_posed = posed_vertices(self.verts, self.T_gt)
        self.model_points = self.verts.astype(np.float64)
        self.max_correspondence_distance = 10.0

    @property
    def K(self):
        return np.array([[self.f, 0.0, self.cx], [0.0, self.f, self.cy], [0.0, 0.0, 1.0]])

    @property
    def n_rays(self):
        return len(self.rays6)

    @property
    def n_tris(self):
        return len(self.tris)

    def scene(self, t_hit):
        return scene_from_depth(t_hit, self.dirs)

    def icp_init(self):
        """registration_icp's init: source (scene, camera frame) -> target (model frame)."""
        return np.linalg.inv(self.T_start)
