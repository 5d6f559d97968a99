/*
 * pedp.h -- C ABI of libpedp_hip.so: the MI355X (gfx950) implementation of the
 * reference's geometric hot path, behind plain pointers and sizes.
 *
 * The reference binds no FFI on this path today: both stages are Python calls into
 * the Open3D wheel.  Each entry point below names the reference call it replaces
 * (paths relative to the reference repository); INTEGRATION.md shows the ctypes
 * stub a maintainer adds on the reference side.  The library occupies the
 * reference's native-extension slot (import-with-fallback convention of
 * Utils.py:41-57, where mycpp and bundlesdf.mycuda are loaded) and additionally
 * exports cluster_poses so estimater.py:118 keeps working without mycpp.
 *
 * Conventions
 *   - every function returns PEDP_OK (0) or a negative pedp_status; the message of
 *     the last failure on the calling thread is pedp_last_error().
 *   - `mem` arguments: PEDP_HOST = the array pointers are host memory (the call
 *     copies, runs, copies back and returns after completion); PEDP_DEVICE = they
 *     are device memory on the context's GPU (the call only enqueues work on the
 *     context's stream; use pedp_ctx_synchronize or the stream to wait).
 *   - the callee never frees or retains caller arrays; handles own device copies.
 *   - one context = one device + one HIP stream; a context is not thread-safe,
 *     different contexts are independent.
 */
#ifndef PEDP_H
#define PEDP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PEDP_OK = 0,
    PEDP_ERR_BAD_ARG = -1,
    PEDP_ERR_NO_NORMALS = -2, /* point-to-plane needs target normals (Open3D raises) */
    PEDP_ERR_HIP = -3,
    PEDP_ERR_ALLOC = -4,
    PEDP_ERR_COLLECTIVE = -5 /* the caller's all-reduce hook or an RCCL call failed */
} pedp_status;

enum { PEDP_HOST = 0, PEDP_DEVICE = 1 };
enum { PEDP_POINT_TO_PLANE = 0, PEDP_POINT_TO_POINT = 1 };

typedef struct pedp_ctx_s *pedp_ctx_t;
typedef struct pedp_mesh_s *pedp_mesh_t;
typedef struct pedp_rayset_s *pedp_rayset_t;
typedef struct pedp_cloud_s *pedp_cloud_t;

int pedp_version(void);
const char *pedp_last_error(void);
int pedp_device_count(int *count);

/* stream: a hipStream_t the caller owns (e.g. torch.cuda.current_stream().cuda_stream),
 * or NULL to let the context create its own. */
int pedp_ctx_create(int device, void *stream, pedp_ctx_t *out);
void pedp_ctx_destroy(pedp_ctx_t ctx);
int pedp_ctx_synchronize(pedp_ctx_t ctx);

/* ---------------------------------------------------------------- ray projection
 * Replaces src/defect_projection.py:245-256:
 *     t_mesh = o3d.t.geometry.TriangleMesh.from_legacy(mesh)        -> pedp_mesh_create
 *     scene = o3d.t.geometry.RaycastingScene(); scene.add_triangles(t_mesh)
 *     intersections = scene.cast_rays(o3d_rays)                     -> pedp_raycast
 * verts: V x 3 float32 (from_legacy's cast), tris: F x 3 uint32, both host memory.
 */
int pedp_mesh_create(pedp_ctx_t ctx, const float *verts, int64_t V, const uint32_t *tris,
                     int64_t F, pedp_mesh_t *out);
void pedp_mesh_destroy(pedp_mesh_t mesh);
int pedp_mesh_size(pedp_mesh_t mesh, int64_t *V, int64_t *F);

/* cast_rays: rays6 = N x [ox oy oz dx dy dz] float32 (directions are used as given,
 * not re-normalised; tnear = 0, tfar = +inf, no culling).  Outputs, all length N:
 * t_hit (+inf on miss), prim_id (0xFFFFFFFF on miss), uv (N x 2, nullable; 0 on miss).
 * Closest hit; ties: smaller t, then smaller triangle index. */
int pedp_raycast(pedp_ctx_t ctx, pedp_mesh_t mesh, const float *rays6, int64_t N, int mem,
                 float *t_hit, uint32_t *prim_id, float *uv);

/* Tuning knobs of the sweep (0 = keep default): triangle chunks per ray block
 * (multiple of 8: chunk c is served by XCD c % 8) and sweep variant:
 *   0 auto (2 below 16,384 rays, else 4; 3 for a ray count whose last variant-4 cast had to be
 *     completed by the exhaustive sweep)
 *   1 every ray against every triangle: rays of one origin on the matrix pipe (bf16 MFMA filter with a proven
 *     slack, the exact test on the pairs that pass it), other rays on the packed fp32 loop of variant 5
 *   2 triangle-per-lane with a wavefront-wide min-t reduction
 *   3 ray-per-lane with conservative cluster culling when all rays share one origin
 *     (falls back to 1 on the device otherwise)
 *   4 triangle-driven: rays of one origin threaded into a grid of directions, every triangle
 *     tested against the rays in the cells its image covers (completed by 1 on the device when
 *     the rays do not share an origin, leave the grid's half space or crowd a cell).
 *   5 every ray against every triangle on the vector pipe (ray per lane, packed fp32; round 3's exhaustive kernel)
 * All variants return identical bits. */
int pedp_raycast_configure(pedp_ctx_t ctx, int tri_chunks, int variant);
/* A camera's rays as a resident object.  A camera sends the same rays every frame (the reference generates them from the
 * pixel grid in the camera's frame and moves the MESH, src/defect_projection.py:196-223, :549-550): a ray set copies
 * the N x 6 float32 [origin | direction] rows once and builds, once, what the triangle-driven ray stage (variant 4)
 * derives from the rays alone -- the frame, the grid over the mapped directions, the cells' chains.  A cast against a
 * ray set is then the triangle kernels and the result kernel only: four launches, the rays never re-read
 * (pedp_raycast reads them twice and rebuilds the chains per call).  Results are those of pedp_raycast, bit for bit;
 * rays the grid cannot serve (origins that differ, directions outside the frame's half space, crowded cells) are cast
 * by pedp_raycast's other variants from the resident copy.  mem: where rays6 / the outputs live. */
int pedp_rayset_create(pedp_ctx_t ctx, const float *rays6, int64_t N, int mem, pedp_rayset_t *out);
void pedp_rayset_destroy(pedp_rayset_t rays);
int pedp_raycast_rayset(pedp_ctx_t ctx, pedp_mesh_t mesh, pedp_rayset_t rays, int mem, float *t_hit, uint32_t *prim_id, float *uv);
/* how the last cast against the set was answered: *variant 4 (the grid) with *grid_status 0, or the variant
 * pedp_raycast fell back to with the reason the grid gave (bits as in pedp_raycast_last_variant) */
int pedp_rayset_last_variant(pedp_rayset_t rays, int *variant, int *grid_status);

/* Diagnostics of variant 1's matrix-pipe filter (tests): for N host rays of ONE origin, the filter's score of every (ray,
 * triangle) pair -- >= 0: the pair goes to the exact test -- and the slack inside that score; N x F float32 each, host. */
int pedp_debug_mfma_scores(pedp_ctx_t ctx, pedp_mesh_t mesh, const float *rays6, int64_t N, float *score, float *slack);

/* Which variant the last pedp_raycast of this context ran and, for variant 4, whether the grid
 * answered the cast (grid_status 0) or the exhaustive sweep had to (bit 0 origins differ, 1 a ray
 * outside the half space, 2 a crowded cell, 3 item table full).  Synchronises the stream. */
int pedp_raycast_last_variant(pedp_ctx_t ctx, int *variant, int *grid_status);
/* Diagnostics of the triangle-driven ray stage, for the tests that MEASURE its margin: runs a variant-4 cast of the N
 * host rays; tri_out = F x 12 floats per triangle [live, every-cell, fx0, fx1, fy0, fy1: the bounding rectangle of its
 * mapped corners in cell units before widening, mx, my: the margins, x0, x1, y0, y1: the cells visited]; ray_out = N x 3
 * [kind (1 = mapped), continuous cell coordinate x, y]; grid = [GX, GY, grid_status]. */
int pedp_debug_rast_rects(pedp_ctx_t ctx, pedp_mesh_t mesh, const float *rays6, int64_t N, float *tri_out, float *ray_out,
                          int *grid);

/* Milliseconds the last pedp_raycast spent in its sweep stage -- for variant 4 the bounds, chain, triangle
 * and tile kernels, for variant 3 the direction binning, cull masks, segment table and the sweep itself (HIP events on the context's stream,
 * around everything between the operand set-up and the final t / id / uv write-out);
 * synchronises on the second event. */
int pedp_raycast_last_sweep_ms(pedp_ctx_t ctx, float *ms);

/* ---------------------------------------------------------------- fused defect projection
 * the per-frame work around cast_rays moved onto the device.
 *
 * Posable mesh.  Replaces the per-frame `copy.deepcopy(mesh)` + `mesh.transform(T)` +
 * `TriangleMesh.from_legacy(mesh)` chain (src/pose_estimation.py:406-409,
 * src/defect_projection.py:549-550, :245; poses composed in run.py:95-119): the model-frame
 * float64 vertices stay resident, pedp_mesh_set_pose forms T * (x, y, z, 1) in float64 (fixed
 * operation order), divides by the fourth component, casts to float32 and rebuilds the triangle
 * records and culling spheres on the device.  T: row-major 4x4 float64 on the host, NULL =
 * identity.  set_pose only enqueues work on the context's stream. */
int pedp_mesh_create_posable(pedp_ctx_t ctx, const double *verts, int64_t V, const uint32_t *tris,
                             int64_t F, pedp_mesh_t *out);
int pedp_mesh_set_pose(pedp_mesh_t mesh, const double T[16]);
/* transform_object(reader.target_mesh, T) (src/pose_estimation.py:406-409; run.py:109-110, :179-181) for the viewer's
 * copy of the mesh: the V x 3 float64 vertices T * (x, y, z, 1) / w of a posable mesh's model, the arithmetic
 * pedp_mesh_set_pose starts from, before the float32 cast.  The mesh's own pose is not changed.  out: host
 * (returns after completion) or device memory. */
int pedp_mesh_posed_vertices(pedp_mesh_t mesh, const double T[16], int mem, double *out);

/* Pinhole intrinsics as read from PinholeCameraIntrinsic.intrinsic_matrix
 * (src/defect_projection.py:209-212) plus the heat map's size. */
typedef struct {
    double fx, fy, cx, cy;
    int32_t width, height;
} pedp_pinhole;

/* heatmap_to_points + compute_rays + intersect_rays_with_mesh in one call
 * (src/defect_projection.py:165-179, :196-223, :225-266):
 *   select pixels with heatmap > threshold in row-major order (np.where); per pixel
 *   d = (xn, yn, 1) / sqrt((xn*xn + yn*yn) + 1), xn = (x - cx) / fx, yn = (y - cy) / fy in float64;
 *   [origin | d] cast to float32 and cast against the mesh (same sweep as pedp_raycast);
 *   rays with t_hit != inf are kept in order; point = origin + d * t_hit in float64.
 * heatmap: height x width float64, row-major.  origin: 3 float64 on the host (the reference passes
 * zeros).  Outputs hold `capacity` rows: points capacity x 3 f64, intensities f64, pixels
 * capacity x 2 int32 (x, y; nullable), prim_id u32 (nullable).  n_rays / n_hits are host
 * scalars (the call synchronises the stream).  PEDP_DEVICE: heatmap and the output arrays are
 * device memory.  If n_hits > capacity nothing is written, n_hits is still reported and the call
 * returns PEDP_ERR_BAD_ARG; capacity = width * height always suffices. */
int pedp_project_heatmap(pedp_ctx_t ctx, pedp_mesh_t mesh, const pedp_pinhole *cam, const double *heatmap,
                         double threshold, const double origin[3], int mem, int64_t capacity,
                         double *points, double *intensities, int32_t *pixels, uint32_t *prim_id,
                         int64_t *n_rays, int64_t *n_hits);

/* The same with what the reference does to the hits right after (src/defect_projection.py:268-294
 * create_intersection_pcd; run.py:118, :200 `.transform(reader.color_to_depth)`), and with the heat map where and as
 * it is.  opts (NULL = the call above):
 *   heat_f32   the heat map is float32 (upcast exactly; every comparison and the intensities in float64)
 *   heat_mem   PEDP_HOST / PEDP_DEVICE for the heat map alone (a detector's output already on the GPU, or a map
 *              kept resident between detections, run.py:66, :147), whatever `mem` says about the outputs
 *   jet_lut    256 x 3 float64 on the host: matplotlib's `jet` lookup table (pedp_hip.ray_projection builds it);
 *   colors     capacity x 3 float64 (where `mem` says): jet_lut[clip(trunc(s * 256), 0, 255)] of the min-max
 *              normalised hit intensities s = (I - min I) / (max I - min I), float64, one rounding per operation;
 *              equal intensities divide by zero like the reference (NaN -> black).  Both NULL: no colours.
 *   post       4x4 float64 on the host (nullable): the hit points are moved by it -- ((T0 x + T1 y) + T2 z) + T3 per
 *              row, pedp_transform_points' order -- before they are written. */
typedef struct {
    int32_t heat_f32;
    int32_t heat_mem;
    const double *jet_lut;
    double *colors;
    const double *post;
} pedp_project_opts;
int pedp_project_heatmap_ex(pedp_ctx_t ctx, pedp_mesh_t mesh, const pedp_pinhole *cam, const void *heatmap,
                            double threshold, const double origin[3], int mem, int64_t capacity,
                            double *points, double *intensities, int32_t *pixels, uint32_t *prim_id,
                            int64_t *n_rays, int64_t *n_hits, const pedp_project_opts *opts);

/* ---------------------------------------------------------------- depth pre-filters
 * the reference's warp-lang kernels (CUDA-only JIT) and its depth back-projection.
 * Images are H x W float32, row-major; PEDP_HOST copies in and out and returns after completion,
 * PEDP_DEVICE only enqueues on the context's stream.  Defaults of the reference in brackets.
 *
 * erode_depth (Utils.py:356-396; estimater.py:171, :255): a pixel becomes 0 when more than
 * ratio_thres [0.8] of its (2 radius + 1)^2 window [radius 2] is invalid (< 0.001 or >= zfar
 * [100]) or differs from the centre by more than depth_diff_thres [0.001]; else it is kept. */
int pedp_erode_depth(pedp_ctx_t ctx, const float *depth, int H, int W, int radius, float depth_diff_thres,
                     float ratio_thres, float zfar, int mem, float *out);
/* bilateral_filter_depth (Utils.py:304-357; estimater.py:172, :256): mean of the valid window
 * pixels, then a Gaussian (sigmaD [2] in pixels, sigmaR [100000] in depth) weighted average over
 * the valid pixels within 0.01 of that mean; float32, window walked column by column. */
int pedp_bilateral_filter_depth(pedp_ctx_t ctx, const float *depth, int H, int W, int radius, float zfar,
                                float sigmaD, float sigmaR, int mem, float *out);
/* depth2xyzmap (Utils.py:401-420; run.py:89, estimater.py:175, :212), uvs=None form: xyz is
 * H x W x 3 float32, ((u - cx) z / fx, (v - cy) z / fy, z) formed in float64; depth < 0.001 -> 0.
 * K: row-major 3 x 3 float64 on the host. */
int pedp_depth2xyzmap(pedp_ctx_t ctx, const float *depth, int H, int W, const double K[9], int mem, float *xyz);
/* depth2xyzmap_batch (Utils.py:423-442; estimater.py:259): B images, Ks = B x 9 float32 on the
 * host, float32 arithmetic, invalid = depth < 0.001 or depth > zfar. */
int pedp_depth2xyzmap_batch(pedp_ctx_t ctx, const float *depths, int B, int H, int W, const float *Ks, float zfar,
                            int mem, float *xyz);

/* The depth entry of a camera frame in ONE call (estimater.py:255-259 and the scene cloud run.py's loop works on):
 *     erode_depth -> bilateral_filter_depth -> depth2xyzmap_batch (one image, float32) -> the pixels with z >= z_min in
 *     row-major order as float64 points scaled by `scale` (metres -> millimetres: 1000)
 * -- the kernels of the three calls above back to back on the context's stream, one upload of the image (depth_mem =
 * PEDP_HOST), no trip to the host between them, one 4-byte read-back (the count).  d_filtered (H x W float32) and d_xyz
 * (H x W x 3 float32) are DEVICE memory, nullable (library scratch is used instead); d_points: DEVICE memory for
 * H x W x 3 float64, the first n_points rows are written (what torch's `xyz[xyz[..., 2] >= z_min].double() * scale`
 * gives).  Results are those of the single calls bit for bit. */
typedef struct {
    int32_t erode_radius;
    float erode_diff, erode_ratio, erode_zfar;      /* erode_depth(radius, depth_diff_thres, ratio_thres, zfar) */
    int32_t bilateral_radius;
    float bilateral_zfar, sigmaD, sigmaR;            /* bilateral_filter_depth(radius, zfar, sigmaD, sigmaR) */
    float K[9];                                      /* intrinsics of depth2xyzmap_batch (float32, row-major) */
    float xyz_zfar;                                  /* its zfar (inf: none) */
    float z_min;                                     /* validity of a back-projected point: 0.001 */
    double scale;
} pedp_depth_entry_params;
int pedp_depth_to_scene(pedp_ctx_t ctx, const float *depth, int H, int W, int depth_mem, const pedp_depth_entry_params *prm,
                        float *d_filtered, float *d_xyz, double *d_points, int64_t *n_points);

/* ---------------------------------------------------------------- point-cloud operations
 * the Open3D calls preprocess_source chains (src/pose_estimation.py:186-268),
 * restated from the published open3d==0.18.0 algorithms.  Host arrays only (N x 3 float64); the
 * calls return after completion.  Where Open3D's result depends on something that cannot be
 * recovered (hash-map order, random_device seed, thread interleaving) the rule used is stated.
 *
 * voxel_down_sample (:204-205): voxel index floor((p - (min_bound - voxel/2)) / voxel) per axis;
 * a voxel's points (and normals, nullable) are summed in point order and divided by the count.
 * Output order: ascending (ix, iy, iz) (Open3D: unordered_map order).  capacity = N always fits. */
int pedp_voxel_down_sample(pedp_ctx_t ctx, const double *pts, const double *normals, int64_t N, double voxel_size,
                           double *out_pts, double *out_normals, int64_t capacity, int64_t *n_out);

/* The same grid over points that are already on the device (N x 3 float64 device memory, e.g. the output of
 * pedp_depth2xyzmap_batch turned into millimetres by the caller): the scene of a frame then never visits the
 * host at full resolution.  Voxel averages come back to host memory like above; no normals.  The caller
 * orders its own work on the array before the call (the library reads it on the context's stream). */
int pedp_voxel_down_sample_device_in(pedp_ctx_t ctx, const double *d_pts, int64_t N, double voxel_size, double *out_pts,
                                     int64_t capacity, int64_t *n_out);
/* cluster_dbscan (:284): neighbours are points with d^2 < eps^2 (itself included), core points
 * have >= min_points neighbours; labels as Open3D's breadth-first sweep assigns them: clusters
 * numbered by their smallest core index, a border point takes the smallest id among its core
 * neighbours, noise is -1. */
int pedp_cluster_dbscan(pedp_ctx_t ctx, const double *pts, int64_t N, double eps, int min_points, int32_t *labels);
/* Per-point part of remove_statistical_outlier (:308-312): mean distance to the k nearest
 * neighbours (the point itself is one of them), summed in ascending order; the global mean /
 * deviation / threshold over N doubles is host arithmetic (pedp_hip.cloud_ops). */
int pedp_knn_mean_distance(pedp_ctx_t ctx, const double *pts, int64_t N, int k, double *avg);
/* estimate_normals with KDTreeSearchParamHybrid(radius, max_nn) (:301-306, callers :174, :216, :250-251):
 * neighbours = the max_nn nearest points with d^2 < radius^2 (itself included) in ascending
 * (d^2, index); >= 3 of them give the covariance (nine cumulants) whose smallest eigenvector
 * (Open3D's non-iterative FastEigen3x3) is the normal, otherwise (0, 0, 1).  prior (nullable,
 * N x 3): existing normals; the result is flipped to agree with them, as Open3D does. */
int pedp_estimate_normals(pedp_ctx_t ctx, const double *pts, int64_t N, double radius, int max_nn, const double *prior,
                          double *normals);

/* o3d.pipelines.registration.compute_fpfh_feature(cloud, KDTreeSearchParamHybrid(radius, max_nn))
 * (src/pose_estimation.py:132-137, :175-180, :255-260): out = N x 33 float64 (row i = column i of
 * Open3D's Feature.data).  The cloud's normals are required.  Restated from the published
 * open3d==0.18.0 Feature.cpp (parity unpinned); max_nn <= 128. */
int pedp_fpfh(pedp_ctx_t ctx, const double *pts, const double *normals, int64_t N, double radius, int max_nn, double *out);

/* The correspondences of registration_ransac_based_on_feature_matching (src/pose_estimation.py:482-501):
 * idx[i] = the target feature nearest to source feature i (squared L2 over the 33 components in
 * float64, ties to the lower index; -1 when there is no target). */
int pedp_feature_match(pedp_ctx_t ctx, const double *fs, int64_t Ns, const double *ft, int64_t Nt, int32_t *idx);
/* segment_plane with ransac_n = 3 (:323-329).  Iteration t samples three distinct indices from a
 * counter-based generator of (seed, t); plane through them; inliers |n.p + d| < threshold; the best
 * iteration has the most inliers (earliest on ties); all iterations are evaluated (Open3D:
 * random_device samples, probabilistic early stop).  Returns, like Open3D, the inliers of the best
 * sampled plane (ascending) and the plane refitted to them (GetPlaneFromPoints). */
int pedp_segment_plane(pedp_ctx_t ctx, const double *pts, int64_t N, double distance_threshold, int num_iterations,
                       uint64_t seed, double plane[4], int32_t *inliers, int64_t *n_inliers);

/* preprocess_source of a frame in one call (src/pose_estimation.py:186-268; every branch run.py can take except
 * param['mesh']), the scene staying on the device between the stages:
 *     voxel_down_sample(down_sample) :204-205 -> segment_plane :323-329 -> [i == 0: estimate_normals of the
 *     down-sampled cloud :216] -> select_by_index(inliers, invert=True) :234 -> cluster_dbscan(eps 10, 10) +
 *     largest cluster :270-299 -> remove_statistical_outlier(75, 0.01) :308-312 -> [i == 0: estimate_normals
 *     of the result :250-251, oriented like the ones it carries]
 * The kernels and rules of the single operations above; results are bit for bit those of calling them one
 * after the other.  pts: N x 3 float64, host memory or (pts_on_device != 0) device memory that the caller has
 * ordered before the call.  out_pts / out_normals (normals only when first_frame): capacity x 3 float64 host
 * arrays; capacity = N always fits.  stage_counts (nullable): points after the voxel grid, the plane removal,
 * the cluster selection, the outlier filter.  status: PEDP_PREPROCESS_OK, _NO_CLUSTER (nothing left after the
 * plane, or DBSCAN found noise only -- the reference prints "No valid clusters found." and fails on None),
 * _DEGENERATE (fewer than three points) or _AMBIGUOUS (box cut only: the plane normal is perpendicular to the
 * average normal within rounding, the caller's own arithmetic decides the flip); n_out is 0 for those three.
 *
 * The background cloud run.py passes (run.py:99-101, :154-156) has no entry here: without param['box'] nothing of
 * it is read (:232-236 overwrite the cut), with param['box'] background_removal returns its input (:386-388), and
 * its down-sampled copy and normals (:204, :252) are locals of the reference's function.
 *
 * flags & PEDP_PREPROCESS_BOX: param['box'] -- the plane's inliers stay, the cloud is cut by the half space
 * remove_points_below_plane keeps (:366-378: signed distance <= 0 to the refit plane, flipped towards the average
 * normal by flip_plane_normal_if_needed :342-359); like the reference's, the cut cloud carries no normals, so the
 * final ones (first frame) take Open3D's default orientation.
 *
 * pedp_preprocess_source_ex, report (nullable): what the reference's two INFO lines print (:220, :356) and the box
 * cut uses --
 *     [0..3] the plane segment_plane returns (refit over the best plane's inliers, unflipped)
 *     [4..6] mean normal of the down-sampled cloud on the 10-unit grid, NOT normalised (first frame; else 1, 1, 1)
 *     [7]    box cut only: 1 if the plane was flipped        [8] inliers of the refit   [9] voxels of the 10-unit grid
 * Without a report and without the box flag none of this is computed (nothing can observe it). */
typedef struct pedp_preprocess_params {
    double voxel_size;          /* params['down_sample'] */
    double plane_distance;      /* params['plane_removal']['distance_threshold'] */
    int32_t plane_iterations;   /* params['plane_removal']['num_iterations'] */
    int32_t first_frame;        /* i == 0 */
    uint64_t seed;              /* segment_plane's sampler (see pedp_segment_plane) */
    double normal_radius;       /* estimate_normals: 2 */
    int32_t normal_max_nn;      /*                   5 */
    int32_t cluster_min_points; /* filter_largest_cluster: 10 */
    double cluster_eps;         /*                         10 */
    int32_t outlier_neighbors;  /* remove_statistical_outliers: 75 */
    int32_t flags;              /* PEDP_PREPROCESS_BOX */
    double outlier_std_ratio;   /*                              0.01 */
    double average_normal_voxel; /* compute_average_normal's grid: 10 (0 = that default) */
} pedp_preprocess_params;
#define PEDP_PREPROCESS_BOX 1
#define PEDP_PREPROCESS_OK 0
#define PEDP_PREPROCESS_NO_CLUSTER 1
#define PEDP_PREPROCESS_DEGENERATE 2
#define PEDP_PREPROCESS_AMBIGUOUS 3
#define PEDP_PREPROCESS_REPORT_DOUBLES 12
int pedp_preprocess_source(pedp_ctx_t ctx, const double *pts, int64_t N, int pts_on_device, const pedp_preprocess_params *prm,
                           double *out_pts, double *out_normals, int64_t capacity, int64_t *n_out, int64_t stage_counts[4],
                           int *status);
int pedp_preprocess_source_ex(pedp_ctx_t ctx, const double *pts, int64_t N, int pts_on_device, const pedp_preprocess_params *prm,
                              double *out_pts, double *out_normals, int64_t capacity, int64_t *n_out, int64_t stage_counts[4],
                              int *status, double report[PEDP_PREPROCESS_REPORT_DOUBLES]);

/* ---------------------------------------------------------------- ICP
 * Replaces src/pose_estimation.py:519-521 and :654-660:
 *     o3d.pipelines.registration.registration_icp(source, target, max_corr_dist, init,
 *         TransformationEstimationPointToPlane()[, ICPConvergenceCriteria(max_iteration=1)])
 * Clouds: N x 3 float64 host arrays (Open3D's Vector3dVector); normals nullable.
 */
int pedp_cloud_create(pedp_ctx_t ctx, const double *pts, const double *normals, int64_t N,
                      pedp_cloud_t *out);
/* The same from DEVICE memory (N x 3 float64 on the context's GPU, e.g. a torch tensor built from
 * depth2xyzmap's output): device-to-device copies on the context's stream, the bounding box by a reduction on the
 * device behind them (no read-back; centroid and magnitudes only when the cloud is first used as a target).
 * The caller orders its own work on the arrays BEFORE the call (the library reads them on the context's stream);
 * the call returns when the copies are complete -- the arrays may then be freed or overwritten at once -- while the
 * box reduction may still be running. */
int pedp_cloud_create_device(pedp_ctx_t ctx, const double *d_pts, const double *d_normals, int64_t N,
                             pedp_cloud_t *out);
void pedp_cloud_destroy(pedp_cloud_t cloud);
int pedp_cloud_size(pedp_cloud_t cloud, int64_t *N, int *has_normals);

/* Optional hook called once per correspondence pass with a DEVICE pointer to the
 * n-double partial-sum packet of this rank; it must enqueue an in-place sum over all
 * ranks on `stream` (RCCL all-reduce through torch.distributed) and return 0. */
typedef int (*pedp_allreduce_fn)(void *user, double *dev_packet, int n, void *stream);

typedef struct {
    double max_correspondence_distance;
    int estimator;           /* PEDP_POINT_TO_PLANE (reference default) / PEDP_POINT_TO_POINT */
    int max_iteration;       /* Open3D default 30 */
    double relative_fitness; /* Open3D default 1e-6; negative disables the early exit */
    double relative_rmse;    /* Open3D default 1e-6 */
    pedp_allreduce_fn allreduce; /* NULL on one GPU */
    void *allreduce_user;
    int64_t n_source_global; /* fitness denominator when the scene is sharded; 0 = local N */
    int use_comm;            /* 1: sum the packet over the ranks of the context's own RCCL
                                communicator (pedp_comm_create) on the stream, no host hook */
} pedp_icp_params;

/* init / T_out: row-major 4x4 float64, source -> target (scene -> model), host memory.
 * corr (nullable, host, N_source): target index per source point or -1.
 * trace (nullable, host): (max_iteration + 1) x 18 doubles [fitness, rmse, T] per pass. */
int pedp_icp(pedp_ctx_t ctx, pedp_cloud_t source, pedp_cloud_t target,
             const pedp_icp_params *params, const double init[16], double T_out[16],
             double *fitness, double *inlier_rmse, int32_t *n_iter_done, int32_t *corr,
             double *trace);

/* pedp_icp in two halves.  pedp_icp_begin enqueues every pass of the registration on the context's stream and returns at
 * once (passes after convergence are no-ops on the device: the host does not look at the criteria in between);
 * pedp_icp_end waits for them and hands the result over -- the same bits as pedp_icp's.  What the host does in between
 * (enqueueing a frame's ray stage on ANOTHER context, bench.py's step) overlaps with the registration instead of waiting in
 * front of it.  Between the two calls this context must not be used for anything else (its workspace and its page-locked
 * state block belong to the pending registration): a second pedp_icp_begin, pedp_icp or pedp_icp_batched returns
 * PEDP_ERR_BAD_ARG; destroying the context or either cloud with a registration pending is an error of the caller.  want_trace:
 * pedp_icp_end's `trace` may be non-null.  The reference has no counterpart (registration_icp is one blocking call,
 * src/pose_estimation.py:447-453). */
int pedp_icp_begin(pedp_ctx_t ctx, pedp_cloud_t source, pedp_cloud_t target, const pedp_icp_params *params,
                   const double init[16], int want_trace);
int pedp_icp_end(pedp_ctx_t ctx, double T_out[16], double *fitness, double *inlier_rmse, int32_t *n_iter_done,
                 int32_t *corr, double *trace);

/* Batched refine (FoundationPose hypothesis sizing, estimater.py:104-122): B start
 * poses share one source and one target; no early exit across the batch.  Up to 8 registrations
 * are in flight on internal streams (each replaying one captured hipGraph per pose); the call
 * returns when all are complete.  Results equal B separate pedp_icp calls on the same handles bit
 * for bit (the float64 sums follow the scene's cached spatial order; a different order -- another
 * handle, another region of start poses -- gives the same correspondences and poses equal to
 * rounding). */
int pedp_icp_batched(pedp_ctx_t ctx, pedp_cloud_t source, pedp_cloud_t target,
                     const pedp_icp_params *params, const double *inits /* B x 16 */, int B,
                     double *T_out /* B x 16 */, double *fitness /* B */,
                     double *inlier_rmse /* B */);

/* The same with one parameter struct per start pose (prms[B]): radius and convergence criteria may
 * differ from pose to pose (estimator and max_iteration may not), each registration stops by its
 * own criteria, n_iter_done (nullable, B) receives the iteration counts.  This is the form
 * improve_result's randomised restarts take (src/pose_estimation.py:577-613: every restart has its
 * own distance threshold and Open3D's default criteria). */
int pedp_icp_batched_ex(pedp_ctx_t ctx, pedp_cloud_t source, pedp_cloud_t target,
                        const pedp_icp_params *prms /* B */, const double *inits /* B x 16 */, int B,
                        double *T_out /* B x 16 */, double *fitness /* B */, double *inlier_rmse /* B */,
                        int32_t *n_iter_done /* B, nullable */);

/* One exact nearest-neighbour pass (the correspondence step alone): for every source
 * point transformed by T, the index of the closest target point and the squared
 * distance, float64-exact (ties: lowest index).  idx/d2: host arrays, length N. */
int pedp_nn(pedp_ctx_t ctx, pedp_cloud_t source, pedp_cloud_t target, const double T[16],
            int32_t *idx, double *d2);
/* Milliseconds of the last TIMED MFMA sweep kernel on this context (HIP events on the stream,
 * directly around the kernel): the sweep of pedp_nn, or of the pass selected by pedp_icp_configure. */
int pedp_nn_last_sweep_ms(pedp_ctx_t ctx, float *ms);

/* RANSAC draws of registration_ransac_based_on_feature_matching (src/pose_estimation.py:482-501,
 * ransac_n = 3): iteration itr0 + k, k = 0 .. count-1, takes three correspondences
 * (corr[i] = target point of source point i, e.g. from pedp_feature_match) with replacement, fits
 * Umeyama without scaling and applies the reference's three checkers in its order
 * (CorrespondenceCheckerBasedOnEdgeLength(edge_similarity), ...BasedOnDistance(max_distance),
 * ...BasedOnNormal(normal_angle, radians; skipped when a cloud has no normals)).  accepted[k] = 1 when
 * all pass; T[16 k ..] = the fitted source -> target transformation either way.  The draw is a
 * counter-based function of (seed, iteration) (Open3D's random_device-seeded engines are not
 * recoverable); count <= 2^22 per call. */
int pedp_ransac_hypotheses(pedp_ctx_t ctx, pedp_cloud_t source, pedp_cloud_t target, const int32_t *corr, uint64_t seed,
                           int64_t itr0, int count, double edge_similarity, double max_distance, double normal_angle,
                           uint8_t *accepted, double *T);

/* Measurement knobs of pedp_icp / pedp_icp_batched on this context.
 *   exhaustive = 1: no culling -- every scene point stays a candidate and the MFMA kernel sweeps
 *     every (scene point, target point) pair in every pass (the all-pairs workload of SURVEY s8d);
 *     correspondences and poses are identical to the culled run, only the work differs.
 *   timed_pass >= 0: record HIP events around the sweep kernel of that correspondence pass (read
 *     with pedp_nn_last_sweep_ms); -2: around the sweep kernel of every fourth pass from pass 1 on
 *     (up to eight), pedp_nn_last_sweep_ms then reports their mean; -3 (fused path): ONE pair of
 *     events around all the passes' launches of a registration, pedp_nn_last_sweep_ms reports the
 *     span divided by the number of launches -- the mean launch-to-launch time of the pass kernel,
 *     boundaries included; -1 = none.
 * "Sweep kernel": nn_sweep_kernel on the segmented path, icp_pass_kernel (the whole pass: per-chunk
 * work, sweep included, and its close by the last workgroup) on the fused path of radius-limited
 * registrations. */
int pedp_icp_configure(pedp_ctx_t ctx, int exhaustive, int timed_pass);

/* Work statistics of the last pedp_icp on this context: correspondence passes run, (scene,
 * target) pairs the MFMA sweep evaluated (scene points farther than the radius from the
 * target's bounding box are dropped before the sweep), and points whose fp32 filter was
 * ambiguous and went through the exact float64 brute-force kernel. */
int pedp_icp_last_stats(pedp_ctx_t ctx, int64_t *passes, int64_t *pairs_swept, int64_t *fallback_points);

/* Passes of the last single pedp_icp on this context that ran under a visit plan: with more live scene chunks than
 * CUs (and at most twice as many) the workgroup that is through first ranks the chunks by the time the pass before
 * spent on them and hands the lightest ones to the workgroup positions that share a CU.  Scheduling only -- partial
 * sums stay indexed by live rank, results do not change in any bit (PEDP_ICP_NO_VISIT_PLAN=1 switches it off;
 * tests/test_icp_gpu.py compares).  No counterpart in the reference (src/pose_estimation.py:447-453 calls Open3D). */
int pedp_icp_last_planned_passes(pedp_ctx_t ctx, int64_t *planned);

/* Diagnostics of the dense sweep's bf16 form (tests measure its error against float64): for n_src scene rows (b.x, b.y, b.z, .)
 * = (-2 s') and n_tgt model rows (t'.x, t'.y, t'.z, |t'|^2), float32 x 4 each on the host, counts multiples of 16,
 * g[i * n_tgt + j] = |t'_j|^2 - 2 s'_i . t'_j exactly as the sweep's v_mfma_f32_16x16x32_bf16 produces it (same operand packing). */
int pedp_debug_nn_bf16(pedp_ctx_t ctx, const float *src4, int64_t n_src, const float *tgt4, int64_t n_tgt, float *g);

/* ---------------------------------------------------------------- multi-GPU collectives
 * One process per GPU.  The reference has no distributed path; these entry points
 * exist so that the two exchange steps of the sharded path run inside the library, on
 * the context's stream, over RCCL/xGMI: the all-gather of the ray shards' hit records and the
 * per-pass all-reduce of the ICP packet (pedp_icp_params.use_comm).  RCCL is bound at run time.
 * Rendezvous is the host's business: rank 0 makes the id, the host side hands the same 128 bytes
 * to every rank (pedp_hip.dist does it through torch.distributed), every rank calls create. */
enum { PEDP_COMM_ID_BYTES = 128 };
int pedp_comm_unique_id(uint8_t id[PEDP_COMM_ID_BYTES]);
int pedp_comm_create(pedp_ctx_t ctx, const uint8_t id[PEDP_COMM_ID_BYTES], int nranks, int rank);
int pedp_comm_destroy(pedp_ctx_t ctx);
int pedp_comm_size(pedp_ctx_t ctx, int *nranks, int *rank);
/* Device pointers; only enqueue on the context's stream.  allgather: every rank contributes
 * bytes_per_rank bytes, recv holds nranks * bytes_per_rank in rank order. */
int pedp_comm_allgather(pedp_ctx_t ctx, const void *send, void *recv, int64_t bytes_per_rank);
int pedp_comm_allreduce_f64(pedp_ctx_t ctx, double *buf, int64_t n);

/* ---------------------------------------------------------------- cluster_poses
 * Replaces mycpp.cluster_poses (mycpp/src/app/pybind_api.cpp:24-68; caller
 * estimater.py:118).  Host only.  poses: n x 16 float32 row-major, syms: s x 16.
 * keep_idx: caller array of n ints, receives the indices of the kept poses. */
int pedp_cluster_poses(float angle_diff_deg, float dist_diff, const float *poses, int n,
                       const float *syms, int s, int32_t *keep_idx, int *n_keep);

/* ---------------------------------------------------------------- page-locked host memory
 * Result arrays a caller allocates anew for every frame are paid for twice: the pages of a fresh allocation fault in one
 * by one while the result is copied into them (0.15 ms per megabyte measured inside the frame chain), and a pageable
 * destination goes through the library's staging buffer.  pedp_host_alloc hands out page-locked memory (hipHostMalloc)
 * that downloads reach directly; pedp_hip keeps a pool of such blocks under its result arrays (_lib.host_array). */
int pedp_host_alloc(size_t bytes, void **out);
void pedp_host_free(void *p);

/* ---------------------------------------------------------------- rigid transform of host arrays
 * What o3d.geometry.PointCloud.transform / TriangleMesh.transform do to the holders' float64 arrays on the path
 * (src/pose_estimation.py:406-409 transform_object, :815-816; run.py:118, :197, :200): out_i = R in_i + t with
 * ((T0 x + T1 y) + T2 z) + T3 per row, one rounding per operation (the oracle's order); rotate_only != 0 leaves the
 * translation out (normals).  Host only, out may be in.  A numpy (N, 3) @ (3, 3) product takes ten times as long
 * (a BLAS call with inner dimension 3) and its rounding depends on the BLAS build. */
int pedp_transform_points(const double T[16], const double *in, int64_t n, int rotate_only, double *out);

#ifdef __cplusplus
}
#endif
#endif
