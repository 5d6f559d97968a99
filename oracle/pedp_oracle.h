/*
 * pedp_oracle.h -- CPU ORACLE for the ICP + ray-projection hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package may include, link,
 * import or execute this: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported baseline.
 *
 * PARITY UNPINNED: the reference (/root/reference) holds no tests, fixtures or
 * golden vectors for this path, and its arithmetic lives in
 * third-party wheels that are absent here (open3d==0.18.0 with its bundled
 * Embree and nanoflann; requirements.txt:23).  This oracle is therefore a
 * restatement of (a) the reference's own call sites and control flow
 * (src/pose_estimation.py:505-522, :547-622, :624-683; src/defect_projection.py:
 * 165-266) and (b) the published algorithms of the pinned dependency as
 * summarised in SURVEY.md s3.3 / Appendix A.  It is cross-checked in tests/
 * against independent numpy/scipy maths (cKDTree NN, numpy SVD/solve, analytic
 * known-answer cases), not against the reference's outputs.
 */
#ifndef PEDP_ORACLE_H
#define PEDP_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- ray casting (float32; defect_projection.py:245-264) ---- */

/* Per-triangle record (v0, e1 = v1 - v0, e2 = v2 - v0, m = e2 x e1), PEDP_ORACLE_TRI = 12
 * floats, computed in fp32 from the fp32 vertices TriangleMesh.from_legacy + add_triangles
 * would hand to the ray caster (defect_projection.py:245, :253-254). */
#define PEDP_ORACLE_TRI 12
void pedp_oracle_tri_setup(const float *verts, int64_t V, const uint32_t *tris,
                           int64_t F, float *tri9);

/* One Moeller-Trumbore test in the oracle's fixed operation order.  Returns 1 on
 * hit and writes t,u,v; 0 on miss. */
int pedp_oracle_mt_test(const float o[3], const float d[3], const float *tri12,
                        float *t, float *u, float *v);

/* Closest hit of every ray against every triangle (brute force).  rays6 is N x
 * [ox oy oz dx dy dz] like cast_rays' Float32[N,6] tensor (defect_projection.py:
 * 248-256).  t_hit = +inf and prim_id = 0xFFFFFFFF on miss; ties: smaller t, then
 * smaller triangle index.  uv may be NULL. */
int pedp_oracle_raycast(const float *tri9, int64_t F, const float *rays6, int64_t N,
                        float *t_hit, uint32_t *prim_id, float *uv, int nthreads);

/* Same result through a bounding-volume hierarchy (CPU baseline of the same
 * algorithmic class as Embree; build is part of the call like in the reference,
 * defect_projection.py:253-254).  build_seconds / cast_seconds may be NULL. */
/* All accepted (ray, triangle) pairs (small cases: N x F tests). */
int64_t pedp_oracle_accepted_pairs(const float *tri9, int64_t F, const float *rays6, int64_t N, int32_t *pairs, int64_t capacity);
int pedp_oracle_raycast_bvh(const float *tri9, int64_t F, const float *rays6, int64_t N,
                            float *t_hit, uint32_t *prim_id, float *uv, int nthreads,
                            double *build_seconds, double *cast_seconds);

/* ---- ICP (float64; pose_estimation.py:519-521, :654-660) ---- */

#define PEDP_ORACLE_P2PLANE 0
#define PEDP_ORACLE_P2POINT 1

/* Exact nearest neighbour of every source point among tgt (brute force, f64).
 * idx[i] in [0,Nt), d2[i] = squared distance.  Ties: lowest index. */
void pedp_oracle_nn(const double *src, int64_t Ns, const double *tgt, int64_t Nt,
                    int32_t *idx, double *d2, int nthreads);
/* Same through a KD-tree (CPU baseline in Open3D's algorithmic class). */
void pedp_oracle_nn_kdtree(const double *src, int64_t Ns, const double *tgt, int64_t Nt,
                           int32_t *idx, double *d2, int nthreads);

/* p' = T p for N points (row-major 4x4), fixed op order ((a+b)+c)+d, no FMA. */
void pedp_oracle_transform(const double T[16], const double *pts, int64_t N, double *out);

/* registration_icp restatement.  src/tgt: N x 3 f64; tgt_normals required for
 * point-to-plane (returns -2 if NULL).  init maps source into the target frame.
 * corr (nullable, Ns): target index of each source point's correspondence or -1.
 * trace (nullable): (max_iter+1) x 18 doubles = per pass [fitness, rmse, T(16)];
 * pass 0 is the initial correspondence pass.  use_kdtree selects the NN engine
 * (identical results).  Returns 0 or negative error. */
int pedp_oracle_icp(const double *src, int64_t Ns, const double *tgt,
                    const double *tgt_normals, int64_t Nt, double max_corr_dist,
                    const double init[16], int estimator, int max_iter,
                    double rel_fitness, double rel_rmse, double T_out[16],
                    double *fitness, double *inlier_rmse, int32_t *n_iter_done,
                    int32_t *corr, double *trace, int use_kdtree, int nthreads);

/* Pieces, exported for unit tests. */
int pedp_oracle_solve6_ldlt(const double A[36], const double b[6], double x[6]);
void pedp_oracle_vec6_to_T(const double x[6], double T[16]);
void pedp_oracle_kabsch(const double *S, const double *Tg, int64_t K, double T[16]);
void pedp_oracle_rot_xyz(const double abc[3], double R[9]); /* Rx(a)Ry(b)Rz(c) */

/* point-cloud operations of preprocess_source (cloudops.c): src/pose_estimation.py:186-392 */
int64_t pedp_oracle_voxel_down_sample(const double *pts, const double *normals, int64_t N, double voxel,
                                      double *out_pts, double *out_normals);
void pedp_oracle_dbscan(const double *pts, int64_t N, double eps, int min_points, int32_t *labels);
void pedp_oracle_knn_mean_distance(const double *pts, int64_t N, int k, double *avg);
void pedp_oracle_sample3(uint64_t seed, int64_t t, int64_t N, int64_t out[3]);
void pedp_oracle_plane_from_points(const double *pts, const int32_t *idx, int64_t n, double pl[4]);
int64_t pedp_oracle_segment_plane(const double *pts, int64_t N, double threshold, int num_iterations, uint64_t seed,
                                  double plane[4], int32_t *inliers);
void pedp_oracle_smallest_eigenvector(const double cov[9], double out[3]);
void pedp_oracle_estimate_normals(const double *pts, int64_t N, double radius, int max_nn, const double *prior,
                                  double *out);

/* feature-based global registration (features.c): src/pose_estimation.py:132-137, :467-503 */
void pedp_oracle_fpfh(const double *pts, const double *nrm, int64_t N, double radius, int max_nn, double *out);
void pedp_oracle_feature_match(const double *fs, int64_t Ns, const double *ft, int64_t Nt, int32_t *idx);
int pedp_oracle_ransac_hypothesis(uint64_t seed, int64_t itr, const double *src, const double *src_nrm, int64_t Ns,
                                  const double *tgt, const double *tgt_nrm, const int32_t *corr, double edge,
                                  double dist, double angle, double T[16]);
double pedp_oracle_corres_inlier_ratio(const double *src, int64_t Ns, const double *tgt, const int32_t *corr,
                                       const double T[16], double max_dist);

/* depth pre-filters (depth.c): Utils.py:304-442 */
void pedp_oracle_erode_depth(const float *depth, int H, int W, int radius, float depth_diff_thres, float ratio_thres,
                             float zfar, float *out, int nthreads);
void pedp_oracle_bilateral_depth(const float *depth, int H, int W, int radius, float zfar, float sigmaD, float sigmaR,
                                 float *out, int nthreads);
void pedp_oracle_depth2xyzmap(const float *depth, int H, int W, const double *K, float *xyz);
void pedp_oracle_depth2xyzmap_batch(const float *depths, int B, int H, int W, const float *Ks, float zfar, float *xyz);

/* ---- cluster_poses (float32; mycpp/src/app/pybind_api.cpp:24-68) ---- */
int pedp_oracle_cluster_poses(float angle_diff_deg, float dist_diff, const float *poses,
                              int n, const float *syms, int s, int32_t *keep_idx,
                              int *n_keep);

#ifdef __cplusplus
}
#endif
#endif
