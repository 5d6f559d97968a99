/* TEST INFRASTRUCTURE -- CPU restatement of the Open3D point-cloud operations that
 * preprocess_source chains (src/pose_estimation.py:186-268), written from the published
 * open3d==0.18.0 algorithms (the wheel is absent; parity unpinned).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 *   pedp_oracle_voxel_down_sample    PointCloud::VoxelDownSample     (pose_estimation.py:204-205)
 *   pedp_oracle_dbscan               PointCloud::ClusterDBSCAN       (:284, eps 10, min_points 10)
 *   pedp_oracle_knn_mean_distance    PointCloud::RemoveStatisticalOutliers, per-point part (:308-312)
 *   pedp_oracle_segment_plane        PointCloud::SegmentPlane        (:323-329)
 *
 * Where Open3D's result depends on something that is not recoverable (hash-map iteration order,
 * a random_device seed, thread interleaving) the choice made here is stated at the function and
 * is the definition the HIP side is tested against.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "pedp_oracle.h"

static double d2(const double *a, const double *b) {
    const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;
}

/* ---- voxel grid.  Open3D: voxel_min_bound = min_bound - voxel_size / 2, index = floor((p -
 * voxel_min_bound) / voxel_size) per axis, the points (and normals) of a voxel are summed in
 * point order and divided by the count.  Output order: Open3D iterates an unordered_map (not
 * recoverable); here voxels come out in ascending (ix, iy, iz). */
typedef struct { uint64_t key; int64_t idx; } keyed;
static int cmp_keyed(const void *a, const void *b) {
    const keyed *x = (const keyed *)a, *y = (const keyed *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

int64_t pedp_oracle_voxel_down_sample(const double *pts, const double *normals, int64_t N, double voxel,
                                      double *out_pts, double *out_normals) {
    if (N <= 0) return 0;
    double lo[3] = {pts[0], pts[1], pts[2]};
    for (int64_t i = 1; i < N; ++i)
        for (int k = 0; k < 3; ++k)
            if (pts[3 * i + k] < lo[k]) lo[k] = pts[3 * i + k];
    for (int k = 0; k < 3; ++k) lo[k] = lo[k] - voxel * 0.5;
    keyed *ks = (keyed *)malloc(sizeof(keyed) * (size_t)N);
    for (int64_t i = 0; i < N; ++i) {
        uint64_t key = 0;
        for (int k = 0; k < 3; ++k) {
            const int64_t c = (int64_t)floor((pts[3 * i + k] - lo[k]) / voxel);
            key = (key << 21) | ((uint64_t)c & 0x1FFFFF);
        }
        ks[i].key = key;
        ks[i].idx = i;
    }
    qsort(ks, (size_t)N, sizeof(keyed), cmp_keyed);
    int64_t n_out = 0;
    for (int64_t a = 0; a < N;) {
        int64_t b = a;
        double s[3] = {0, 0, 0}, sn[3] = {0, 0, 0};
        while (b < N && ks[b].key == ks[a].key) {
            const int64_t i = ks[b].idx;
            for (int k = 0; k < 3; ++k) s[k] += pts[3 * i + k];
            if (normals)
                for (int k = 0; k < 3; ++k) sn[k] += normals[3 * i + k];
            ++b;
        }
        const double cnt = (double)(b - a);
        for (int k = 0; k < 3; ++k) out_pts[3 * n_out + k] = s[k] / cnt;
        if (normals && out_normals)
            for (int k = 0; k < 3; ++k) out_normals[3 * n_out + k] = sn[k] / cnt;
        ++n_out;
        a = b;
    }
    free(ks);
    return n_out;
}

/* ---- DBSCAN, Open3D's breadth-first form: neighbours = points with d^2 < eps^2 (nanoflann's
 * radius search is strict; the point itself counts), core iff at least min_points neighbours;
 * points are scanned in index order, every unvisited core point starts the next cluster and the
 * cluster grows through the neighbour lists of its core points; a non-core point reached by a
 * cluster takes that cluster's label if it is still unlabelled or noise.  Labels: -1 noise. */
void pedp_oracle_dbscan(const double *pts, int64_t N, double eps, int min_points, int32_t *labels) {
    const double e2 = eps * eps;
    int32_t *cnt = (int32_t *)calloc((size_t)(N > 0 ? N : 1), sizeof(int32_t));
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < N; ++i) {
        int32_t c = 0;
        for (int64_t j = 0; j < N; ++j)
            if (d2(pts + 3 * i, pts + 3 * j) < e2) ++c;
        cnt[i] = c;
    }
    for (int64_t i = 0; i < N; ++i) labels[i] = -2; /* undefined */
    int64_t *queue = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
    int32_t cluster = 0;
    for (int64_t i = 0; i < N; ++i) {
        if (labels[i] != -2) continue;
        if (cnt[i] < min_points) { labels[i] = -1; continue; }
        labels[i] = cluster;
        int64_t head = 0, tail = 0;
        queue[tail++] = i;
        while (head < tail) {
            const int64_t q = queue[head++];  /* q is a core point of this cluster */
            for (int64_t j = 0; j < N; ++j) {
                if (!(d2(pts + 3 * q, pts + 3 * j) < e2)) continue;
                if (labels[j] == -1) { labels[j] = cluster; continue; } /* noise becomes border */
                if (labels[j] != -2) continue;
                labels[j] = cluster;
                if (cnt[j] >= min_points) queue[tail++] = j;
            }
        }
        ++cluster;
    }
    free(queue);
    free(cnt);
}

/* ---- statistical outlier removal, per-point part: mean of the distances to the k nearest
 * neighbours (the point itself is one of them, as in Open3D's SearchKNN on its own cloud), summed
 * in ascending order of distance.  -1 when the cloud is empty.  The global mean / Bessel-corrected
 * deviation / threshold are formed by the caller in index order (pedp_oracle.py). */
static int cmp_double(const void *a, const void *b) {
    const double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : (x > y);
}
void pedp_oracle_knn_mean_distance(const double *pts, int64_t N, int k, double *avg) {
#pragma omp parallel
    {
        double *d = (double *)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
#pragma omp for schedule(dynamic, 16)
        for (int64_t i = 0; i < N; ++i) {
            for (int64_t j = 0; j < N; ++j) d[j] = d2(pts + 3 * i, pts + 3 * j);
            qsort(d, (size_t)N, sizeof(double), cmp_double);
            const int64_t m = N < k ? N : k;
            double s = 0.0;
            for (int64_t j = 0; j < m; ++j) s += sqrt(d[j]);
            avg[i] = m > 0 ? s / (double)m : -1.0;
        }
        free(d);
    }
}

/* ---- plane RANSAC.  Open3D draws its samples from a random_device-seeded engine and stops early
 * by a probability rule evaluated under OpenMP: neither is recoverable.  Definition here: iteration
 * t samples the three distinct indices pedp_oracle_sample3(seed, t, N); plane through them
 * (TriangleMesh::ComputeTrianglePlane: unit normal of (p1-p0)x(p2-p0), d = -n.p0), degenerate
 * samples skipped; inliers |n.p + d| < threshold; the best iteration has the most inliers, ties go
 * to the earliest iteration; all num_iterations are evaluated.  Returned like Open3D: the inliers
 * of the best sampled plane, and the plane refitted to them (GetPlaneFromPoints: centroid +
 * largest-determinant closed form). */
static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
void pedp_oracle_sample3(uint64_t seed, int64_t t, int64_t N, int64_t out[3]) {
    uint64_t s = splitmix64(seed ^ splitmix64((uint64_t)t));
    int n = 0;
    while (n < 3) {
        s = splitmix64(s);
        const int64_t c = (int64_t)(s % (uint64_t)N);
        int dup = 0;
        for (int q = 0; q < n; ++q) dup |= (out[q] == c);
        if (!dup) out[n++] = c;
    }
}
static int triangle_plane(const double *p0, const double *p1, const double *p2, double pl[4]) {
    const double a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, b[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
    const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    const double n = sqrt((x * x + y * y) + z * z);
    if (!(n > 0.0)) return 0;
    pl[0] = x / n; pl[1] = y / n; pl[2] = z / n;
    pl[3] = -((pl[0] * p0[0] + pl[1] * p0[1]) + pl[2] * p0[2]);
    return 1;
}
static double plane_dist(const double pl[4], const double *p) {
    return fabs(((pl[0] * p[0] + pl[1] * p[1]) + pl[2] * p[2]) + pl[3]);
}
/* GetPlaneFromPoints (Open3D 0.18 PointCloudSegmentation.cpp: centroid of the inliers, their six second moments about it,
 * the largest-determinant closed form).  Open3D adds the inliers up one after the other; here every sum is taken in BLOCKS
 * of the CLOUD: the 256 points [256 j, 256 j + 256) contribute their value if they are inliers and zero otherwise, a block
 * is added in a fixed binary tree -- t[i] += t[i + w] for w = 128, 64 ... 1 -- and the block sums are added in order of j.
 * That is the order a GPU takes without a trip to the host and without compacting the inliers (pedp_preprocess_source
 * refits on the device), a pairwise-style sum that lies closer to the exact one than the running sum, and it differs from
 * Open3D's result by rounding only (1e-16 relative; the build's tolerance to Open3D's arithmetic is 1e-5).  Every product
 * and difference is one rounding, no contraction.  idx must ascend (segment_plane lists the inliers in index order). */
static double block_tree_sum(double *t /* 256 values, overwritten */) {
    for (int w = 128; w >= 1; w >>= 1)
        for (int i = 0; i < w; ++i) t[i] += t[i + w];
    return t[0];
}
/* sum over the inliers of f_q(point), q < nq, blocked as above; r0: subtracted from the point first (NULL: nothing) */
static void blocked_sums(const double *pts, const int32_t *idx, int64_t n, const double *r0, int nq, double *out) {
    static const int A[6] = {0, 0, 0, 1, 1, 2}, B[6] = {0, 1, 2, 1, 2, 2};
    double t[6][256];
    for (int q = 0; q < nq; ++q) out[q] = 0.0;
    int64_t i = 0;
    while (i < n) {
        const int64_t blk = idx[i] / 256;
        for (int q = 0; q < nq; ++q)
            for (int k = 0; k < 256; ++k) t[q][k] = 0.0;
        for (; i < n && idx[i] / 256 == blk; ++i) {
            const double *p = pts + 3 * (int64_t)idx[i];
            const int k = (int)(idx[i] % 256);
            if (!r0) { t[0][k] = p[0]; t[1][k] = p[1]; t[2][k] = p[2]; continue; }
            const double r[3] = {p[0] - r0[0], p[1] - r0[1], p[2] - r0[2]};
            for (int q = 0; q < nq; ++q) t[q][k] = r[A[q]] * r[B[q]];
        }
        for (int q = 0; q < nq; ++q) out[q] += block_tree_sum(t[q]);
    }
}
void pedp_oracle_plane_from_points(const double *pts, const int32_t *idx, int64_t n, double pl[4]) {
    pl[0] = pl[1] = pl[2] = pl[3] = 0.0;
    if (n < 3) return;
    double c[3], m[6];
    blocked_sums(pts, idx, n, NULL, 3, c);
    for (int k = 0; k < 3; ++k) c[k] /= (double)n;
    blocked_sums(pts, idx, n, c, 6, m);
    const double xx = m[0], xy = m[1], xz = m[2], yy = m[3], yz = m[4], zz = m[5];
    const double det_x = yy * zz - yz * yz, det_y = xx * zz - xz * xz, det_z = xx * yy - xy * xy;
    double a, b, cc;
    if (det_x >= det_y && det_x >= det_z) { a = det_x; b = xz * yz - xy * zz; cc = xy * yz - xz * yy; }
    else if (det_y >= det_z) { a = xz * yz - xy * zz; b = det_y; cc = xy * xz - yz * xx; }
    else { a = xy * yz - xz * yy; b = xy * xz - yz * xx; cc = det_z; }
    const double nrm = sqrt((a * a + b * b) + cc * cc);
    if (!(nrm > 0.0)) return;
    pl[0] = a / nrm; pl[1] = b / nrm; pl[2] = cc / nrm;
    pl[3] = -((pl[0] * c[0] + pl[1] * c[1]) + pl[2] * c[2]);
}
int64_t pedp_oracle_segment_plane(const double *pts, int64_t N, double threshold, int num_iterations, uint64_t seed,
                                  double plane[4], int32_t *inliers) {
    plane[0] = plane[1] = plane[2] = plane[3] = 0.0;
    if (N < 3) return 0;
    int64_t best_cnt = -1;
    double best[4] = {0, 0, 0, 0};
    for (int t = 0; t < num_iterations; ++t) {
        int64_t s[3];
        double pl[4];
        pedp_oracle_sample3(seed, t, N, s);
        if (!triangle_plane(pts + 3 * s[0], pts + 3 * s[1], pts + 3 * s[2], pl)) continue;
        int64_t cnt = 0;
#pragma omp parallel for reduction(+ : cnt)
        for (int64_t i = 0; i < N; ++i) cnt += plane_dist(pl, pts + 3 * i) < threshold;
        if (cnt > best_cnt) { best_cnt = cnt; memcpy(best, pl, sizeof(best)); }
    }
    if (best_cnt < 0) return 0;
    int64_t n = 0;
    for (int64_t i = 0; i < N; ++i)
        if (plane_dist(best, pts + 3 * i) < threshold) inliers[n++] = (int32_t)i;
    pedp_oracle_plane_from_points(pts, inliers, n, plane);
    return n;
}

/* ---- normal estimation, PointCloud::EstimateNormals with KDTreeSearchParamHybrid(radius, max_nn)
 * (src/pose_estimation.py:301-306): neighbours = the max_nn nearest points with d^2 < radius^2 (the
 * point itself included), taken in ascending (d^2, index); with >= 3 of them the covariance from
 * the nine cumulants (utility::ComputeCovariance) and the eigenvector of its smallest eigenvalue by
 * the non-iterative solver Open3D uses (FastEigen3x3, after Eberly, "A Robust Eigensolver for 3x3
 * Symmetric Matrices"); otherwise the identity covariance, whose "smallest" eigenvector is (0,0,1).
 * A zero normal falls back to the prior normal or (0,0,1); with prior normals the result is flipped
 * to agree with them.  The eigenvector's sign is otherwise whatever the solver yields (Open3D does
 * not orient it either); point-to-plane ICP does not depend on it. */
static void cross3(const double *a, const double *b, double *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static void eigenvector0(const double A[9], double e, double *out) {
    const double r0[3] = {A[0] - e, A[1], A[2]}, r1[3] = {A[1], A[4] - e, A[5]}, r2[3] = {A[2], A[5], A[8] - e};
    double c01[3], c02[3], c12[3];
    cross3(r0, r1, c01); cross3(r0, r2, c02); cross3(r1, r2, c12);
    const double d0 = dot3(c01, c01), d1 = dot3(c02, c02), d2_ = dot3(c12, c12);
    const double *best = c01;
    double dm = d0;
    if (d1 > dm) { dm = d1; best = c02; }
    if (d2_ > dm) { dm = d2_; best = c12; }
    const double s = sqrt(dm);
    for (int k = 0; k < 3; ++k) out[k] = best[k] / s;
}
static void eigenvector1(const double A[9], const double *ev0, double e, double *out) {
    double U[3], V[3];
    if (fabs(ev0[0]) > fabs(ev0[1])) {
        const double inv = 1.0 / sqrt(ev0[0] * ev0[0] + ev0[2] * ev0[2]);
        U[0] = -ev0[2] * inv; U[1] = 0.0; U[2] = ev0[0] * inv;
    } else {
        const double inv = 1.0 / sqrt(ev0[1] * ev0[1] + ev0[2] * ev0[2]);
        U[0] = 0.0; U[1] = ev0[2] * inv; U[2] = -ev0[1] * inv;
    }
    cross3(ev0, U, V);
    const double AU[3] = {(A[0] * U[0] + A[1] * U[1]) + A[2] * U[2], (A[1] * U[0] + A[4] * U[1]) + A[5] * U[2],
                          (A[2] * U[0] + A[5] * U[1]) + A[8] * U[2]};
    const double AV[3] = {(A[0] * V[0] + A[1] * V[1]) + A[2] * V[2], (A[1] * V[0] + A[4] * V[1]) + A[5] * V[2],
                          (A[2] * V[0] + A[5] * V[1]) + A[8] * V[2]};
    double m00 = dot3(U, AU) - e, m01 = dot3(U, AV), m11 = dot3(V, AV) - e;
    const double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
    if (a00 >= a11) {
        if (fmax(a00, a01) > 0.0) {
            if (a00 >= a01) { m01 /= m00; m00 = 1.0 / sqrt(1.0 + m01 * m01); m01 *= m00; }
            else { m00 /= m01; m01 = 1.0 / sqrt(1.0 + m00 * m00); m00 *= m01; }
            for (int k = 0; k < 3; ++k) out[k] = m01 * U[k] - m00 * V[k];
        } else {
            for (int k = 0; k < 3; ++k) out[k] = U[k];
        }
    } else {
        if (fmax(a11, a01) > 0.0) {
            if (a11 >= a01) { m01 /= m11; m11 = 1.0 / sqrt(1.0 + m01 * m01); m01 *= m11; }
            else { m11 /= m01; m01 = 1.0 / sqrt(1.0 + m11 * m11); m11 *= m01; }
            for (int k = 0; k < 3; ++k) out[k] = m11 * U[k] - m01 * V[k];
        } else {
            for (int k = 0; k < 3; ++k) out[k] = U[k];
        }
    }
}
void pedp_oracle_smallest_eigenvector(const double cov[9], double out[3]) {
    double A[9];
    double mx = cov[0];
    for (int k = 1; k < 9; ++k) if (cov[k] > mx) mx = cov[k];
    out[0] = out[1] = out[2] = 0.0;
    if (mx == 0.0) return;
    for (int k = 0; k < 9; ++k) A[k] = cov[k] / mx;
    const double norm = (A[1] * A[1] + A[2] * A[2]) + A[5] * A[5];
    if (norm > 0.0) {
        const double q = ((A[0] + A[4]) + A[8]) / 3.0;
        const double b00 = A[0] - q, b11 = A[4] - q, b22 = A[8] - q;
        const double p = sqrt((((b00 * b00 + b11 * b11) + b22 * b22) + norm * 2.0) / 6.0);
        const double c00 = b11 * b22 - A[5] * A[5], c01 = A[1] * b22 - A[5] * A[2], c02 = A[1] * A[5] - b11 * A[2];
        const double det = ((b00 * c00 - A[1] * c01) + A[2] * c02) / ((p * p) * p);
        double half = det * 0.5;
        half = half < -1.0 ? -1.0 : (half > 1.0 ? 1.0 : half);
        const double angle = acos(half) / 3.0;
        const double beta2 = cos(angle) * 2.0, beta0 = cos(angle + 2.09439510239319549) * 2.0, beta1 = -(beta0 + beta2);
        const double e0 = q + p * beta0, e1 = q + p * beta1, e2 = q + p * beta2;
        double v0[3], v1[3], v2[3];
        if (half >= 0.0) {
            eigenvector0(A, e2, v2);
            if (e2 < e0 && e2 < e1) { memcpy(out, v2, sizeof(v2)); return; }
            eigenvector1(A, v2, e1, v1);
            if (e1 < e0 && e1 < e2) { memcpy(out, v1, sizeof(v1)); return; }
            cross3(v1, v2, out);
        } else {
            eigenvector0(A, e0, v0);
            if (e0 < e1 && e0 < e2) { memcpy(out, v0, sizeof(v0)); return; }
            eigenvector1(A, v0, e1, v1);
            if (e1 < e0 && e1 < e2) { memcpy(out, v1, sizeof(v1)); return; }
            cross3(v0, v1, out);
        }
    } else { /* diagonal */
        if (A[0] < A[4] && A[0] < A[8]) out[0] = 1.0;
        else if (A[4] < A[0] && A[4] < A[8]) out[1] = 1.0;
        else out[2] = 1.0;
    }
}

typedef struct { double d; int64_t j; } cand;
static int cmp_cand(const void *a, const void *b) {
    const cand *x = (const cand *)a, *y = (const cand *)b;
    if (x->d != y->d) return x->d < y->d ? -1 : 1;
    return x->j < y->j ? -1 : (x->j > y->j);
}
void pedp_oracle_estimate_normals(const double *pts, int64_t N, double radius, int max_nn, const double *prior,
                                  double *out) {
    const double r2 = radius * radius;
#pragma omp parallel
    {
        cand *c = (cand *)malloc(sizeof(cand) * (size_t)(N > 0 ? N : 1));
#pragma omp for schedule(dynamic, 32)
        for (int64_t i = 0; i < N; ++i) {
            int64_t n = 0;
            for (int64_t j = 0; j < N; ++j) {
                const double d = d2(pts + 3 * i, pts + 3 * j);
                if (d < r2) { c[n].d = d; c[n].j = j; ++n; }
            }
            qsort(c, (size_t)n, sizeof(cand), cmp_cand);
            if (n > max_nn) n = max_nn;
            double cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            if (n >= 3) {
                double cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                for (int64_t q = 0; q < n; ++q) {
                    const double *p = pts + 3 * c[q].j;
                    cu[0] += p[0]; cu[1] += p[1]; cu[2] += p[2];
                    cu[3] += p[0] * p[0]; cu[4] += p[0] * p[1]; cu[5] += p[0] * p[2];
                    cu[6] += p[1] * p[1]; cu[7] += p[1] * p[2]; cu[8] += p[2] * p[2];
                }
                for (int k = 0; k < 9; ++k) cu[k] /= (double)n;
                cov[0] = cu[3] - cu[0] * cu[0]; cov[4] = cu[6] - cu[1] * cu[1]; cov[8] = cu[8] - cu[2] * cu[2];
                cov[1] = cov[3] = cu[4] - cu[0] * cu[1];
                cov[2] = cov[6] = cu[5] - cu[0] * cu[2];
                cov[5] = cov[7] = cu[7] - cu[1] * cu[2];
            }
            double nrm[3];
            pedp_oracle_smallest_eigenvector(cov, nrm);
            if (sqrt(dot3(nrm, nrm)) == 0.0) {
                if (prior) memcpy(nrm, prior + 3 * i, sizeof(nrm));
                else { nrm[0] = 0.0; nrm[1] = 0.0; nrm[2] = 1.0; }
            }
            if (prior && dot3(nrm, prior + 3 * i) < 0.0)
                for (int k = 0; k < 3; ++k) nrm[k] = -nrm[k];
            memcpy(out + 3 * i, nrm, sizeof(nrm));
        }
        free(c);
    }
}
