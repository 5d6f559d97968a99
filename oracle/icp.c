/*
 * icp.c -- ORACLE (test infrastructure, see pedp_oracle.h): float64 restatement of
 * open3d==0.18.0 registration_icp as the reference calls it.
 *
 * Reference call sites: src/pose_estimation.py:519-521 (refine_registration:
 * registration_icp(source, target, distance_threshold, transformation,
 * TransformationEstimationPointToPlane()), default criteria) and :654-660
 * (predict_z_axis_adjustment: same with ICPConvergenceCriteria(max_iteration=1)).
 * The arithmetic itself is in the absent open3d wheel [3P]; semantics restated
 * from SURVEY.md s3.3 / Appendix A.1:
 *   - correspondence = exact nearest target point (KDTreeFlann::SearchHybrid(p, r, 1)),
 *     kept iff d^2 < r^2 (strict);
 *   - fitness = K / N_source, inlier_rmse = sqrt(sum d^2 / K), both 0 when K = 0;
 *   - point-to-plane update: r = (s - t).n, J = [s x n ; n], solve (sum J J^T) x = -sum J r
 *     by pivoted LDLT, update = [Rz(x2) Ry(x1) Rx(x0) | x3..5];
 *   - point-to-point update: Umeyama without scaling (centroids + 3x3 SVD);
 *   - loop: initial pass, then update -> T = update*T -> transform the already
 *     transformed cloud -> new pass -> stop when |d fitness| < rel_fitness and
 *     |d rmse| < rel_rmse.
 * Compile with -ffp-contract=off.
 */
#include "pedp_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ small linear algebra */

static void mat4_mul(const double A[16], const double B[16], double C[16]) {
    double R[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
            R[4 * i + j] = s;
        }
    memcpy(C, R, sizeof(R));
}

static void mat4_identity(double T[16]) {
    memset(T, 0, 16 * sizeof(double));
    T[0] = T[5] = T[10] = T[15] = 1.0;
}

void pedp_oracle_transform(const double T[16], const double *pts, int64_t N, double *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        out[3 * i + 0] = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
        out[3 * i + 1] = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
        out[3 * i + 2] = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
    }
}

/* Eigen-style LDLT (left-looking, pivot = largest remaining ORIGINAL diagonal entry),
 * then P^T L^-T D^-1 L^-1 P b with D entries <= DBL_MIN treated as zero. */
int pedp_oracle_solve6_ldlt(const double Ain[36], const double b[6], double x[6]) {
    enum { n = 6 };
    double A[n][n], tmp[n];
    int tr[n];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) A[i][j] = Ain[n * i + j];
    for (int k = 0; k < n; ++k) {
        int p = k;
        double big = fabs(A[k][k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(A[i][i]) > big) { big = fabs(A[i][i]); p = i; }
        tr[k] = p;
        if (p != k) { /* symmetric swap on the lower triangle (full storage kept consistent) */
            for (int j = 0; j < n; ++j) { double t = A[k][j]; A[k][j] = A[p][j]; A[p][j] = t; }
            for (int i = 0; i < n; ++i) { double t = A[i][k]; A[i][k] = A[i][p]; A[i][p] = t; }
        }
        if (k > 0) {
            for (int j = 0; j < k; ++j) tmp[j] = A[j][j] * A[k][j];
            double s = 0.0;
            for (int j = 0; j < k; ++j) s += A[k][j] * tmp[j];
            A[k][k] -= s;
            for (int i = k + 1; i < n; ++i) {
                double u = 0.0;
                for (int j = 0; j < k; ++j) u += A[i][j] * tmp[j];
                A[i][k] -= u;
            }
        }
        double akk = A[k][k];
        if (fabs(akk) > 0.0)
            for (int i = k + 1; i < n; ++i) A[i][k] /= akk;
    }
    double y[n];
    for (int i = 0; i < n; ++i) y[i] = b[i];
    for (int k = 0; k < n; ++k)
        if (tr[k] != k) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    for (int i = 0; i < n; ++i) /* L^-1 */
        for (int j = 0; j < i; ++j) y[i] -= A[i][j] * y[j];
    for (int i = 0; i < n; ++i) {
        if (fabs(A[i][i]) > DBL_MIN) y[i] /= A[i][i];
        else y[i] = 0.0;
    }
    for (int i = n - 1; i >= 0; --i) /* L^-T */
        for (int j = i + 1; j < n; ++j) y[i] -= A[j][i] * y[j];
    for (int k = n - 1; k >= 0; --k)
        if (tr[k] != k) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    int ok = 1;
    for (int i = 0; i < n; ++i) {
        x[i] = y[i];
        if (!(y[i] == y[i]) || isinf(y[i])) ok = 0;
    }
    return ok;
}

/* TransformVector6dToMatrix4d: R = Rz(x2) Ry(x1) Rx(x0), t = x3..5 */
void pedp_oracle_vec6_to_T(const double x[6], double T[16]) {
    double ca = cos(x[0]), sa = sin(x[0]);
    double cb = cos(x[1]), sb = sin(x[1]);
    double cc = cos(x[2]), sc = sin(x[2]);
    mat4_identity(T);
    T[0] = cc * cb;  T[1] = cc * sb * sa - sc * ca;  T[2] = cc * sb * ca + sc * sa;
    T[4] = sc * cb;  T[5] = sc * sb * sa + cc * ca;  T[6] = sc * sb * ca - cc * sa;
    T[8] = -sb;      T[9] = cb * sa;                 T[10] = cb * ca;
    T[3] = x[3]; T[7] = x[4]; T[11] = x[5];
}

/* open3d.geometry.get_rotation_matrix_from_xyz: Rx(a) Ry(b) Rz(c)
 * (pose_estimation.py:584) */
void pedp_oracle_rot_xyz(const double abc[3], double R[9]) {
    double ca = cos(abc[0]), sa = sin(abc[0]);
    double cb = cos(abc[1]), sb = sin(abc[1]);
    double cc = cos(abc[2]), sc = sin(abc[2]);
    R[0] = cb * cc;                 R[1] = -cb * sc;                R[2] = sb;
    R[3] = sa * sb * cc + ca * sc;  R[4] = -sa * sb * sc + ca * cc; R[5] = -sa * cb;
    R[6] = -ca * sb * cc + sa * sc; R[7] = ca * sb * sc + sa * cc;  R[8] = ca * cb;
}

/* 3x3 SVD by one-sided Jacobi: A = U diag(w) V^T, w sorted descending. */
static void svd3(const double Ain[9], double U[9], double w[3], double V[9]) {
    double A[3][3], Vv[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = Ain[3 * i + j];
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; ++i) {
                    alpha += A[i][p] * A[i][p];
                    beta += A[i][q] * A[i][q];
                    gamma += A[i][p] * A[i][q];
                }
                if (gamma == 0.0) continue;
                off = fmax(off, fabs(gamma) / sqrt(fmax(alpha * beta, DBL_MIN)));
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 3; ++i) {
                    double ap = A[i][p], aq = A[i][q];
                    A[i][p] = c * ap - s * aq;
                    A[i][q] = s * ap + c * aq;
                    double vp = Vv[i][p], vq = Vv[i][q];
                    Vv[i][p] = c * vp - s * vq;
                    Vv[i][q] = s * vp + c * vq;
                }
            }
        if (off < 1e-16) break;
    }
    double nrm[3];
    int ord[3] = {0, 1, 2};
    for (int j = 0; j < 3; ++j) nrm[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
    for (int a = 0; a < 2; ++a)
        for (int b2 = a + 1; b2 < 3; ++b2)
            if (nrm[ord[b2]] > nrm[ord[a]]) { int t = ord[a]; ord[a] = ord[b2]; ord[b2] = t; }
    double Um[3][3];
    double tiny = nrm[ord[0]] * 1e-300 + DBL_MIN;
    for (int k = 0; k < 3; ++k) {
        int j = ord[k];
        w[k] = nrm[j];
        for (int i = 0; i < 3; ++i) {
            V[3 * i + k] = Vv[i][j];
            Um[i][k] = (nrm[j] > tiny) ? A[i][j] / nrm[j] : 0.0;
        }
    }
    /* complete U for rank-deficient input (planar / collinear sets) */
    double rel = 1e-13 * w[0];
    if (w[0] <= tiny) { /* zero matrix */
        for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) Um[i][k] = (i == k);
    } else {
        if (w[1] <= rel) { /* rank 1: any unit vector orthogonal to u0 */
            double a[3] = {Um[0][0], Um[1][0], Um[2][0]};
            int m = (fabs(a[0]) <= fabs(a[1]) && fabs(a[0]) <= fabs(a[2])) ? 0 : (fabs(a[1]) <= fabs(a[2]) ? 1 : 2);
            double e[3] = {0, 0, 0};
            e[m] = 1.0;
            double dt = a[m];
            double b2[3] = {e[0] - dt * a[0], e[1] - dt * a[1], e[2] - dt * a[2]};
            double nb = sqrt(b2[0] * b2[0] + b2[1] * b2[1] + b2[2] * b2[2]);
            for (int i = 0; i < 3; ++i) Um[i][1] = b2[i] / nb;
        }
        if (w[2] <= rel) { /* u2 = u0 x u1 */
            Um[0][2] = Um[1][0] * Um[2][1] - Um[2][0] * Um[1][1];
            Um[1][2] = Um[2][0] * Um[0][1] - Um[0][0] * Um[2][1];
            Um[2][2] = Um[0][0] * Um[1][1] - Um[1][0] * Um[0][1];
        }
    }
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) U[3 * i + k] = Um[i][k];
}

static double det3(const double M[9]) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* Eigen::umeyama(src, dst, with_scaling=false): T maps S onto Tg. */
void pedp_oracle_kabsch(const double *S, const double *Tg, int64_t K, double T[16]) {
    mat4_identity(T);
    if (K <= 0) return;
    double ms[3] = {0, 0, 0}, mt[3] = {0, 0, 0};
    for (int64_t i = 0; i < K; ++i)
        for (int k = 0; k < 3; ++k) { ms[k] += S[3 * i + k]; mt[k] += Tg[3 * i + k]; }
    for (int k = 0; k < 3; ++k) { ms[k] /= (double)K; mt[k] /= (double)K; }
    double sig[9] = {0};
    for (int64_t i = 0; i < K; ++i)
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                sig[3 * a + b] += (Tg[3 * i + a] - mt[a]) * (S[3 * i + b] - ms[b]);
    for (int k = 0; k < 9; ++k) sig[k] /= (double)K;
    double U[9], w[3], V[9];
    svd3(sig, U, w, V);
    double sgn = (det3(U) * det3(V) < 0.0) ? -1.0 : 1.0;
    double R[9];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            R[3 * a + b] = U[3 * a + 0] * V[3 * b + 0] + U[3 * a + 1] * V[3 * b + 1] +
                           sgn * U[3 * a + 2] * V[3 * b + 2];
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) T[4 * a + b] = R[3 * a + b];
        T[4 * a + 3] = mt[a] - (R[3 * a] * ms[0] + R[3 * a + 1] * ms[1] + R[3 * a + 2] * ms[2]);
    }
}

/* ------------------------------------------------------------------ nearest neighbour */

static inline double dist2(const double *a, const double *b) {
    double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;
}

void pedp_oracle_nn(const double *src, int64_t Ns, const double *tgt, int64_t Nt, int32_t *idx,
                    double *d2, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < Ns; ++i) {
        double best = INFINITY;
        int32_t bi = -1;
        for (int64_t j = 0; j < Nt; ++j) {
            double d = dist2(src + 3 * i, tgt + 3 * j);
            if (d < best) { best = d; bi = (int32_t)j; }
        }
        idx[i] = bi;
        d2[i] = best;
    }
}

/* KD-tree in the manner of nanoflann (the tree Open3D's KDTreeFlann wraps): median split on
 * the widest axis, leaves of <= 16 points, and a search that carries the per-axis squared
 * distance from the query to the current cell (incremental box distance), seeded with the
 * distance to the root bounding box -- so far-away queries prune almost everything after the
 * first leaf.  Like KDTreeFlann::SearchHybrid(.., max_nn = 1) it is a plain 1-NN search; the
 * radius is applied afterwards by the caller. */
typedef struct {
    int32_t first, count; /* leaf: slots in perm; internal: count == 0 */
    int32_t left, right;
    int32_t axis;
    double divlow, divhigh; /* max of left / min of right along axis */
} kd_node;
typedef struct {
    kd_node *nodes;
    int32_t n_nodes;
    int32_t *perm;
    const double *pts;
    double lo[3], hi[3];
} kd_t;

static int kd_axis;
static const double *kd_pts;
static int kd_cmp(const void *a, const void *b) {
    double x = kd_pts[3 * (int64_t)(*(const int32_t *)a) + kd_axis];
    double y = kd_pts[3 * (int64_t)(*(const int32_t *)b) + kd_axis];
    return (x > y) - (x < y);
}

static int32_t kd_build_rec(kd_t *t, int32_t first, int32_t count) {
    int32_t me = t->n_nodes++;
    kd_node *nd = &t->nodes[me];
    nd->first = first; nd->count = 0; nd->left = nd->right = -1; nd->axis = 0; nd->divlow = nd->divhigh = 0;
    if (count <= 16) { nd->count = count; return me; }
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int32_t i = first; i < first + count; ++i)
        for (int k = 0; k < 3; ++k) {
            double v = t->pts[3 * (int64_t)t->perm[i] + k];
            if (v < lo[k]) lo[k] = v;
            if (v > hi[k]) hi[k] = v;
        }
    int ax = 0;
    if (hi[1] - lo[1] > hi[ax] - lo[ax]) ax = 1;
    if (hi[2] - lo[2] > hi[ax] - lo[ax]) ax = 2;
    kd_axis = ax;
    kd_pts = t->pts;
    qsort(t->perm + first, (size_t)count, sizeof(int32_t), kd_cmp);
    int32_t nl = count / 2;
    double divlow = t->pts[3 * (int64_t)t->perm[first + nl - 1] + ax];
    double divhigh = t->pts[3 * (int64_t)t->perm[first + nl] + ax];
    int32_t l = kd_build_rec(t, first, nl);
    int32_t r = kd_build_rec(t, first + nl, count - nl);
    nd = &t->nodes[me]; /* nodes array is preallocated: pointer stays valid */
    nd->axis = ax; nd->divlow = divlow; nd->divhigh = divhigh; nd->left = l; nd->right = r;
    return me;
}

/* mind = lower bound of the squared distance from q to any point of this cell, dists[] its
 * per-axis parts.  A cell is skipped only if its bound, shrunk by a relative 1e-12 against
 * rounding, still exceeds the best distance: ties are never pruned. */
static void kd_search(const kd_t *t, int32_t ni, const double *q, double mind, double dists[3],
                      double *best, int32_t *bi) {
    const kd_node *nd = &t->nodes[ni];
    if (nd->count > 0) {
        for (int32_t i = nd->first; i < nd->first + nd->count; ++i) {
            int32_t j = t->perm[i];
            double d = dist2(q, t->pts + 3 * (int64_t)j);
            if (d < *best || (d == *best && j < *bi)) { *best = d; *bi = j; }
        }
        return;
    }
    double val = q[nd->axis];
    double d1 = val - nd->divlow, d2 = val - nd->divhigh;
    int32_t near, far;
    double cut;
    if (d1 + d2 < 0) { near = nd->left; far = nd->right; cut = d2 * d2; }
    else { near = nd->right; far = nd->left; cut = d1 * d1; }
    kd_search(t, near, q, mind, dists, best, bi);
    double keep = dists[nd->axis];
    double mfar = mind + cut - keep;
    if (!(mfar * (1.0 - 1e-12) > *best)) {
        dists[nd->axis] = cut;
        kd_search(t, far, q, mfar, dists, best, bi);
        dists[nd->axis] = keep;
    }
}

static void kd_create(kd_t *t, const double *tgt, int64_t Nt) {
    t->nodes = (kd_node *)malloc(sizeof(kd_node) * (size_t)(2 * Nt + 2));
    t->perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)(Nt > 0 ? Nt : 1));
    t->n_nodes = 0;
    t->pts = tgt;
    for (int k = 0; k < 3; ++k) { t->lo[k] = INFINITY; t->hi[k] = -INFINITY; }
    for (int64_t i = 0; i < Nt; ++i) {
        t->perm[i] = (int32_t)i;
        for (int k = 0; k < 3; ++k) {
            if (tgt[3 * i + k] < t->lo[k]) t->lo[k] = tgt[3 * i + k];
            if (tgt[3 * i + k] > t->hi[k]) t->hi[k] = tgt[3 * i + k];
        }
    }
    kd_build_rec(t, 0, (int32_t)Nt);
}

static void kd_destroy(kd_t *t) {
    free(t->nodes);
    free(t->perm);
}

/* All queries against a built tree.  bound = INFINITY: plain 1-NN (KDTreeFlann::SearchHybrid's knn
 * step).  A finite bound (r^2) starts every search with that as the best distance, so cells that
 * cannot hold a point closer than r are never opened and a query without such a point returns
 * (-1, inf): the same correspondences as searching everything and applying d^2 < r^2 afterwards
 * (the acceptance test stays strict, ties inside the bound are never pruned). */
static void kd_query_all(const kd_t *t, const double *src, int64_t Ns, int32_t *idx, double *d2, double bound) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < Ns; ++i) {
        const double *q = src + 3 * i;
        double dists[3], mind = 0.0;
        for (int k = 0; k < 3; ++k) {
            double e = q[k] < t->lo[k] ? t->lo[k] - q[k] : (q[k] > t->hi[k] ? q[k] - t->hi[k] : 0.0);
            dists[k] = e * e;
            mind += dists[k];
        }
        double best = bound;
        int32_t bi = 0x7FFFFFFF;
        if (!(mind * (1.0 - 1e-12) > best)) kd_search(t, 0, q, mind, dists, &best, &bi);
        if (bi == 0x7FFFFFFF) { idx[i] = -1; d2[i] = INFINITY; }
        else { idx[i] = bi; d2[i] = best; }
    }
}

void pedp_oracle_nn_kdtree(const double *src, int64_t Ns, const double *tgt, int64_t Nt,
                           int32_t *idx, double *d2, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    if (Nt <= 0) {
        for (int64_t i = 0; i < Ns; ++i) { idx[i] = -1; d2[i] = INFINITY; }
        return;
    }
    kd_t t;
    kd_create(&t, tgt, Nt);
    kd_query_all(&t, src, Ns, idx, d2, INFINITY);
    kd_destroy(&t);
}

/* ------------------------------------------------------------------ registration_icp */

typedef struct {
    double fitness, rmse;
    int64_t K;
} pass_result;

/* GetRegistrationResultAndCorrespondences */
static pass_result corr_pass(const double *pcd, int64_t Ns, const double *tgt, int64_t Nt, double r,
                             int32_t *idx, double *d2, const kd_t *tree, int nthreads) {
    pass_result res = {0.0, 0.0, 0};
    if (r <= 0.0 || Ns == 0 || Nt == 0) {
        for (int64_t i = 0; i < Ns; ++i) idx[i] = -1;
        return res;
    }
    /* the tree is built once per registration (registration_icp: kdtree.SetGeometry(target)) */
    if (tree) kd_query_all(tree, pcd, Ns, idx, d2, r * r);
    else pedp_oracle_nn(pcd, Ns, tgt, Nt, idx, d2, nthreads);
    double r2 = r * r, err = 0.0;
    int64_t K = 0;
    for (int64_t i = 0; i < Ns; ++i) {
        if (d2[i] < r2) { err += d2[i]; ++K; }
        else idx[i] = -1;
    }
    res.K = K;
    if (K > 0) {
        res.fitness = (double)K / (double)Ns;
        res.rmse = sqrt(err / (double)K);
    }
    return res;
}

static void update_p2plane(const double *pcd, int64_t Ns, const double *tgt, const double *nrm,
                           const int32_t *idx, double upd[16]) {
    double A[36] = {0}, b[6] = {0};
    int64_t K = 0;
    for (int64_t i = 0; i < Ns; ++i) {
        if (idx[i] < 0) continue;
        const double *s = pcd + 3 * i, *t = tgt + 3 * (int64_t)idx[i], *n = nrm + 3 * (int64_t)idx[i];
        double r = ((s[0] - t[0]) * n[0] + (s[1] - t[1]) * n[1]) + (s[2] - t[2]) * n[2];
        double J[6] = {s[1] * n[2] - s[2] * n[1], s[2] * n[0] - s[0] * n[2], s[0] * n[1] - s[1] * n[0],
                       n[0], n[1], n[2]};
        for (int a = 0; a < 6; ++a) {
            for (int c = 0; c < 6; ++c) A[6 * a + c] += J[a] * J[c];
            b[a] += J[a] * r;
        }
        ++K;
    }
    mat4_identity(upd);
    if (K == 0) return; /* ComputeTransformation: empty correspondence set -> identity */
    double nb[6], x[6];
    for (int a = 0; a < 6; ++a) nb[a] = -b[a];
    if (pedp_oracle_solve6_ldlt(A, nb, x)) pedp_oracle_vec6_to_T(x, upd);
}

static void update_p2point(const double *pcd, int64_t Ns, const double *tgt, const int32_t *idx,
                           double upd[16]) {
    int64_t K = 0;
    for (int64_t i = 0; i < Ns; ++i) K += (idx[i] >= 0);
    mat4_identity(upd);
    if (K == 0) return;
    double *S = (double *)malloc(sizeof(double) * 3 * (size_t)K);
    double *T = (double *)malloc(sizeof(double) * 3 * (size_t)K);
    int64_t k = 0;
    for (int64_t i = 0; i < Ns; ++i) {
        if (idx[i] < 0) continue;
        memcpy(S + 3 * k, pcd + 3 * i, 24);
        memcpy(T + 3 * k, tgt + 3 * (int64_t)idx[i], 24);
        ++k;
    }
    pedp_oracle_kabsch(S, T, K, upd);
    free(S);
    free(T);
}

int pedp_oracle_icp(const double *src, int64_t Ns, const double *tgt, const double *tgt_normals,
                    int64_t Nt, double max_corr_dist, const double init[16], int estimator,
                    int max_iter, double rel_fitness, double rel_rmse, double T_out[16],
                    double *fitness, double *inlier_rmse, int32_t *n_iter_done, int32_t *corr,
                    double *trace, int use_kdtree, int nthreads) {
    if (Ns < 0 || Nt < 0 || max_iter < 0) return -1;
    if (estimator == PEDP_ORACLE_P2PLANE && !tgt_normals) return -2;
    if (estimator != PEDP_ORACLE_P2PLANE && estimator != PEDP_ORACLE_P2POINT) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    double T[16];
    memcpy(T, init, sizeof(T));
    double *pcd = (double *)malloc(sizeof(double) * 3 * (size_t)(Ns ? Ns : 1));
    double *d2 = (double *)malloc(sizeof(double) * (size_t)(Ns ? Ns : 1));
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(Ns ? Ns : 1));
    if (!pcd || !d2 || !idx) { free(pcd); free(d2); free(idx); return -3; }
    int is_identity = 1;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (fabs(init[4 * i + j] - (i == j ? 1.0 : 0.0)) > 1e-12) is_identity = 0;
    if (is_identity) memcpy(pcd, src, sizeof(double) * 3 * (size_t)Ns);
    else pedp_oracle_transform(init, src, Ns, pcd);

    kd_t tree_s;
    const kd_t *tree = NULL;
    if (use_kdtree && Nt > 0) { kd_create(&tree_s, tgt, Nt); tree = &tree_s; }
    pass_result res = corr_pass(pcd, Ns, tgt, Nt, max_corr_dist, idx, d2, tree, nthreads);
    if (trace) { trace[0] = res.fitness; trace[1] = res.rmse; memcpy(trace + 2, T, sizeof(T)); }
    int it = 0;
    for (; it < max_iter; ++it) {
        double upd[16];
        if (estimator == PEDP_ORACLE_P2PLANE) update_p2plane(pcd, Ns, tgt, tgt_normals, idx, upd);
        else update_p2point(pcd, Ns, tgt, idx, upd);
        mat4_mul(upd, T, T);
        pedp_oracle_transform(upd, pcd, Ns, pcd);
        pass_result prev = res;
        res = corr_pass(pcd, Ns, tgt, Nt, max_corr_dist, idx, d2, tree, nthreads);
        if (trace) {
            double *tr = trace + 18 * (it + 1);
            tr[0] = res.fitness; tr[1] = res.rmse; memcpy(tr + 2, T, sizeof(T));
        }
        if (fabs(prev.fitness - res.fitness) < rel_fitness && fabs(prev.rmse - res.rmse) < rel_rmse) {
            ++it;
            break;
        }
    }
    memcpy(T_out, T, sizeof(T));
    if (fitness) *fitness = res.fitness;
    if (inlier_rmse) *inlier_rmse = res.rmse;
    if (n_iter_done) *n_iter_done = it;
    if (corr) memcpy(corr, idx, sizeof(int32_t) * (size_t)Ns);
    if (tree) kd_destroy(&tree_s);
    free(pcd); free(d2); free(idx);
    return 0;
}
