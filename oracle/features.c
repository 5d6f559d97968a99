/* TEST INFRASTRUCTURE -- CPU restatement of the feature-based global registration the reference
 * runs in front of its ICP when `determine_pose(..., icp=True)` (src/pose_estimation.py:686-747):
 *
 *   pedp_oracle_fpfh             o3d.pipelines.registration.compute_fpfh_feature   (:132-137, :175-180, :255-260)
 *   pedp_oracle_feature_match    the nearest-feature correspondences inside
 *                                registration_ransac_based_on_feature_matching       (:482-501)
 *   pedp_oracle_ransac_hypothesis  one RANSAC draw of that function: three correspondences,
 *                                Umeyama without scaling, the three correspondence checkers
 *
 * Written from the published open3d==0.18.0 algorithms (Feature.cpp, Registration.cpp,
 * CorrespondenceChecker.cpp); the wheel is absent: parity unpinned.  Open3D draws its samples from
 * per-thread std::mt19937 engines seeded by random_device and walks the iterations under OpenMP; the
 * draw here is a counter-based function of (seed, iteration) -- the same splitmix64 as the plane
 * RANSAC -- and the definition the HIP side is tested against.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "pedp_oracle.h"

static double sqd(const double *a, const double *b) {
    const double x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
    return (x * x + y * y) + z * z;
}
static double dot(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static void cross(const double *a, const double *b, double *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* Feature.cpp ComputePairFeatures: (theta, alpha, phi, distance) of the Darboux frame at p1 */
static void pair_features(const double *p1, const double *n1, const double *p2, const double *n2, double r[4]) {
    double dp[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    r[0] = r[1] = r[2] = 0.0;
    r[3] = sqrt(dot(dp, dp));
    if (r[3] == 0.0) { r[3] = 0.0; return; }
    double a[3] = {n1[0], n1[1], n1[2]}, b[3] = {n2[0], n2[1], n2[2]};
    const double angle1 = dot(a, dp) / r[3], angle2 = dot(b, dp) / r[3];
    if (acos(fabs(angle1)) > acos(fabs(angle2))) {
        for (int k = 0; k < 3; ++k) { a[k] = n2[k]; b[k] = n1[k]; dp[k] = -dp[k]; }
        r[2] = -angle2;
    } else {
        r[2] = angle1;
    }
    double v[3], w[3];
    cross(dp, a, v);
    const double vn = sqrt(dot(v, v));
    if (vn == 0.0) { r[0] = r[1] = r[2] = r[3] = 0.0; return; }
    for (int k = 0; k < 3; ++k) v[k] /= vn;
    cross(a, v, w);
    r[1] = dot(v, b);
    r[0] = atan2(dot(w, b), dot(a, b));
}

static int bin11(double x) {
    int h = (int)floor(x);
    if (h < 0) h = 0;
    if (h >= 11) h = 10;
    return h;
}

typedef struct { double d; int64_t j; } cand;
static int cmp_cand(const void *a, const void *b) {
    const cand *x = (const cand *)a, *y = (const cand *)b;
    if (x->d != y->d) return x->d < y->d ? -1 : 1;
    return x->j < y->j ? -1 : (x->j > y->j);
}

/* KDTreeSearchParamHybrid(radius, max_nn): the max_nn nearest of the points closer than radius
 * (strictly, like nanoflann), nearest first; ties by index.  The first entry is "the point itself". */
static int64_t hybrid(const double *pts, int64_t N, int64_t i, double r2, int max_nn, cand *c) {
    int64_t n = 0;
    for (int64_t j = 0; j < N; ++j) {
        const double d = sqd(pts + 3 * i, pts + 3 * j);
        if (d < r2) { c[n].d = d; c[n].j = j; ++n; }
    }
    qsort(c, (size_t)n, sizeof(cand), cmp_cand);
    return n > max_nn ? max_nn : n;
}

/* out: N x 33 (the transpose of Open3D's Feature::data_) */
void pedp_oracle_fpfh(const double *pts, const double *nrm, int64_t N, double radius, int max_nn, double *out) {
    const double r2 = radius * radius;
    double *spfh = (double *)calloc((size_t)(N > 0 ? N : 1) * 33, sizeof(double));
    memset(out, 0, sizeof(double) * 33 * (size_t)N);
#pragma omp parallel
    {
        cand *c = (cand *)malloc(sizeof(cand) * (size_t)(N > 0 ? N : 1));
#pragma omp for schedule(dynamic, 32)
        for (int64_t i = 0; i < N; ++i) {  /* ComputeSPFHFeature */
            const int64_t n = hybrid(pts, N, i, r2, max_nn, c);
            if (n > 1) {
                const double incr = 100.0 / (double)(n - 1);
                double *h = spfh + 33 * i;
                for (int64_t k = 1; k < n; ++k) {
                    double pf[4];
                    pair_features(pts + 3 * i, nrm + 3 * i, pts + 3 * c[k].j, nrm + 3 * c[k].j, pf);
                    h[bin11(11.0 * (pf[0] + M_PI) / (2.0 * M_PI))] += incr;
                    h[11 + bin11(11.0 * (pf[1] + 1.0) * 0.5)] += incr;
                    h[22 + bin11(11.0 * (pf[2] + 1.0) * 0.5)] += incr;
                }
            }
        }
#pragma omp for schedule(dynamic, 32)
        for (int64_t i = 0; i < N; ++i) {  /* ComputeFPFHFeature: neighbours' SPFH weighted by 1 / squared distance */
            const int64_t n = hybrid(pts, N, i, r2, max_nn, c);
            if (n > 1) {
                double sum[3] = {0.0, 0.0, 0.0};
                double *f = out + 33 * i;
                for (int64_t k = 1; k < n; ++k) {
                    const double dist = c[k].d;
                    if (dist == 0.0) continue;
                    for (int j = 0; j < 33; ++j) {
                        const double val = spfh[33 * c[k].j + j] / dist;
                        sum[j / 11] += val;
                        f[j] += val;
                    }
                }
                for (int j = 0; j < 3; ++j)
                    if (sum[j] != 0.0) sum[j] = 100.0 / sum[j];
                for (int j = 0; j < 33; ++j) {
                    f[j] *= sum[j / 11];
                    f[j] += spfh[33 * i + j];
                }
            }
        }
        free(c);
    }
    free(spfh);
}

/* nearest target feature of every source feature (squared L2 over the 33 components summed in
 * order; ties: the lower index), what KDTreeFlann::SearchKNN(feature, 1) returns */
void pedp_oracle_feature_match(const double *fs, int64_t Ns, const double *ft, int64_t Nt, int32_t *idx) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < Ns; ++i) {
        double best = INFINITY;
        int32_t bj = -1;
        for (int64_t j = 0; j < Nt; ++j) {
            double d = 0.0;
            for (int k = 0; k < 33; ++k) {
                const double e = fs[33 * i + k] - ft[33 * j + k];
                d += e * e;
            }
            if (d < best) { best = d; bj = (int32_t)j; }
        }
        idx[i] = bj;
    }
}

static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* One draw of RegistrationRANSACBasedOnCorrespondence (ransac_n = 3): correspondences
 * corres[rand()] three times WITH replacement (Registration.cpp), Umeyama without scaling, then
 * CorrespondenceCheckerBasedOnEdgeLength(edge), ...BasedOnDistance(dist), ...BasedOnNormal(angle) in
 * the reference's order (src/pose_estimation.py:487-497).  corr[i] = index of source point i's
 * target point.  Returns 1 and T (source -> target) when every checker passes, else 0. */
int pedp_oracle_ransac_hypothesis(uint64_t seed, int64_t itr, const double *src, const double *src_nrm, int64_t Ns,
                                  const double *tgt, const double *tgt_nrm, const int32_t *corr, double edge,
                                  double dist, double angle, double T[16]) {
    uint64_t s = splitmix64(seed ^ splitmix64((uint64_t)itr));
    int64_t pick[3];
    double S[9], G[9];
    for (int k = 0; k < 3; ++k) {
        s = splitmix64(s);
        pick[k] = (int64_t)(s % (uint64_t)Ns);
        memcpy(S + 3 * k, src + 3 * pick[k], sizeof(double) * 3);
        memcpy(G + 3 * k, tgt + 3 * (int64_t)corr[pick[k]], sizeof(double) * 3);
    }
    pedp_oracle_kabsch(S, G, 3, T);
    /* edge length: every pair of the draw, both ways */
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j) {
            const double ds = sqrt(sqd(S + 3 * i, S + 3 * j)), dt = sqrt(sqd(G + 3 * i, G + 3 * j));
            if (ds < dt * edge || dt < ds * edge) return 0;
        }
    /* distance of the transformed source points to their target points */
    for (int k = 0; k < 3; ++k) {
        double p[3];
        for (int a = 0; a < 3; ++a)
            p[a] = ((T[4 * a] * S[3 * k] + T[4 * a + 1] * S[3 * k + 1]) + T[4 * a + 2] * S[3 * k + 2]) + T[4 * a + 3];
        if (sqrt(sqd(p, G + 3 * k)) > dist) return 0;
    }
    /* normals (skipped when a cloud has none, as Open3D warns and passes) */
    if (src_nrm && tgt_nrm) {
        const double cos_thr = cos(angle);
        for (int k = 0; k < 3; ++k) {
            const double *n = src_nrm + 3 * pick[k];
            double rn[3];
            for (int a = 0; a < 3; ++a) rn[a] = (T[4 * a] * n[0] + T[4 * a + 1] * n[1]) + T[4 * a + 2] * n[2];
            if (dot(rn, tgt_nrm + 3 * (int64_t)corr[pick[k]]) < cos_thr) return 0;
        }
    }
    return 1;
}

/* EvaluateInlierCorrespondenceRatio: share of the correspondences closer than max_dist under T */
double pedp_oracle_corres_inlier_ratio(const double *src, int64_t Ns, const double *tgt, const int32_t *corr,
                                       const double T[16], double max_dist) {
    int64_t inl = 0;
    const double m2 = max_dist * max_dist;
    for (int64_t i = 0; i < Ns; ++i) {
        double p[3];
        for (int a = 0; a < 3; ++a)
            p[a] = ((T[4 * a] * src[3 * i] + T[4 * a + 1] * src[3 * i + 1]) + T[4 * a + 2] * src[3 * i + 2]) + T[4 * a + 3];
        if (sqd(p, tgt + 3 * (int64_t)corr[i]) < m2) ++inl;
    }
    return Ns > 0 ? (double)inl / (double)Ns : 0.0;
}
