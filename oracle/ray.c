/*
 * ray.c -- ORACLE (test infrastructure, see pedp_oracle.h): float32 closest-hit
 * ray casting of rays [o|d] against a triangle soup.
 *
 * Follows: src/defect_projection.py:245-264 (intersect_rays_with_mesh: rays cast to
 * Float32[N,6], RaycastingScene.cast_rays, t_hit == inf on miss) and SURVEY.md
 * Appendix A.2.  The ray/triangle arithmetic of the reference lives in Embree
 * (bundled in open3d==0.18.0, absent here): its traversal order and SIMD rcp are
 * not reproducible, so this file DEFINES the test: Moeller-Trumbore in the
 * division-deferred form Embree's own Moeller-Trumbore intersector publishes
 * (U,V,T scaled by |det|, sign of det folded in by XOR, one division per accepted
 * hit), every operation a single fp32 rounding in the order written below.
 * Compile with -ffp-contract=off: fmaf() appears exactly where an FMA is meant.
 *
 *   per triangle (once):  e1 = v1 - v0,  e2 = v2 - v0,  m = e2 x e1
 *       cross(a,b).x = fma(a.y, b.z, -(a.z*b.y)) ... (cyclic)
 *   per ray/triangle test, scalar-triple-product form of Moeller-Trumbore
 *       dot(a,b) = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
 *   det = d . m                      ( = e1 . (d x e2) )
 *   s   = o - v0
 *   un  = d . (e2 x s)               ( = s . (d x e2) )
 *   vn  = d . (s x e1)
 *   tn  = -(s . m)                   ( = e2 . (s x e1) )
 *   sg = signbit(det);  ad = |det|;  U = un^sg; V = vn^sg; T = tn^sg
 *   hit  <=>  det != 0  &&  U >= 0  &&  V >= 0  &&  (U+V) <= ad  &&  T >= 0
 *   t = |T / ad|,  u = U / ad,  v = V / ad          (IEEE division)
 * This form is chosen because everything that depends on the ray ORIGIN (s, e2 x s, s x e1,
 * s . m) is independent of the ray direction: for the reference's rays, which all start at
 * one camera centre (defect_projection.py:545, :248), a GPU kernel may evaluate those terms
 * once per triangle with the very same operations and be left with three dot products per
 * test -- bit-identical to this per-test evaluation.
 * tnear = 0 inclusive, tfar = +inf, no back-face culling, inclusive edges.
 * Closest hit; ties: smaller t, then smaller triangle index.
 */
#include "pedp_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline float dot3(const float a[3], const float b[3]) {
    return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
}
static inline void cross3(const float a[3], const float b[3], float c[3]) {
    c[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
    c[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
    c[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}

void pedp_oracle_tri_setup(const float *verts, int64_t V, const uint32_t *tris, int64_t F,
                           float *tri9) {
    (void)V;
    for (int64_t f = 0; f < F; ++f) {
        const float *a = verts + 3 * (int64_t)tris[3 * f + 0];
        const float *b = verts + 3 * (int64_t)tris[3 * f + 1];
        const float *c = verts + 3 * (int64_t)tris[3 * f + 2];
        float *r = tri9 + PEDP_ORACLE_TRI * f;
        for (int k = 0; k < 3; ++k) {
            r[k] = a[k];
            r[3 + k] = b[k] - a[k];
            r[6 + k] = c[k] - a[k];
        }
        cross3(r + 6, r + 3, r + 9); /* m = e2 x e1 */
    }
}

int pedp_oracle_mt_test(const float o[3], const float d[3], const float *tri, float *t,
                        float *u, float *v) {
    const float *v0 = tri, *e1 = tri + 3, *e2 = tri + 6, *m = tri + 9;
    float s[3], a[3], b[3];
    float det = dot3(d, m);
    s[0] = o[0] - v0[0];
    s[1] = o[1] - v0[1];
    s[2] = o[2] - v0[2];
    cross3(e2, s, a);
    float un = dot3(d, a);
    cross3(s, e1, b);
    float vn = dot3(d, b);
    float tn = -dot3(s, m);
    uint32_t sg = f2u(det) & 0x80000000u;
    float ad = u2f(f2u(det) & 0x7FFFFFFFu);
    float U = u2f(f2u(un) ^ sg), Vv = u2f(f2u(vn) ^ sg), T = u2f(f2u(tn) ^ sg);
    float W = U + Vv;
    if (!(det != 0.0f)) return 0;
    if (!(U >= 0.0f)) return 0;
    if (!(Vv >= 0.0f)) return 0;
    if (!(W <= ad)) return 0;
    if (!(T >= 0.0f)) return 0;
    *t = u2f(f2u(T / ad) & 0x7FFFFFFFu);
    *u = U / ad;
    *v = Vv / ad;
    return 1;
}

static inline void cast_one_brute(const float *tri9, int64_t F, const float *ray, float *t_hit,
                                  uint32_t *prim, float *uv) {
    float bt = INFINITY, bu = 0.f, bv = 0.f;
    uint32_t bi = 0xFFFFFFFFu;
    for (int64_t f = 0; f < F; ++f) {
        float t, u, v;
        if (pedp_oracle_mt_test(ray, ray + 3, tri9 + PEDP_ORACLE_TRI * f, &t, &u, &v)) {
            if (t < bt || (t == bt && (uint32_t)f < bi)) {
                bt = t; bi = (uint32_t)f; bu = u; bv = v;
            }
        }
    }
    *t_hit = bt;
    *prim = bi;
    if (uv) { uv[0] = bu; uv[1] = bv; }
}

int pedp_oracle_raycast(const float *tri9, int64_t F, const float *rays6, int64_t N, float *t_hit,
                        uint32_t *prim_id, float *uv, int nthreads) {
    if (N < 0 || F < 0) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < N; ++i)
        cast_one_brute(tri9, F, rays6 + 6 * i, t_hit + i, prim_id + i, uv ? uv + 2 * i : NULL);
    return 0;
}

/* Every (ray, triangle) pair the test accepts, not only the closest hit per ray: pairs[2 k] = ray, pairs[2 k + 1] =
 * triangle, in (ray, triangle) order; returns the count (pairs beyond `capacity` are counted, not stored).  The
 * margin of the GPU's triangle-driven ray stage is measured against this set (tests/test_ray_gpu.py). */
int64_t pedp_oracle_accepted_pairs(const float *tri9, int64_t F, const float *rays6, int64_t N, int32_t *pairs, int64_t capacity) {
    int64_t n = 0;
    for (int64_t i = 0; i < N; ++i)
        for (int64_t f = 0; f < F; ++f) {
            float t, u, v;
            if (pedp_oracle_mt_test(rays6 + 6 * i, rays6 + 6 * i + 3, tri9 + PEDP_ORACLE_TRI * f, &t, &u, &v)) {
                if (n < capacity) { pairs[2 * n] = (int32_t)i; pairs[2 * n + 1] = (int32_t)f; }
                ++n;
            }
        }
    return n;
}

/* ------------------------------------------------------------------ BVH baseline
 * Median-split BVH over triangle bounds, leaves of <= 4 triangles, ordered
 * stack traversal.  Leaves run the SAME pedp_oracle_mt_test and the same tie rule,
 * and boxes are inflated / slabs evaluated in double so culling is conservative:
 * the result equals the brute-force sweep (tests/test_oracle_ray.py checks that).
 */
typedef struct {
    float lo[3], hi[3];
    int32_t left;  /* internal: index of left child (right = left+1); leaf: first prim slot */
    int32_t count; /* 0 for internal nodes */
} bvh_node;

typedef struct {
    bvh_node *nodes;
    int32_t n_nodes;
    int32_t *prim; /* permutation of triangle ids */
    float *cent;   /* F x 3 centroids */
    float *blo, *bhi;
    int64_t n_prims;
} bvh_t;

static void tri_bounds(const float *r, float lo[3], float hi[3]) {
    for (int k = 0; k < 3; ++k) {
        float a = r[k], b = r[k] + r[3 + k], c = r[k] + r[6 + k];
        /* v1, v2 are re-derived from (v0,e1,e2) and may differ by an ulp from the
         * original vertices: pad relative + absolute. */
        float mn = fminf(a, fminf(b, c)), mx = fmaxf(a, fmaxf(b, c));
        float pad = 1e-5f * fmaxf(fabsf(mn), fabsf(mx)) + 1e-6f * (mx - mn) + 1e-30f;
        lo[k] = mn - pad;
        hi[k] = mx + pad;
    }
}

static int32_t bvh_build_rec(bvh_t *b, int32_t first, int32_t count, int32_t node_idx) {
    bvh_node *nd = &b->nodes[node_idx];
    for (int k = 0; k < 3; ++k) { nd->lo[k] = INFINITY; nd->hi[k] = -INFINITY; }
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int32_t i = first; i < first + count; ++i) {
        int32_t p = b->prim[i];
        for (int k = 0; k < 3; ++k) {
            nd->lo[k] = fminf(nd->lo[k], b->blo[3 * p + k]);
            nd->hi[k] = fmaxf(nd->hi[k], b->bhi[3 * p + k]);
            clo[k] = fminf(clo[k], b->cent[3 * p + k]);
            chi[k] = fmaxf(chi[k], b->cent[3 * p + k]);
        }
    }
    if (count <= 4) { nd->left = first; nd->count = count; return node_idx; }
    int ax = 0;
    if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
    if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
    float mid = 0.5f * (clo[ax] + chi[ax]);
    int32_t i = first, j = first + count - 1;
    while (i <= j) {
        if (b->cent[3 * b->prim[i] + ax] < mid) ++i;
        else { int32_t t = b->prim[i]; b->prim[i] = b->prim[j]; b->prim[j] = t; --j; }
    }
    int32_t nl = i - first;
    if (nl == 0 || nl == count) nl = count / 2; /* degenerate: split by order */
    int32_t l = b->n_nodes;
    b->n_nodes += 2;
    nd->left = l;
    nd->count = 0;
    bvh_build_rec(b, first, nl, l);
    bvh_build_rec(b, first + nl, count - nl, l + 1);
    return node_idx;
}

static int bvh_build(bvh_t *b, const float *tri9, int64_t F) {
    memset(b, 0, sizeof(*b));
    if (F == 0) return 0;
    b->nodes = (bvh_node *)malloc(sizeof(bvh_node) * (size_t)(2 * F + 1));
    b->prim = (int32_t *)malloc(sizeof(int32_t) * (size_t)F);
    b->cent = (float *)malloc(sizeof(float) * 3 * (size_t)F);
    b->blo = (float *)malloc(sizeof(float) * 3 * (size_t)F);
    b->bhi = (float *)malloc(sizeof(float) * 3 * (size_t)F);
    if (!b->nodes || !b->prim || !b->cent || !b->blo || !b->bhi) return -1;
    for (int64_t f = 0; f < F; ++f) {
        b->prim[f] = (int32_t)f;
        tri_bounds(tri9 + PEDP_ORACLE_TRI * f, b->blo + 3 * f, b->bhi + 3 * f);
        for (int k = 0; k < 3; ++k) b->cent[3 * f + k] = 0.5f * (b->blo[3 * f + k] + b->bhi[3 * f + k]);
    }
    b->n_nodes = 1;
    b->n_prims = F;
    bvh_build_rec(b, 0, (int32_t)F, 0);
    return 0;
}

static void bvh_free(bvh_t *b) {
    free(b->nodes); free(b->prim); free(b->cent); free(b->blo); free(b->bhi);
}

/* conservative slab test in double; returns entry distance or +inf if missed */
static inline double slab(const bvh_node *nd, const double o[3], const double inv[3], double tmax) {
    double t0 = 0.0, t1 = tmax;
    for (int k = 0; k < 3; ++k) {
        double a = ((double)nd->lo[k] - o[k]) * inv[k];
        double c = ((double)nd->hi[k] - o[k]) * inv[k];
        if (a != a || c != c) continue; /* 0*inf: origin on a slab plane of a flat axis: do not cull */
        double near = a < c ? a : c, far = a < c ? c : a;
        if (near > t0) t0 = near;
        if (far < t1) t1 = far;
    }
    return (t0 <= t1 * (1.0 + 1e-9) + 1e-12) ? t0 : INFINITY;
}

static void cast_one_bvh(const bvh_t *b, const float *tri9, const float *ray, float *t_hit,
                         uint32_t *prim, float *uv) {
    float bt = INFINITY, bu = 0.f, bv = 0.f;
    uint32_t bi = 0xFFFFFFFFu;
    double o[3] = {ray[0], ray[1], ray[2]}, inv[3];
    for (int k = 0; k < 3; ++k) inv[k] = 1.0 / (double)ray[3 + k];
    int32_t stack[256];
    int sp = 0;
    if (b->n_nodes == 0) goto done;
    stack[sp++] = 0;
    while (sp > 0) {
        const bvh_node *nd = &b->nodes[stack[--sp]];
        /* allow ties at equal t (need the smaller id): cull only strictly beyond best */
        double lim = (bt == INFINITY) ? INFINITY : (double)bt * (1.0 + 1e-6) + 1e-9;
        double te = slab(nd, o, inv, lim);
        if (te == INFINITY) continue;
        if (nd->count > 0) {
            for (int32_t i = nd->left; i < nd->left + nd->count; ++i) {
                uint32_t f = (uint32_t)b->prim[i];
                float t, u, v;
                if (pedp_oracle_mt_test(ray, ray + 3, tri9 + PEDP_ORACLE_TRI * (int64_t)f, &t, &u, &v)) {
                    if (t < bt || (t == bt && f < bi)) { bt = t; bi = f; bu = u; bv = v; }
                }
            }
        } else {
            double tl = slab(&b->nodes[nd->left], o, inv, lim);
            double tr = slab(&b->nodes[nd->left + 1], o, inv, lim);
            if (sp + 2 > 256) { /* pathological depth: answer this ray by the plain sweep */
                cast_one_brute(tri9, b->n_prims, ray, t_hit, prim, uv);
                return;
            }
            if (tl <= tr) {
                if (tr != INFINITY) stack[sp++] = nd->left + 1;
                if (tl != INFINITY) stack[sp++] = nd->left;
            } else {
                if (tl != INFINITY) stack[sp++] = nd->left;
                if (tr != INFINITY) stack[sp++] = nd->left + 1;
            }
        }
    }
done:
    *t_hit = bt;
    *prim = bi;
    if (uv) { uv[0] = bu; uv[1] = bv; }
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int pedp_oracle_raycast_bvh(const float *tri9, int64_t F, const float *rays6, int64_t N,
                            float *t_hit, uint32_t *prim_id, float *uv, int nthreads,
                            double *build_seconds, double *cast_seconds) {
    if (N < 0 || F < 0 || F > 0x7FFFFFF0) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    bvh_t b;
    double t0 = now_s();
    if (bvh_build(&b, tri9, F) != 0) { bvh_free(&b); return -3; }
    double t1 = now_s();
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < N; ++i)
        cast_one_bvh(&b, tri9, rays6 + 6 * i, t_hit + i, prim_id + i, uv ? uv + 2 * i : NULL);
    double t2 = now_s();
    bvh_free(&b);
    if (build_seconds) *build_seconds = t1 - t0;
    if (cast_seconds) *cast_seconds = t2 - t1;
    return 0;
}
