// Ray / triangle closest-hit sweep for gfx950 (MI355X).
//
// Replaces RaycastingScene.add_triangles + cast_rays as called by
// src/defect_projection.py:245-256 (one camera ray per heat-map pixel against every
// triangle of the posed mesh).  Arithmetic contract: oracle/ray.c -- Moeller-Trumbore in
// scalar-triple-product form, division deferred to accepted hits, every operation one fp32
// rounding in a fixed order.  Results are bit-identical to it: t_hit, primitive ids, (u, v).
//
//   per triangle:  e1 = v1 - v0, e2 = v2 - v0, m = e2 x e1
//   per test:      det = d . m ; s = o - v0 ; un = d . (e2 x s) ; vn = d . (s x e1) ; tn = -(s . m)
//
// Variant 1, ray per lane (default).  One ray per lane, origin/direction in VGPRs for the
// whole sweep.  The triangle index is wave-uniform, so records are fetched with SCALAR loads
// and feed the VALU as SGPR operands (the scalar cache is the broadcast: no LDS traffic, no
// VGPRs for triangle data).  Two triangles ride in the two halves of every packed fp32
// instruction (v_pk_mul_f32 / v_pk_fma_f32 deliver 1.54x the flops per issue slot of the
// scalar forms on this chip, tools/microbench_valu.hip): records are stored
// pair-interleaved so one aligned SGPR pair = (A.x, B.x) = one packed operand.
//   * general origins: 12 x 2 floats per pair (v0, e1, e2, m), 24 arithmetic ops per test;
//   * all rays share one origin (always true for the reference, defect_projection.py:545):
//     the origin-dependent terms e2 x s, s x e1, s . m are evaluated once per triangle by
//     pair_shared_kernel with the very same operations, leaving 3 dot products per test
//     (10 x 2 floats per pair).  A device-side flag selects the path, no host round trip.
// The hot loop only decides "inside the triangle?" with a branch-free score
//     min3(un*det, vn*det, |det| - |un+vn|) >= 0
// which is a superset of the oracle's accept set (products keep the sign, underflow gives
// +-0 which passes); the division, tn, and the 64-bit min sit on a rarely taken wave-level
// branch that applies the oracle's exact predicate.
// Grid = (ray blocks) x (triangle chunks); chunk c is always served by workgroups with
// blockIdx % 8 == c % 8, i.e. one XCD, so each XCD's L2 only holds its part of the records.
// Partial results meet in one packed 64-bit atomicMin per ray and chunk:
// key = (bits(t) << 32) | triangle id orders by t (t >= 0: IEEE bits are monotone), then by
// triangle index -- the oracle's tie rule, independent of execution order.
//
// Variant 3 (default when all rays share one origin), ray per lane WITH conservative
// culling -- still every triangle is accounted for, most of them by a bound instead of a
// test.  Triangles are taken in clusters of 16 consecutive records with a bounding sphere and
// every 64 clusters in a super-cluster sphere (mesh build); per call every (super-)cluster gets
// its cone from the shared origin (unit axis v, cos/sin of the half-angle psi); rays are binned
// by direction (octahedral map, 256 x 256 cells in Hilbert-curve order, counting sort) so a
// packet of 64 consecutive rays covers one small solid angle, whose cone (axis a, half-angle
// theta) is reduced from the actual rays.  A ray of the packet can only hit a triangle of a
// cluster if angle(a, v) <= theta + psi, so clusters with v.a < cos(theta + psi) are dropped:
//   ray_cull_mask_kernel   one wave per packet, super-cluster cones first, then lane l tests
//                          cluster 64 w + l of every surviving word: the ballot IS the mask word
//   ray_segment_kernel     cuts every packet's survivor list into segments of equal length
//                          (device-side scan), so each sweep wave has the same work
//   ray_sweep_seg_kernel   one wave per segment: mask ranks -> LDS cluster list -> the loop body
//                          of variant 1 on those clusters
// Culling is conservative (margins below), so results stay bit-identical to the oracle.
//
// Variant 2, triangle per lane: few rays against a big mesh.  Lanes own consecutive
// triangles (coalesced 16-B loads of the AoS records), 4 wave-uniform rays per wave, and the
// packed key is min-reduced across the 64 lanes: the wavefront-wide min-t reduction.
#include "pedp_internal.h"
#include <vector>
#include <new>

namespace {

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

constexpr unsigned long long KEY_MISS = 0xFFFFFFFFFFFFFFFFull;
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float mulr(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fmar(float a, float b, float c) { return __fmaf_rn(a, b, c); }
__device__ __forceinline__ float subr(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float addr(float a, float b) { return __fadd_rn(a, b); }

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return fmar(az, bz, fmar(ay, by, mulr(ax, bx)));
}

struct Ray {
    float ox, oy, oz, dx, dy, dz;
};

struct MT {
    float det, un, vn, tn;
};

// The oracle's operation order (oracle/ray.c header comment), verbatim.  rec = 12 floats.
struct TriRec {
    float v0x, v0y, v0z, e1x, e1y, e1z, e2x, e2y, e2z, mx, my, mz;
};
__device__ __forceinline__ MT mt_eval(const Ray &r, const TriRec &q) {
    const float v0x = q.v0x, v0y = q.v0y, v0z = q.v0z, e1x = q.e1x, e1y = q.e1y, e1z = q.e1z;
    const float e2x = q.e2x, e2y = q.e2y, e2z = q.e2z, mx = q.mx, my = q.my, mz = q.mz;
    MT m;
    m.det = dot3(r.dx, r.dy, r.dz, mx, my, mz);
    const float sx = subr(r.ox, v0x), sy = subr(r.oy, v0y), sz = subr(r.oz, v0z);
    const float ax = fmar(e2y, sz, -mulr(e2z, sy));  // a = e2 x s
    const float ay = fmar(e2z, sx, -mulr(e2x, sz));
    const float az = fmar(e2x, sy, -mulr(e2y, sx));
    m.un = dot3(r.dx, r.dy, r.dz, ax, ay, az);
    const float bx = fmar(sy, e1z, -mulr(sz, e1y));  // b = s x e1
    const float by = fmar(sz, e1x, -mulr(sx, e1z));
    const float bz = fmar(sx, e1y, -mulr(sy, e1x));
    m.vn = dot3(r.dx, r.dy, r.dz, bx, by, bz);
    m.tn = -dot3(sx, sy, sz, mx, my, mz);
    return m;
}
__device__ __forceinline__ MT mt_eval(const Ray &r, const float *rec) {
    const TriRec q = {rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], rec[6], rec[7], rec[8], rec[9], rec[10], rec[11]};
    return mt_eval(r, q);
}

// Exact acceptance predicate of the oracle.
__device__ __forceinline__ bool mt_accept(const MT &m) {
    unsigned sg = __float_as_uint(m.det) & 0x80000000u;
    float U = __uint_as_float(__float_as_uint(m.un) ^ sg);
    float V = __uint_as_float(__float_as_uint(m.vn) ^ sg);
    float T = __uint_as_float(__float_as_uint(m.tn) ^ sg);
    float W = addr(U, V);
    return (m.det != 0.0f) & (U >= 0.0f) & (V >= 0.0f) & (T >= 0.0f) & (W <= fabsf(m.det));
}

__device__ __forceinline__ unsigned long long mt_key(const MT &m, unsigned id) {
    unsigned sg = __float_as_uint(m.det) & 0x80000000u;
    float T = __uint_as_float(__float_as_uint(m.tn) ^ sg);
    float t = __fdiv_rn(T, fabsf(m.det));
    unsigned tb = __float_as_uint(t) & 0x7FFFFFFFu;
    return ((unsigned long long)tb << 32) | (unsigned long long)id;
}

// ------------------------------------------------------------------ mesh setup
__global__ void tri_setup_kernel(const float *__restrict__ verts, const uint32_t *__restrict__ tris,
                                 int64_t F, int64_t F_padded, float *__restrict__ rec) {
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F_padded) return;
    float *r = rec + f * PEDP_TRI_STRIDE;
    if (f >= F) {
#pragma unroll
        for (int k = 0; k < PEDP_TRI_STRIDE; ++k) r[k] = 0.0f;  // det == 0: can never be hit
        return;
    }
    const float *a = verts + 3 * (int64_t)tris[3 * f + 0];
    const float *b = verts + 3 * (int64_t)tris[3 * f + 1];
    const float *c = verts + 3 * (int64_t)tris[3 * f + 2];
    float e1[3], e2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        e1[k] = subr(b[k], a[k]);
        e2[k] = subr(c[k], a[k]);
        r[k] = a[k];
        r[3 + k] = e1[k];
        r[6 + k] = e2[k];
    }
    r[9] = fmar(e2[1], e1[2], -mulr(e2[2], e1[1]));  // m = e2 x e1
    r[10] = fmar(e2[2], e1[0], -mulr(e2[0], e1[2]));
    r[11] = fmar(e2[0], e1[1], -mulr(e2[1], e1[0]));
}

// ------------------------------------------------------------------ pair-interleaved records
constexpr int PAIR_GEN = 24;  // floats per pair record, general origin: v0 e1 e2 m
constexpr int PAIR_SH = 20;   // floats per pair record, shared origin:  a' b' c' kappa (pair_shared_kernel)

__global__ void pair_general_kernel(const float *__restrict__ aos, int64_t F_padded, float *__restrict__ out) {
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F_padded) return;
    const float *r = aos + f * PEDP_TRI_STRIDE;
    float *o = out + (f >> 1) * PAIR_GEN + (f & 1);
#pragma unroll
    for (int k = 0; k < 12; ++k) o[2 * k] = r[k];
}

// Shared-origin pair records.  With one origin O for every ray, s = O - v0 and with it a = e2 x s, b = s x e1 and
// tn = -(s . m) are constants of the triangle (computed here in the oracle's own operations), and so is the SIDE of the
// triangle the rays come from: the oracle accepts only if T = tn ^ sign(det) >= 0, i.e. sign(det) = sign(tn) =: sigma.
// Under that orientation the accept test reads  d . (sigma a) >= 0,  d . (sigma b) >= 0,  U + V <= |det|;  the first two
// dot products are the oracle's un, vn up to the sign (negating a vector negates its rounded dot product exactly), the
// third is replaced by a bound that is never tighter:  d . c + kappa |d|_1 >= 0  with c = sigma (m - a - b) and
// kappa = 20 u (|m|_1 + |a|_1 + |b|_1): every rounding that separates fl(d . c) from fl(d . m) - fl(fl(d . a) + fl(d . b))
// (three dot products of three roundings each, one addition, two subtractions per component of c) is below half of
// that.  The hot loop thus needs THREE packed dot products, one packed fma and one min3 per triangle -- a superset of
// the oracle's accept set; the exact predicate runs on the rarely taken branch, from the AoS record.
//   record (20 floats per pair, the two triangles interleaved): a'[3] b'[3] c'[3] kappa
//   tn == 0 (the origin in the triangle's plane: either side) -> a record that always passes to the exact test;
//   m == 0 exactly (pad records, zero-area triangles: det = 0 for every ray) -> a record that never does.
__global__ void pair_shared_kernel(const float *__restrict__ aos, int64_t F_padded, const float *__restrict__ rays6,
                                   const int *__restrict__ shared_flag, float *__restrict__ out) {
    if (*shared_flag == 0) return;
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F_padded) return;
    const float *r = aos + f * PEDP_TRI_STRIDE;
    const float e1x = r[3], e1y = r[4], e1z = r[5], e2x = r[6], e2y = r[7], e2z = r[8];
    const float mx = r[9], my = r[10], mz = r[11];
    const float sx = subr(rays6[0], r[0]), sy = subr(rays6[1], r[1]), sz = subr(rays6[2], r[2]);
    const float ax = fmar(e2y, sz, -mulr(e2z, sy));
    const float ay = fmar(e2z, sx, -mulr(e2x, sz));
    const float az = fmar(e2x, sy, -mulr(e2y, sx));
    const float bx = fmar(sy, e1z, -mulr(sz, e1y));
    const float by = fmar(sz, e1x, -mulr(sx, e1z));
    const float bz = fmar(sx, e1y, -mulr(sy, e1x));
    const float tn = -dot3(sx, sy, sz, mx, my, mz);
    float v[10];
    if (mx == 0.0f && my == 0.0f && mz == 0.0f) {          // never accepted
        for (int k = 0; k < 9; ++k) v[k] = 0.0f;
        v[9] = -1.0f;
    } else if (!(tn > 0.0f) && !(tn < 0.0f)) {             // tn == 0 (or NaN): always to the exact test
        for (int k = 0; k < 9; ++k) v[k] = 0.0f;
        v[9] = 1e30f;
    } else {
        const float sg = tn < 0.0f ? -1.0f : 1.0f;
        v[0] = sg * ax; v[1] = sg * ay; v[2] = sg * az;
        v[3] = sg * bx; v[4] = sg * by; v[5] = sg * bz;
        v[6] = sg * subr(subr(mx, ax), bx); v[7] = sg * subr(subr(my, ay), by); v[8] = sg * subr(subr(mz, az), bz);
        v[9] = 1.2e-6f * ((fabsf(mx) + fabsf(my) + fabsf(mz)) + (fabsf(ax) + fabsf(ay) + fabsf(az)) + (fabsf(bx) + fabsf(by) + fabsf(bz))) + 1e-37f;
    }
    float *o = out + (f >> 1) * PAIR_SH + (f & 1);
#pragma unroll
    for (int k = 0; k < 10; ++k) o[2 * k] = v[k];
}

// all ray origins bit-identical to ray 0's?  flag preset to non-zero, cleared on a mismatch
__global__ void origin_check_kernel(const float *__restrict__ rays6, int64_t N, int *__restrict__ flag) {
    const unsigned ox = __float_as_uint(rays6[0]), oy = __float_as_uint(rays6[1]), oz = __float_as_uint(rays6[2]);
    bool same = true;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
        same = same && (__float_as_uint(rays6[6 * i]) == ox) && (__float_as_uint(rays6[6 * i + 1]) == oy) &&
               (__float_as_uint(rays6[6 * i + 2]) == oz);
    if (__builtin_amdgcn_ballot_w64(!same) != 0 && (threadIdx.x & 63) == 0) atomicAnd(flag, 0);
}

// ------------------------------------------------------------------ sweep, ray per lane (packed)
__device__ __forceinline__ f2 splat(float a) { return (f2){a, a}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 dot3p(f2 ax, f2 ay, f2 az, f2 bx, f2 by, f2 bz) {
    return fma2(az, bz, fma2(ay, by, ax * bx));
}

struct MT2 {  // det, un, vn of the two triangles of a pair
    f2 det, un, vn;
};

__device__ __forceinline__ MT2 eval_pair_general(const Ray &r, const f2 *t) {
    const f2 v0x = t[0], v0y = t[1], v0z = t[2], e1x = t[3], e1y = t[4], e1z = t[5];
    const f2 e2x = t[6], e2y = t[7], e2z = t[8], mx = t[9], my = t[10], mz = t[11];
    const f2 dx = splat(r.dx), dy = splat(r.dy), dz = splat(r.dz);
    MT2 m;
    m.det = dot3p(dx, dy, dz, mx, my, mz);
    const f2 sx = splat(r.ox) - v0x, sy = splat(r.oy) - v0y, sz = splat(r.oz) - v0z;
    const f2 ax = fma2(e2y, sz, -(e2z * sy));
    const f2 ay = fma2(e2z, sx, -(e2x * sz));
    const f2 az = fma2(e2x, sy, -(e2y * sx));
    m.un = dot3p(dx, dy, dz, ax, ay, az);
    const f2 bx = fma2(sy, e1z, -(sz * e1y));
    const f2 by = fma2(sz, e1x, -(sx * e1z));
    const f2 bz = fma2(sx, e1y, -(sy * e1x));
    m.vn = dot3p(dx, dy, dz, bx, by, bz);
    return m;
}

// shared origin: score of both halves from the oriented record, >= 0 is a superset of the oracle's accept set
// (pair_shared_kernel); dn = |d|_1 of the ray.  Ten packed and two scalar instructions per pair.
__device__ __forceinline__ f2 score_pair_shared(const Ray &r, float dn, const f2 *t) {
    const f2 dx = splat(r.dx), dy = splat(r.dy), dz = splat(r.dz);
    const f2 ua = dot3p(dx, dy, dz, t[0], t[1], t[2]);
    const f2 ub = dot3p(dx, dy, dz, t[3], t[4], t[5]);
    const f2 uc = fma2(t[9], splat(dn), dot3p(dx, dy, dz, t[6], t[7], t[8]));
    f2 s;
    s.x = fminf(fminf(ua.x, ub.x), uc.x);
    s.y = fminf(fminf(ua.y, ub.y), uc.y);
    return s;
}

// "inside the triangle" score of both halves; >= 0 is a superset of the oracle's accept set
__device__ __forceinline__ f2 inside_score(const MT2 &m) {
    const f2 a = m.un * m.det, b = m.vn * m.det, w = m.un + m.vn;
    f2 s;
    s.x = fminf(fminf(a.x, b.x), subr(fabsf(m.det.x), fabsf(w.x)));
    s.y = fminf(fminf(a.y, b.y), subr(fabsf(m.det.y), fabsf(w.y)));
    return s;
}

constexpr int RPL_BLOCK = 256;
constexpr int RPL_PAIRS = 2;  // pair records per loop iteration (4 triangles)

// the sweep loop: ray r against the 4-triangle groups [g0, g1)
template <bool SHARED>
__device__ __forceinline__ unsigned long long sweep_groups(const Ray &r, const f2 *__restrict__ rec, const float *__restrict__ aos,
                                                           int g0, int g1, unsigned long long best) {
    constexpr int PF = (SHARED ? PAIR_SH : PAIR_GEN) / 2;  // f2 per pair record
    const float dn = fabsf(r.dx) + fabsf(r.dy) + fabsf(r.dz);
    for (int g = g0; g < g1; ++g) {
        const f2 *t = rec + (size_t)g * (RPL_PAIRS * PF);
        f2 sc[RPL_PAIRS];
#pragma unroll
        for (int p = 0; p < RPL_PAIRS; ++p)
            sc[p] = SHARED ? score_pair_shared(r, dn, t + p * PF) : inside_score(eval_pair_general(r, t + p * PF));
        static_assert(RPL_PAIRS == 2, "score reduction below is written for 2 pairs");
        const float top = fmaxf(fmaxf(sc[0].x, sc[0].y), fmaxf(sc[1].x, sc[1].y));
        if (__builtin_amdgcn_ballot_w64(top >= 0.0f) != 0) {  // wave-uniform, rarely taken
            const int f0 = g * (2 * RPL_PAIRS);
#pragma unroll
            for (int k = 0; k < 2 * RPL_PAIRS; ++k) {
                const float s = (k & 1) ? sc[k >> 1].y : sc[k >> 1].x;
                if (s >= 0.0f) {
                    MT m = mt_eval(r, aos + (size_t)(f0 + k) * PEDP_TRI_STRIDE);  // exact, from the AoS record
                    if (mt_accept(m)) {
                        unsigned long long key = mt_key(m, (unsigned)(f0 + k));
                        best = key < best ? key : best;
                    }
                }
            }
        }
    }
    return best;
}

// The two instantiations are launched back to back; the device flag lets exactly one work.
template <bool SHARED>
__global__ __launch_bounds__(RPL_BLOCK) void ray_sweep_rpl_kernel(
    const f2 *__restrict__ rec, const float *__restrict__ aos, int groups_total, int groups_per_chunk, int n_chunks,
    const float *__restrict__ rays6, int64_t N, unsigned long long *__restrict__ keys,
    const int *__restrict__ shared_flag) {
    if ((*shared_flag != 0) != SHARED) return;
    const int b = blockIdx.x;
    const int chunk = b % n_chunks;  // n_chunks % 8 == 0: chunk % 8 == b % 8, one XCD per chunk
    const int64_t rb = b / n_chunks;
    const int64_t ray = rb * RPL_BLOCK + threadIdx.x;
    const int64_t rl = ray < N ? ray : N - 1;  // tail lanes re-run the last ray, never store
    Ray r;
    r.ox = rays6[6 * rl + 0]; r.oy = rays6[6 * rl + 1]; r.oz = rays6[6 * rl + 2];
    r.dx = rays6[6 * rl + 3]; r.dy = rays6[6 * rl + 4]; r.dz = rays6[6 * rl + 5];
    int g0 = chunk * groups_per_chunk;
    int g1 = g0 + groups_per_chunk;
    if (g1 > groups_total) g1 = groups_total;
    const unsigned long long best = sweep_groups<SHARED>(r, rec, aos, g0, g1, KEY_MISS);
    if (ray < N && best != KEY_MISS) atomicMin(&keys[ray], best);
}

// ------------------------------------------------------------------ exhaustive sweep on the matrix pipe (shared origin)
// The three scores of score_pair_shared -- d . a', d . b', d . c' + kappa |d|_1 -- are K = 3 dot products per (ray,
// triangle): the one contraction of this stage.  v_mfma_f32_16x16x32_bf16 takes it as a FILTER in front of the exact
// test (mt_eval / mt_accept / mt_key on the rare branch: not one result bit changes):
//   * every float splits EXACTLY into three bf16 pieces by truncation (hi = top 16 bits of x, mid = top 16 bits of x - hi,
//     lo = x - hi - mid: 3 x 8 mantissa bits, all of x's sign), so d_c x_c = sum over the nine piece products, each exact
//     in f32; one dot product = 27 products, laid along K:  k = 9 c + 3 p + q  <->  d_{c,p} * x_{c,q};
//   * K slot 27 carries the slack: A = |d|_1 rounded up, B = the triangle edge's bound rounded up (below);
//   * A (16 rays x 32) is built once per wave and ray group and stays in registers for the whole sweep; B (32 x 16
//     triangles of one edge) streams: one coalesced 1-KB load per fragment from a per-call record buffer laid out in
//     fragment order ([16-triangle group][edge][K quarter][triangle][8 bf16]);
//   * a wave owns 128 rays (8 A fragments): per 16 triangles 24 MFMAs (16 cycles each) and, per three of them, four
//     v_or3 and two v_max3_i32 (the sign test below) -- issued in the half of each MFMA's 16 cycles the vector pipe is free.
// Superset proof.  The oracle accepts only if fl(d . a') >= 0, fl(d . b') >= 0 and fl(d . c') + kappa |d|_1 >= 0
// (pair_shared_kernel), fl = the three-rounding fma chain: |fl(d . x) - d . x| <= 4 u S, S = sum_c |d_c| |x_c| <= |d|_1 |x|_inf,
// u = 2^-24.  The MFMA returns M = d . x + slack + delta: the pieces are exact, and its f32 accumulation of 29 terms --
// whatever its order and rounding mode, truncation included (unit 2^-23) -- errs by at most 28 * 2^-23 * (S + slack)
// < 2^-18 (S + slack).  With slack = (|d|_1 + 1e-18) * ((2^-17 + 2^-22) |x|_inf [+ kappa for c'] + 1e-18) the value M is >= 0 whenever the
// oracle's own score is: 2^-17 is more than twice what the accumulation can lose, 2^-22 = 4 u covers fl.  Measured
// (tests/test_ray_gpu.py::test_mfma_filter_*): no accepted pair of the oracle is ever rejected, and the filter's actual
// error stays below a hundredth of the slack.  Non-finite or astronomically large operands (> 1e15: their products
// could overflow inside the pipe) never reach it: such a ray or triangle gets zero pieces and a slack that always
// passes, and the exact test decides.
constexpr int MF_WAVES = 4;                // waves per workgroup; a wave owns MF_RG groups of 16 rays (template parameter)
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
union Frag { bf8 v; unsigned short h[8]; uint4 q; };

__device__ __forceinline__ void split3(float x, unsigned short out[3]) {  // x = hi + mid + lo exactly, bf16 each (truncation)
    const float hi = __uint_as_float(__float_as_uint(x) & 0xFFFF0000u);
    const float r1 = subr(x, hi);
    const float mid = __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
    const float lo = subr(r1, mid);
    out[0] = (unsigned short)(__float_as_uint(hi) >> 16);
    out[1] = (unsigned short)(__float_as_uint(mid) >> 16);
    out[2] = (unsigned short)(__float_as_uint(lo) >> 16);
}
__device__ __forceinline__ unsigned short bf16_up(float x) {  // smallest bf16 >= x, x >= 0 finite
    const unsigned b = __float_as_uint(x);
    return (unsigned short)((b >> 16) + ((b & 0xFFFFu) ? 1u : 0u));
}

// A pair passes the filter iff none of its three scores is negative.  That is asked of the SIGN BITS with integer
// instructions -- t = a | b | c is negative iff one of them is, and "some pair passed" is max over the t's >= 0 -- because
// fminf / fmaxf on values that come out of the matrix pipe cost a quieting v_max each in this strict-IEEE file, and an
// inline-asm v_min3 hides its reads of MFMA results from the compiler's hazard recogniser (the hardware does not
// interlock them: measured garbage).  A score of -0.0 would count as negative; the slack term is a strictly positive
// normal number (both of its factors are floored at 1e-18), so a sum is never -0.
__device__ __forceinline__ int sign3(float a, float b, float c) {
    return (int)(__float_as_uint(a) | __float_as_uint(b) | __float_as_uint(c));
}
__device__ __forceinline__ int imax3(int a, int b, int c) { return max(max(a, b), c); }

// per-call records of the shared-origin rays for the matrix sweep: a', b', c', kappa as pair_shared_kernel forms them,
// split and laid out in fragment order.  One thread per triangle.
__global__ void mfma_rec_kernel(const float *__restrict__ aos, int64_t F_padded, const float *__restrict__ rays6,
                                const int *__restrict__ shared_flag, unsigned short *__restrict__ out) {
    if (*shared_flag == 0) return;
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F_padded) return;
    const float *r = aos + f * PEDP_TRI_STRIDE;
    const float e1x = r[3], e1y = r[4], e1z = r[5], e2x = r[6], e2y = r[7], e2z = r[8];
    const float mx = r[9], my = r[10], mz = r[11];
    const float sx = subr(rays6[0], r[0]), sy = subr(rays6[1], r[1]), sz = subr(rays6[2], r[2]);
    const float ax = fmar(e2y, sz, -mulr(e2z, sy));
    const float ay = fmar(e2z, sx, -mulr(e2x, sz));
    const float az = fmar(e2x, sy, -mulr(e2y, sx));
    const float bx = fmar(sy, e1z, -mulr(sz, e1y));
    const float by = fmar(sz, e1x, -mulr(sx, e1z));
    const float bz = fmar(sx, e1y, -mulr(sy, e1x));
    const float tn = -dot3(sx, sy, sz, mx, my, mz);
    float v[9], slack[3];
    const float big = fmaxf(fmaxf(fmaxf(fabsf(ax), fabsf(ay)), fmaxf(fabsf(az), fabsf(bx))),
                            fmaxf(fmaxf(fabsf(by), fabsf(bz)), fmaxf(fmaxf(fabsf(mx), fabsf(my)), fabsf(mz))));
    for (int k = 0; k < 9; ++k) v[k] = 0.0f;
    bool never = false;
    if (mx == 0.0f && my == 0.0f && mz == 0.0f) {              // never accepted (pad records, zero-area triangles: det = 0 for every ray)
        never = true;
        slack[0] = slack[1] = slack[2] = 0.0f;
    } else if ((!(tn > 0.0f) && !(tn < 0.0f)) || !(big < 1e15f)) {   // tn == 0 / NaN, or operands the pipe must not see: always to the exact test
        slack[0] = slack[1] = slack[2] = 1e30f;
    } else {
        const float sg = tn < 0.0f ? -1.0f : 1.0f;
        v[0] = sg * ax; v[1] = sg * ay; v[2] = sg * az;
        v[3] = sg * bx; v[4] = sg * by; v[5] = sg * bz;
        v[6] = sg * subr(subr(mx, ax), bx); v[7] = sg * subr(subr(my, ay), by); v[8] = sg * subr(subr(mz, az), bz);
        const float kappa = 1.2e-6f * ((fabsf(mx) + fabsf(my) + fabsf(mz)) + (fabsf(ax) + fabsf(ay) + fabsf(az)) + (fabsf(bx) + fabsf(by) + fabsf(bz))) + 1e-37f;
        const float w = 7.9e-6f;                               // > 2^-17 + 2^-22
        slack[0] = w * fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fabsf(v[2])) * 1.0001f + 1e-18f;
        slack[1] = w * fmaxf(fmaxf(fabsf(v[3]), fabsf(v[4])), fabsf(v[5])) * 1.0001f + 1e-18f;
        slack[2] = (w * fmaxf(fmaxf(fabsf(v[6]), fabsf(v[7])), fabsf(v[8])) + kappa) * 1.0001f + 1e-18f;
    }
    const int64_t G = f >> 4, t = f & 15;
    for (int e = 0; e < 3; ++e) {
        unsigned short piece[3][3];
        for (int c = 0; c < 3; ++c) split3(v[3 * e + c], piece[c]);
        for (int k = 0; k < 32; ++k) {
            unsigned short val = 0;
            if (k < 27) val = piece[k / 9][k % 3];
            else if (k == 27) val = never ? (unsigned short)0xBF80 /* -1 times the ray's positive slot: the score is negative */ : bf16_up(slack[e]);
            out[((((G * 3 + e) * 4 + (k >> 3)) * 16 + t) << 3) + (k & 7)] = val;
        }
    }
}

// A fragment of ray (d) for the lane's K quarter h: k = 8 h + j
__device__ __forceinline__ bf8 mfma_ray_frag(float dx, float dy, float dz, int h) {
    const float dn = (fabsf(dx) + fabsf(dy)) + fabsf(dz);
    const bool ok = dn < 1e15f;                                 // (false for NaN / inf too)
    unsigned short piece[3][3];
    split3(ok ? dx : 0.0f, piece[0]);
    split3(ok ? dy : 0.0f, piece[1]);
    split3(ok ? dz : 0.0f, piece[2]);
    const unsigned short sl = ok ? bf16_up(dn * 1.0001f + 1e-18f) : (unsigned short)0x7149 /* 1e30 */;
    Frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        unsigned short val = 0;
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            const int k = 8 * hh + j;
            const unsigned short cand = k < 27 ? piece[k / 9][(k % 9) / 3] : (k == 27 ? sl : (unsigned short)0);
            val = h == hh ? cand : val;
        }
        f.h[j] = val;
    }
    return f.v;
}

__device__ __forceinline__ void mfma_exact(const float *__restrict__ rays6, const float *__restrict__ aos, int64_t ray, int64_t N, unsigned tri,
                                           unsigned long long *__restrict__ keys) {
    if (ray >= N) return;
    Ray r;
    r.ox = rays6[6 * ray + 0]; r.oy = rays6[6 * ray + 1]; r.oz = rays6[6 * ray + 2];
    r.dx = rays6[6 * ray + 3]; r.dy = rays6[6 * ray + 4]; r.dz = rays6[6 * ray + 5];
    const MT m = mt_eval(r, aos + (size_t)tri * PEDP_TRI_STRIDE);
    if (mt_accept(m)) atomicMin(&keys[ray], mt_key(m, tri));
}

template <int MF_RG>
__global__ __launch_bounds__(64 * MF_WAVES) void ray_sweep_mfma_kernel(
    const uint4 *__restrict__ rec, const float *__restrict__ aos, int tgroups_total, int tgroups_per_chunk, int n_chunks,
    const float *__restrict__ rays6, int64_t N, unsigned long long *__restrict__ keys, const int *__restrict__ shared_flag) {
    if (*shared_flag == 0) return;
    const int b = blockIdx.x;
    const int chunk = b % n_chunks;  // n_chunks % 8 == 0: one XCD per chunk
    const int64_t rb = b / n_chunks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 15, h = lane >> 4;
    const int64_t ray_base = (rb * MF_WAVES + wave) * (16 * MF_RG);
    if (ray_base >= N) return;
    bf8 A[MF_RG];
#pragma unroll
    for (int g = 0; g < MF_RG; ++g) {
        const int64_t ray = ray_base + 16 * g + row;
        const int64_t rl = ray < N ? ray : N - 1;  // tail rows repeat the last ray; their hits are dropped (ray >= N)
        A[g] = mfma_ray_frag(rays6[6 * rl + 3], rays6[6 * rl + 4], rays6[6 * rl + 5], h);
    }
    int g0 = chunk * tgroups_per_chunk, g1 = g0 + tgroups_per_chunk;
    if (g1 > tgroups_total) g1 = tgroups_total;
    if (g0 >= g1) return;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    const uint4 *p = rec + (size_t)g0 * 192 + lane;
    Frag B0, B1, B2, N0, N1, N2;
    B0.q = p[0]; B1.q = p[64]; B2.q = p[128];
    for (int tg = g0; tg < g1; ++tg) {
        const uint4 *pn = rec + (size_t)(tg + 1 < g1 ? tg + 1 : tg) * 192 + lane;   // the next group's fragments, in flight over this one's MFMAs
        N0.q = pn[0]; N1.q = pn[64]; N2.q = pn[128];
        int any = (int)0x80000000;   // max over the pairs' sign words: >= 0 iff some pair has no negative score
#pragma unroll
        for (int g = 0; g < MF_RG; ++g) {
            const f4 ra = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[g], B0.v, zero, 0, 0, 0);
            const f4 rb2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[g], B1.v, zero, 0, 0, 0);
            const f4 rc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[g], B2.v, zero, 0, 0, 0);
            any = imax3(any, sign3(ra[0], rb2[0], rc[0]), sign3(ra[1], rb2[1], rc[1]));
            any = imax3(any, sign3(ra[2], rb2[2], rc[2]), sign3(ra[3], rb2[3], rc[3]));
        }
        if (__builtin_amdgcn_ballot_w64(any >= 0) != 0) {   // wave-uniform, rarely taken: which pairs, and the exact test for them
#pragma unroll
            for (int g = 0; g < MF_RG; ++g) {
                const f4 ra = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[g], B0.v, zero, 0, 0, 0);
                const f4 rb2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[g], B1.v, zero, 0, 0, 0);
                const f4 rc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[g], B2.v, zero, 0, 0, 0);
                const int t0 = sign3(ra[0], rb2[0], rc[0]), t1 = sign3(ra[1], rb2[1], rc[1]), t2 = sign3(ra[2], rb2[2], rc[2]),
                          t3 = sign3(ra[3], rb2[3], rc[3]);
                if (__builtin_amdgcn_ballot_w64(imax3(t0, t1, max(t2, t3)) >= 0) == 0) continue;   // (no pair of this ray group)
                const int64_t r0 = ray_base + 16 * g + 4 * h;
                const unsigned tri = (unsigned)(tg * 16 + row);
                if (t0 >= 0) mfma_exact(rays6, aos, r0, N, tri, keys);
                if (t1 >= 0) mfma_exact(rays6, aos, r0 + 1, N, tri, keys);
                if (t2 >= 0) mfma_exact(rays6, aos, r0 + 2, N, tri, keys);
                if (t3 >= 0) mfma_exact(rays6, aos, r0 + 3, N, tri, keys);
            }
        }
        B0 = N0; B1 = N1; B2 = N2;
    }
}

// diagnostics (tests/test_ray_gpu.py::test_mfma_filter_*): the filter's score of EVERY (ray, triangle) pair and, beside it,
// the slack alone (the same MFMAs with every K slot of A but the slack's cleared): score - slack is what the pipe made
// of d . x', and how far below zero it falls on a pair the oracle accepts is the share of the slack that pair uses
__global__ __launch_bounds__(64) void mfma_debug_kernel(const uint4 *__restrict__ rec, int tgroups_total, const float *__restrict__ rays6,
                                                        int64_t N, int64_t F, float *__restrict__ score, float *__restrict__ slack) {
    const int lane = threadIdx.x & 63, row = lane & 15, h = lane >> 4;
    const int64_t ray_base = (int64_t)blockIdx.x * 16;
    const int64_t ray = ray_base + row, rl = ray < N ? ray : N - 1;
    Frag A, S;
    A.v = mfma_ray_frag(rays6[6 * rl + 3], rays6[6 * rl + 4], rays6[6 * rl + 5], h);
    S = A;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (!(h == 3 && j == 3)) S.h[j] = 0;   // K slot 27 = quarter 3, element 3
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int tg = 0; tg < tgroups_total; ++tg) {
        const uint4 *p = rec + (size_t)tg * 192 + lane;
        Frag B0, B1, B2;
        B0.q = p[0]; B1.q = p[64]; B2.q = p[128];
        const f4 ra = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.v, B0.v, zero, 0, 0, 0), rb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.v, B1.v, zero, 0, 0, 0),
                 rc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.v, B2.v, zero, 0, 0, 0);
        const f4 sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(S.v, B0.v, zero, 0, 0, 0), sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(S.v, B1.v, zero, 0, 0, 0),
                 sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(S.v, B2.v, zero, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t r = ray_base + 4 * h + i, t = (int64_t)tg * 16 + row;
            if (r < N && t < F) {
                // the edge with the smallest score decides; report that edge's slack
                float m = ra[i], sl = sa[i];
                if (rb[i] < m) { m = rb[i]; sl = sb[i]; }
                if (rc[i] < m) { m = rc[i]; sl = sc[i]; }
                if (sign3(ra[i], rb[i], rc[i]) < 0 && !(m < 0.0f)) m = -0.0f;   // (the kernel's rule: a sign bit set is a rejection)
                score[r * F + t] = m;
                slack[r * F + t] = sl;
            }
        }
    }
}

// ------------------------------------------------------------------ culled sweep (shared origin)
constexpr int CL_TRIS = 16;                              // triangles per cluster
constexpr int CL_GROUPS = CL_TRIS / (2 * RPL_PAIRS);     // loop groups per cluster (4)
constexpr int BIN_BITS = 8;                              // 256 x 256 direction cells
constexpr int BIN_CELLS = 1 << (2 * BIN_BITS);

// bounding sphere (center, radius) of each cluster's real triangles; radius < 0: empty
__global__ void cluster_sphere_kernel(const float *__restrict__ aos, int64_t F, int64_t n_clusters,
                                      float4 *__restrict__ sph) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_clusters) return;
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    int n = 0;
    for (int k = 0; k < CL_TRIS; ++k) {
        int64_t f = c * CL_TRIS + k;
        if (f >= F) break;
        const float *r = aos + f * PEDP_TRI_STRIDE;
        for (int a = 0; a < 3; ++a) {
            float p0 = r[a], p1 = r[a] + r[3 + a], p2 = r[a] + r[6 + a];
            lo[a] = fminf(lo[a], fminf(p0, fminf(p1, p2)));
            hi[a] = fmaxf(hi[a], fmaxf(p0, fmaxf(p1, p2)));
        }
        ++n;
    }
    if (n == 0) { sph[c] = make_float4(0.f, 0.f, 0.f, -1.f); return; }
    float cx = 0.5f * (lo[0] + hi[0]), cy = 0.5f * (lo[1] + hi[1]), cz = 0.5f * (lo[2] + hi[2]);
    float r2 = 0.f;
    for (int k = 0; k < n; ++k) {
        const float *r = aos + (c * CL_TRIS + k) * PEDP_TRI_STRIDE;
        for (int v = 0; v < 3; ++v) {
            float px = r[0] + (v ? r[3 * v] : 0.f) - cx, py = r[1] + (v ? r[3 * v + 1] : 0.f) - cy,
                  pz = r[2] + (v ? r[3 * v + 2] : 0.f) - cz;
            r2 = fmaxf(r2, px * px + py * py + pz * pz);
        }
    }
    // inflate: vertices re-derived from (v0, e1, e2) and fp32 accept decisions near the rim
    float rad = sqrtf(r2) * 1.001f + 1e-5f * (fabsf(cx) + fabsf(cy) + fabsf(cz)) + 1e-30f;
    sph[c] = make_float4(cx, cy, cz, rad);
}

// bounding sphere of 64 consecutive cluster spheres (one mask word of clusters): lets the cull
// kernel dismiss 1024 triangles with one test
// one wave per super-cluster, lane = cluster: min / max are order-free, so the result does not
// depend on the reduction shape
__global__ __launch_bounds__(256) void supercluster_sphere_kernel(const float4 *__restrict__ sph, int64_t n_clusters,
                                                                  int64_t n_super, float4 *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_super) return;  // wave-uniform
    const int64_t c = s * 64 + lane;
    const float4 q = c < n_clusters ? sph[c] : make_float4(0.f, 0.f, 0.f, -1.f);
    const bool live = q.w >= 0.f;
    float lo[3] = {live ? q.x - q.w : 3e38f, live ? q.y - q.w : 3e38f, live ? q.z - q.w : 3e38f};
    float hi[3] = {live ? q.x + q.w : -3e38f, live ? q.y + q.w : -3e38f, live ? q.z + q.w : -3e38f};
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
        }
    if (__builtin_amdgcn_ballot_w64(live) == 0ull) {
        if (lane == 0) out[s] = make_float4(0.f, 0.f, 0.f, -1.f);
        return;
    }
    const float cx = 0.5f * (lo[0] + hi[0]), cy = 0.5f * (lo[1] + hi[1]), cz = 0.5f * (lo[2] + hi[2]);
    const float dx = q.x - cx, dy = q.y - cy, dz = q.z - cz;
    float rad = live ? sqrtf(dx * dx + dy * dy + dz * dz) + q.w : 0.f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) rad = fmaxf(rad, __shfl_xor(rad, off, 64));
    if (lane == 0)
        out[s] = make_float4(cx, cy, cz, rad * 1.0001f + 1e-6f * (fabsf(cx) + fabsf(cy) + fabsf(cz)) + 1e-30f);
}

// per call: cone of each cluster seen from the shared origin.  rec[2c] = (vx, vy, vz, cos psi),
// rec[2c+1].x = sin psi.  cos psi = -2: the origin is inside the sphere (never cull);
// cos psi = 2: empty cluster (always cull).
__global__ void cluster_cone_kernel(const float4 *__restrict__ sph, int64_t n_clusters, const float *__restrict__ rays6,
                                    const int *__restrict__ shared_flag, float4 *__restrict__ rec) {
    if (*shared_flag == 0) return;
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_clusters) return;
    const float4 s = sph[c];
    float4 a = make_float4(0.f, 0.f, 1.f, -2.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s.w < 0.f) {
        a.w = 2.f;
    } else {
        float vx = s.x - rays6[0], vy = s.y - rays6[1], vz = s.z - rays6[2];
        float dist = sqrtf(vx * vx + vy * vy + vz * vz);
        if (dist > s.w * 1.0001f && dist > 0.f) {
            float inv = 1.0f / dist;
            float sn = fminf(s.w * inv * 1.0001f, 1.0f);
            a = make_float4(vx * inv, vy * inv, vz * inv, sqrtf(fmaxf(0.f, 1.0f - sn * sn)));
            b.x = sn;
        }
    }
    rec[2 * c] = a;
    rec[2 * c + 1] = b;
}

// ---- direction binning (counting sort by Hilbert-ordered cell of the octahedral map)
__device__ __forceinline__ void octa(float dx, float dy, float dz, float &u, float &v) {
    float n = fabsf(dx) + fabsf(dy) + fabsf(dz);
    float inv = n > 0.f ? 1.0f / n : 0.f;
    float px = dx * inv, py = dy * inv;
    if (dz < 0.f) {
        float qx = (1.0f - fabsf(py)) * (px >= 0.f ? 1.f : -1.f);
        float qy = (1.0f - fabsf(px)) * (py >= 0.f ? 1.f : -1.f);
        px = qx; py = qy;
    }
    u = px; v = py;
}
__device__ __forceinline__ unsigned enc_f(float f) {  // order-preserving float -> uint
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_f(unsigned e) {
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e);
}
// Hilbert curve index of cell (x, y) on the 2^BIN_BITS grid.  Consecutive indices are always
// neighbouring cells (no long jumps, unlike Morton order), so ANY run of 64 sorted rays -- one
// wave -- covers a compact patch of directions.
__device__ __forceinline__ unsigned hilbert_index(unsigned x, unsigned y) {
    const unsigned n = 1u << BIN_BITS;
    unsigned d = 0;
#pragma unroll
    for (unsigned s = n >> 1; s > 0; s >>= 1) {
        const unsigned rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d += s * s * ((3u * rx) ^ ry);
        if (ry == 0u) {
            if (rx == 1u) { x = n - 1u - x; y = n - 1u - y; }
            const unsigned t = x; x = y; y = t;
        }
    }
    return d;
}

// The direction order of a frame's rays is kept between calls.  A camera sends the same rays every
// frame, and the order only decides how COMPACT the 64-ray packets are: the culling takes every
// packet's cone from the rays the packet actually holds, so an order computed for other rays of
// the same count is merely less efficient, never wrong.  ray_order_check_kernel compares RAY_SAMPLES
// evenly spaced directions bit for bit with those the kept order was built from; only on a
// mismatch (`stale`) do the four binning kernels below run, otherwise they return at once.
constexpr int RAY_SAMPLES = 4096;
__global__ void ray_order_check_kernel(const float *__restrict__ rays6, int64_t N, float *__restrict__ samples,
                                       int *__restrict__ stale, unsigned *__restrict__ bounds) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s == 0) { bounds[0] = 0xFFFFFFFFu; bounds[1] = 0u; bounds[2] = 0xFFFFFFFFu; bounds[3] = 0u; }
    if (s >= RAY_SAMPLES) return;
    const int64_t i = (int64_t)s * N / RAY_SAMPLES;
    bool diff = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const unsigned now = __float_as_uint(rays6[6 * i + 3 + k]), was = __float_as_uint(samples[3 * s + k]);
        diff |= now != was;
        samples[3 * s + k] = __uint_as_float(now);
    }
    if (__builtin_amdgcn_ballot_w64(diff) != 0ull && (threadIdx.x & 63) == 0) atomicOr(stale, 1);
}

// bounds[0..3] = enc(min u), enc(max u), enc(min v), enc(max v); preset by ray_order_check_kernel.
// Also clears the histogram for ray_count_kernel (one word per thread and trip).
__global__ void ray_bounds_kernel(const float *__restrict__ rays6, int64_t N, const int *__restrict__ stale,
                                  unsigned *__restrict__ bounds, unsigned *__restrict__ hist) {
    if (*stale == 0) return;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < BIN_CELLS; k += gridDim.x * blockDim.x) hist[k] = 0u;
    unsigned lo_u = 0xFFFFFFFFu, hi_u = 0u, lo_v = 0xFFFFFFFFu, hi_v = 0u;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        float u, v;
        octa(rays6[6 * i + 3], rays6[6 * i + 4], rays6[6 * i + 5], u, v);
        if (u == u && v == v) {
            unsigned eu = enc_f(u), ev = enc_f(v);
            lo_u = eu < lo_u ? eu : lo_u; hi_u = eu > hi_u ? eu : hi_u;
            lo_v = ev < lo_v ? ev : lo_v; hi_v = ev > hi_v ? ev : hi_v;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned t;
        t = __shfl_xor(lo_u, off, 64); lo_u = t < lo_u ? t : lo_u;
        t = __shfl_xor(hi_u, off, 64); hi_u = t > hi_u ? t : hi_u;
        t = __shfl_xor(lo_v, off, 64); lo_v = t < lo_v ? t : lo_v;
        t = __shfl_xor(hi_v, off, 64); hi_v = t > hi_v ? t : hi_v;
    }
    __shared__ unsigned red[4][4];  // [wave][which]: same-address atomics are slow, one set per block
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = lo_u; red[wave][1] = hi_u; red[wave][2] = lo_v; red[wave][3] = hi_v; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            lo_u = red[w][0] < lo_u ? red[w][0] : lo_u; hi_u = red[w][1] > hi_u ? red[w][1] : hi_u;
            lo_v = red[w][2] < lo_v ? red[w][2] : lo_v; hi_v = red[w][3] > hi_v ? red[w][3] : hi_v;
        }
        atomicMin(&bounds[0], lo_u); atomicMax(&bounds[1], hi_u);
        atomicMin(&bounds[2], lo_v); atomicMax(&bounds[3], hi_v);
    }
}

__device__ __forceinline__ unsigned ray_cell(const float *__restrict__ rays6, int64_t i, const unsigned *__restrict__ bounds) {
    float u, v;
    octa(rays6[6 * i + 3], rays6[6 * i + 4], rays6[6 * i + 5], u, v);
    const float u0 = dec_f(bounds[0]), u1 = dec_f(bounds[1]), v0 = dec_f(bounds[2]), v1 = dec_f(bounds[3]);
    const float su = u1 > u0 ? (float)(1 << BIN_BITS) / (u1 - u0) : 0.f, sv = v1 > v0 ? (float)(1 << BIN_BITS) / (v1 - v0) : 0.f;
    int qu = (int)((u - u0) * su), qv = (int)((v - v0) * sv);
    qu = qu < 0 ? 0 : (qu > (1 << BIN_BITS) - 1 ? (1 << BIN_BITS) - 1 : qu);
    qv = qv < 0 ? 0 : (qv > (1 << BIN_BITS) - 1 ? (1 << BIN_BITS) - 1 : qv);
    if (!(u == u) || !(v == v)) { qu = 0; qv = 0; }
    return hilbert_index((unsigned)qu, (unsigned)qv);
}

__global__ void ray_count_kernel(const float *__restrict__ rays6, int64_t N, const int *__restrict__ stale,
                                 const unsigned *__restrict__ bounds, unsigned *__restrict__ hist) {
    if (*stale == 0) return;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) atomicAdd(&hist[ray_cell(rays6, i, bounds)], 1u);
}

// exclusive scan of BIN_CELLS counters, one workgroup of 1024 threads (64 cells each)
__global__ __launch_bounds__(1024) void bin_scan_kernel(const int *__restrict__ stale, unsigned *__restrict__ hist) {
    if (*stale == 0) return;
    __shared__ unsigned part[1024];
    constexpr int PER = BIN_CELLS / 1024;
    unsigned loc[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { loc[k] = hist[threadIdx.x * PER + k]; sum += loc[k]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        unsigned t = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    unsigned run = part[threadIdx.x] - sum;
#pragma unroll
    for (int k = 0; k < PER; ++k) { hist[threadIdx.x * PER + k] = run; run += loc[k]; }
}

__global__ void ray_scatter_kernel(const float *__restrict__ rays6, int64_t N, const int *__restrict__ stale,
                                   const unsigned *__restrict__ bounds, unsigned *__restrict__ cursor /* scanned */,
                                   unsigned *__restrict__ perm) {
    if (*stale == 0) return;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) perm[atomicAdd(&cursor[ray_cell(rays6, i, bounds)], 1u)] = (unsigned)i;
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}

// ---- culled sweep in three balanced steps ------------------------------------------------
// 1. ray_cull_mask_kernel: one wave per packet (64 direction-sorted rays): cone of the packet,
//    lane-parallel cluster tests, the ballot of 64 tests IS the mask word of surviving clusters.
// 2. ray_segment_kernel: cuts every packet's survivor list into segments of seg_len clusters
//    (>= RSEG_MIN, chosen on the device so all fit the table).  Packets that look along a
//    surface see hundreds of clusters, most see none: without this step a few waves carried
//    the kernel (0.52 ms); with it every sweep wave has the same amount of work.
// 3. ray_sweep_seg_kernel: one wave per segment, ray per lane, the loop body of the exhaustive
//    kernel on the segment's clusters; packets meet in the same 64-bit atomicMin keys.
constexpr int RSEG_MIN = 16;    // clusters per segment at least (256 triangles)
constexpr int RLIST = 2048;     // most clusters one sweep wave walks (LDS list)

__global__ __launch_bounds__(256) void ray_cull_mask_kernel(const float4 *__restrict__ cones,
                                                            const float4 *__restrict__ scones, int n_clusters,
                                                            int n_words, const float *__restrict__ rays6,
                                                            const unsigned *__restrict__ perm, int64_t N,
                                                            unsigned long long *__restrict__ mask,
                                                            int *__restrict__ pk_cnt, const int *__restrict__ shared_flag) {
    if (*shared_flag == 0) return;
    const int lane = threadIdx.x & 63;
    const int64_t pk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pk * 64 >= N) return;
    const int64_t k = pk * 64 + lane;
    const int64_t ri = perm[k < N ? k : N - 1];
    const float dx = rays6[6 * ri + 3], dy = rays6[6 * ri + 4], dz = rays6[6 * ri + 5];
    // cone of this packet: axis = normalised sum of unit directions, cos(theta) = min dot
    const float inv = rsqrtf(dx * dx + dy * dy + dz * dz);
    const float ux = dx * inv, uy = dy * inv, uz = dz * inv;
    float ax = wave_sum_f(ux), ay = wave_sum_f(uy), az = wave_sum_f(uz);
    const float ainv = rsqrtf(ax * ax + ay * ay + az * az);
    ax *= ainv; ay *= ainv; az *= ainv;
    const float ct = wave_min_f(ux * ax + uy * ay + uz * az) - 1e-5f;  // margin: rsqrt + rounding
    // wide or degenerate (NaN) packets do not cull: every comparison below is then false
    const bool can_cull = ct > 0.1f;
    const float st = sqrtf(fmaxf(0.f, 1.0f - ct * ct)) + 1e-5f;
    int cnt = 0;
    // level 1: one test per super-cluster (= one mask word of 64 clusters); lane l takes word
    // sw + l.  level 2: the 64 clusters of every surviving word.
    for (int sw = 0; sw < n_words; sw += 64) {
        const int wl = sw + lane;
        bool live = false;
        if (wl < n_words) {
            const float4 ca = scones[2 * wl];
            const float sp = scones[2 * wl + 1].x;
            const float cosv = ca.x * ax + ca.y * ay + ca.z * az;
            const float lim = ct * ca.w - st * sp - 1e-5f;
            live = !((ca.w > 1.5f) || (can_cull && ca.w > -1.5f && cosv < lim));
            if (!live) mask[(size_t)pk * n_words + wl] = 0ull;
        }
        unsigned long long todo = __builtin_amdgcn_ballot_w64(live);
        while (todo != 0ull) {  // wave-uniform
            const int wi = sw + __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const int c = wi * 64 + lane;
            bool keep = false;
            if (c < n_clusters) {
                const float4 ca = cones[2 * c];
                const float sp = cones[2 * c + 1].x;
                const float cosv = ca.x * ax + ca.y * ay + ca.z * az;
                const float lim = ct * ca.w - st * sp - 1e-5f;  // cos(theta + psi), lowered by a margin
                const bool culled = (ca.w > 1.5f) || (can_cull && ca.w > -1.5f && cosv < lim);
                keep = !culled;
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
            if (lane == 0) mask[(size_t)pk * n_words + wi] = m;
            cnt += __builtin_popcountll(m);
        }
    }
    if (lane == 0) pk_cnt[pk] = cnt;
}

// seg_info[0] = number of segments, seg_info[1] = seg_len
__global__ __launch_bounds__(1024) void ray_segment_kernel(const int *__restrict__ pk_cnt, int n_packets,
                                                           int *__restrict__ seg_pk, int *__restrict__ seg_rank0,
                                                           int *__restrict__ seg_n, int max_segs, int *__restrict__ seg_info,
                                                           const int *__restrict__ shared_flag) {
    if (*shared_flag == 0) { if (threadIdx.x == 0) seg_info[0] = 0; return; }
    __shared__ long long red[16];
    __shared__ int scan[1024];
    __shared__ long long total_s;
    const int tid = threadIdx.x;
    const int per = (n_packets + 1023) / 1024;
    const int b0 = tid * per, b1 = (b0 + per < n_packets) ? b0 + per : n_packets;
    long long v = 0;
    for (int b = b0; b < b1; ++b) v += pk_cnt[b];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
        long long t = 0;
        for (int k = 0; k < 16; ++k) t += red[k];
        total_s = t;
    }
    __syncthreads();
    long long room = (long long)max_segs - n_packets;
    if (room < 1) room = 1;
    long long sl = (total_s + room - 1) / room;
    if (sl < RSEG_MIN) sl = RSEG_MIN;
    if (sl > RLIST) sl = RLIST;  // the host sizes max_segs so that this cannot bind
    const int seg_len = (int)sl;
    int mine = 0;
    for (int b = b0; b < b1; ++b) mine += (pk_cnt[b] + seg_len - 1) / seg_len;
    scan[tid] = mine;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int t = tid >= off ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += t;
        __syncthreads();
    }
    int at = scan[tid] - mine;
    for (int b = b0; b < b1; ++b) {
        const int c = pk_cnt[b];
        for (int r0 = 0; r0 < c; r0 += seg_len) {
            if (at < max_segs) { seg_pk[at] = b; seg_rank0[at] = r0; seg_n[at] = (c - r0 < seg_len) ? c - r0 : seg_len; }
            ++at;
        }
    }
    if (tid == 1023) { seg_info[0] = scan[1023] < max_segs ? scan[1023] : max_segs; seg_info[1] = seg_len; }
}

__global__ __launch_bounds__(RPL_BLOCK) void ray_sweep_seg_kernel(
    const f2 *__restrict__ rec, const float *__restrict__ aos, int n_words, const unsigned long long *__restrict__ mask,
    const int *__restrict__ seg_pk, const int *__restrict__ seg_rank0, const int *__restrict__ seg_n,
    const int *__restrict__ seg_info, const float *__restrict__ rays6, const unsigned *__restrict__ perm, int64_t N,
    unsigned long long *__restrict__ keys) {
    __shared__ unsigned surv[RPL_BLOCK / 64][RLIST];
    constexpr int PF = PAIR_SH / 2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int seg = blockIdx.x * (RPL_BLOCK / 64) + wv;
    if (seg >= seg_info[0]) return;  // wave-uniform
    const int pk = seg_pk[seg], r0 = seg_rank0[seg], n_s = seg_n[seg];
    const int64_t k = (int64_t)pk * 64 + lane;
    const int64_t ri = perm[k < N ? k : N - 1];  // tail lanes re-run the last ray, never store
    Ray r;
    r.ox = rays6[6 * ri + 0]; r.oy = rays6[6 * ri + 1]; r.oz = rays6[6 * ri + 2];
    r.dx = rays6[6 * ri + 3]; r.dy = rays6[6 * ri + 4]; r.dz = rays6[6 * ri + 5];
    // expand ranks [r0, r0 + n_s) of the packet's mask into the LDS cluster list
    unsigned *mine = surv[wv];
    {
        int running = 0;
        const unsigned long long *mw = mask + (size_t)pk * n_words;
        for (int wg = 0; wg < n_words && running < r0 + n_s; wg += 64) {
            unsigned long long word = (wg + lane < n_words) ? mw[wg + lane] : 0ull;
            const int pc = __builtin_popcountll(word);
            int incl = pc;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            int rank = running + incl - pc;
            if (pc > 0 && rank < r0 + n_s && rank + pc > r0) {
                const unsigned c0 = (unsigned)(wg + lane) * 64u;
                while (word != 0ull) {
                    const int bit = __builtin_ctzll(word);
                    word &= word - 1ull;
                    if (rank >= r0 && rank < r0 + n_s) mine[rank - r0] = c0 + (unsigned)bit;
                    ++rank;
                }
            }
            running += __shfl(incl, 63, 64);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes have landed
    unsigned long long best = KEY_MISS;
    const float dn_seg = fabsf(r.dx) + fabsf(r.dy) + fabsf(r.dz);
    for (int q = 0; q < n_s; ++q) {
        const int cl = (int)__builtin_amdgcn_readfirstlane(mine[q]);  // wave-uniform: scalar record loads below
#pragma unroll 1
        for (int gi = 0; gi < CL_GROUPS; ++gi) {
            const int g = cl * CL_GROUPS + gi;
            const f2 *t = rec + (size_t)g * (RPL_PAIRS * PF);
            f2 sc[RPL_PAIRS];
#pragma unroll
            for (int p = 0; p < RPL_PAIRS; ++p) sc[p] = score_pair_shared(r, dn_seg, t + p * PF);
            const float top = fmaxf(fmaxf(sc[0].x, sc[0].y), fmaxf(sc[1].x, sc[1].y));
            if (__builtin_amdgcn_ballot_w64(top >= 0.0f) != 0) {
                const int f0 = g * (2 * RPL_PAIRS);
#pragma unroll
                for (int e = 0; e < 2 * RPL_PAIRS; ++e) {
                    const float s = (e & 1) ? sc[e >> 1].y : sc[e >> 1].x;
                    if (s >= 0.0f) {
                        MT m = mt_eval(r, aos + (size_t)(f0 + e) * PEDP_TRI_STRIDE);
                        if (mt_accept(m)) {
                            unsigned long long key = mt_key(m, (unsigned)(f0 + e));
                            best = key < best ? key : best;
                        }
                    }
                }
            }
        }
    }
    if (k < N && best != KEY_MISS) atomicMin(&keys[ri], best);
}

// ------------------------------------------------------------------ sweep, triangle-driven (variant 4)
// All rays leave one origin O, so a triangle is seen under one fixed solid angle.  Take a frame
// (A, E1, E2) with A = the mean direction of eight rays spread over the array, map every ray to p = (d.E1, d.E2) / d.A (the
// plane at unit distance along A: straight lines stay straight, so a triangle in front of O maps
// to the triangle of its mapped corners) and lay a grid of about N cells over the rays' bounding
// rectangle.  Rays are threaded into per-cell chains (head[cell] -> node -> node ...; a node is the
// ray's direction + the next index, the node index IS the ray index, nothing is sorted or moved).
// Then the TRIANGLES drive the sweep: a thread per triangle maps the three corners, takes their
// bounding rectangle widened by a margin (an eighth of a cell + sixteen times the dilation the fp32
// test's own rounding can give the triangle) and applies the oracle's exact test to the rays in those cells --
// the same mt_eval / mt_accept / mt_key on the same (ray, triangle) operands as the exhaustive
// sweep, met in the same 64-bit atomicMin, so t, primitive ids and (u, v) are the same bits.
// Every ray x triangle pair is still accounted for: a ray can only hit a triangle whose mapped
// rectangle holds the ray's point.  Work per cast is O(N + F + covered cells) instead of O(N x F).
//
//   rast_bounds_kernel   origins equal? rays inside the frame's half space (d.A > 0.17 |d|)?
//                        bounds of p; clears keys and chain heads
//   rast_insert_kernel   grid from the bounds; node[i] = (d_i, atomicExch(head[cell_i], i))
//   rast_tri_kernel      triangle per thread: up to RAST_INLINE cells walked by the thread itself,
//                        larger rectangles cut into tiles of RAST_TILE x RAST_TILE cells
//   rast_item_kernel     a wave per item, a cell per lane
// Whatever the grid cannot answer -- origins differ, a ray outside the half space, more than
// RAST_CHAIN_MAX rays in one cell, a full item table -- clears hdr.ok on the device, and the last
// kernel of the cast (ray_finalize_kernel) then runs the general exhaustive loop for every ray before
// it writes the results; the status reaches the host through a pinned word and the next cast of the
// same ray count takes the cone culling of variant 3 instead.  A triangle that comes within eps of
// the plane through O perpendicular to A has no bounded image: it is tested against every cell.
constexpr unsigned RAST_NONE = 0xFFFFFFFFu;
constexpr int RAST_INLINE = 16;
constexpr int RAST_TILE = 16;          // larger rectangles are cut into tiles of 16 x 16 cells, one wave each
constexpr int RAST_ITEM_CAP = 1 << 18;
constexpr int RAST_CHAIN_MAX = 64;
constexpr float RAST_COS_MIN = 0.17f;
constexpr int RAST_ITEM_WAVES = 2048;
constexpr int RAST_FULL_CAP = 8192;     // triangles without a bounded image that one cast can list

struct RastHdr {  // two per context (used in turn), device memory; ray_finalize_kernel puts the other one back to its start values
    int ok;       // 1: the grid answers this cast; 0: the exhaustive loop in ray_finalize_kernel does
    int n_items;
    int reason;   // why ok was cleared: 1 origins differ, 2 ray outside the half space, 4 chain too long, 8 item table full
    int n_full;   // triangles without a bounded image (listed; rast_full_kernel tests them against every ray)
    float O[3], A[3], E1[3], E2[3];
    float u0, v0, su, sv;
    int GX, GY;
};
struct RastItem { int tri, x0, y0, w, h, pad0, pad1, pad2; };

struct RastFrame { float A[3], E1[3], E2[3]; bool valid; };

__device__ __forceinline__ RastFrame rast_frame(const float *__restrict__ rays6, int64_t N) {
    // A = the mean of eight rays spread over the array (k (N - 1) / 7: first, last and six between them;
    // for a row-major pixel grid they fall on different columns, so A points near the middle of the view)
    float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int64_t i = (N - 1) * k / 7;
        const float dx = rays6[6 * i + 3], dy = rays6[6 * i + 4], dz = rays6[6 * i + 5];
        const float l2 = dx * dx + dy * dy + dz * dz;
        if (l2 > 1e-30f && l2 < 1e30f) {
            const float inv = 1.0f / sqrtf(l2);
            sx += dx * inv; sy += dy * inv; sz += dz * inv;
        }
    }
    const float l2 = sx * sx + sy * sy + sz * sz;
    RastFrame f;
    f.valid = l2 > 1e-6f;
    const float inv = 1.0f / sqrtf(f.valid ? l2 : 1.0f);
    f.A[0] = sx * inv; f.A[1] = sy * inv; f.A[2] = sz * inv;
    const float ax = fabsf(f.A[0]), ay = fabsf(f.A[1]), az = fabsf(f.A[2]);
    float ex = 0.f, ey = 0.f, ez = 0.f;
    if (ax <= ay && ax <= az) ex = 1.f; else if (ay <= az) ey = 1.f; else ez = 1.f;
    float cx = ey * f.A[2] - ez * f.A[1], cy = ez * f.A[0] - ex * f.A[2], cz = ex * f.A[1] - ey * f.A[0];
    const float ci = 1.0f / sqrtf(fmaxf(cx * cx + cy * cy + cz * cz, 1e-30f));
    f.E1[0] = cx * ci; f.E1[1] = cy * ci; f.E1[2] = cz * ci;
    f.E2[0] = f.A[1] * f.E1[2] - f.A[2] * f.E1[1];
    f.E2[1] = f.A[2] * f.E1[0] - f.A[0] * f.E1[2];
    f.E2[2] = f.A[0] * f.E1[1] - f.A[1] * f.E1[0];
    return f;
}

// 0: no direction (zero or NaN: such a ray hits nothing in the exhaustive sweep either), 1: mapped, 2: outside the half space
__device__ __forceinline__ int rast_map(float dx, float dy, float dz, const float *A, const float *E1, const float *E2,
                                        float &u, float &v) {
    const float l2 = dx * dx + dy * dy + dz * dz;
    if (!(l2 > 0.f)) return 0;            // zero, or NaN
    if (!(l2 < 3e38f)) return 2;          // infinite components: leave them to the exhaustive sweep
    const float w = dx * A[0] + dy * A[1] + dz * A[2];
    if (!(w * fabsf(w) > RAST_COS_MIN * RAST_COS_MIN * l2)) return 2;
    const float iw = 1.0f / w;
    u = (dx * E1[0] + dy * E1[1] + dz * E1[2]) * iw;
    v = (dx * E2[0] + dy * E2[1] + dz * E2[2]) * iw;
    return 1;
}

constexpr int RAST_BBLOCKS = 256;   // workgroups of rast_bounds_kernel = partial bounds the insert kernel folds
constexpr int RAST_BUNROLL = 4;

__global__ __launch_bounds__(256) void rast_bounds_kernel(const float *__restrict__ rays6, int64_t N, RastHdr *__restrict__ h,
                                                         unsigned long long *__restrict__ keys, unsigned *__restrict__ head,
                                                         uint4 *__restrict__ part) {
    const RastFrame f = rast_frame(rays6, N);
    const unsigned ox = __float_as_uint(rays6[0]), oy = __float_as_uint(rays6[1]), oz = __float_as_uint(rays6[2]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { h->O[k] = rays6[k]; h->A[k] = f.A[k]; h->E1[k] = f.E1[k]; h->E2[k] = f.E2[k]; }
        if (!f.valid) { atomicAnd(&h->ok, 0); atomicOr(&h->reason, 2); }
    }
    unsigned lo_u = 0xFFFFFFFFu, hi_u = 0u, lo_v = 0xFFFFFFFFu, hi_v = 0u;
    int bad = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < N; i0 += RAST_BUNROLL * stride) {
        float r[RAST_BUNROLL][6];
#pragma unroll
        for (int k = 0; k < RAST_BUNROLL; ++k) {  // all loads of the trip first
            const int64_t i = i0 + k * stride < N ? i0 + k * stride : i0;
#pragma unroll
            for (int c = 0; c < 6; ++c) r[k][c] = rays6[6 * i + c];
        }
#pragma unroll
        for (int k = 0; k < RAST_BUNROLL; ++k) {
            const int64_t i = i0 + k * stride;
            if (i >= N) break;
            keys[i] = KEY_MISS;
            head[i] = RAST_NONE;
            if (__float_as_uint(r[k][0]) != ox || __float_as_uint(r[k][1]) != oy || __float_as_uint(r[k][2]) != oz) bad |= 1;
            float u, v;
            const int kind = rast_map(r[k][3], r[k][4], r[k][5], f.A, f.E1, f.E2, u, v);
            if (kind == 2) bad |= 2;
            if (kind == 1) {
                const unsigned eu = enc_f(u), ev = enc_f(v);
                lo_u = eu < lo_u ? eu : lo_u; hi_u = eu > hi_u ? eu : hi_u;
                lo_v = ev < lo_v ? ev : lo_v; hi_v = ev > hi_v ? ev : hi_v;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned t;
        t = __shfl_xor(lo_u, off, 64); lo_u = t < lo_u ? t : lo_u;
        t = __shfl_xor(hi_u, off, 64); hi_u = t > hi_u ? t : hi_u;
        t = __shfl_xor(lo_v, off, 64); lo_v = t < lo_v ? t : lo_v;
        t = __shfl_xor(hi_v, off, 64); hi_v = t > hi_v ? t : hi_v;
        bad |= __shfl_xor(bad, off, 64);
    }
    __shared__ unsigned red[4][4];
    __shared__ int badw[4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = lo_u; red[wave][1] = hi_u; red[wave][2] = lo_v; red[wave][3] = hi_v; badw[wave] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            lo_u = red[w][0] < lo_u ? red[w][0] : lo_u; hi_u = red[w][1] > hi_u ? red[w][1] : hi_u;
            lo_v = red[w][2] < lo_v ? red[w][2] : lo_v; hi_v = red[w][3] > hi_v ? red[w][3] : hi_v;
            bad |= badw[w];
        }
        part[blockIdx.x] = make_uint4(lo_u, hi_u, lo_v, hi_v);
        if (bad) { atomicAnd(&h->ok, 0); atomicOr(&h->reason, bad); }
    }
}

// fold of the RAST_BBLOCKS partial bounds; every workgroup that needs the grid does it for itself
__device__ __forceinline__ void rast_fold_bounds(const uint4 *__restrict__ part, unsigned *bnd /* LDS, 4 words */) {
    __shared__ unsigned fred[4][4];
    unsigned lo_u = 0xFFFFFFFFu, hi_u = 0u, lo_v = 0xFFFFFFFFu, hi_v = 0u;
    for (int k = threadIdx.x; k < RAST_BBLOCKS; k += blockDim.x) {
        const uint4 q = part[k];
        lo_u = q.x < lo_u ? q.x : lo_u; hi_u = q.y > hi_u ? q.y : hi_u;
        lo_v = q.z < lo_v ? q.z : lo_v; hi_v = q.w > hi_v ? q.w : hi_v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned t;
        t = __shfl_xor(lo_u, off, 64); lo_u = t < lo_u ? t : lo_u;
        t = __shfl_xor(hi_u, off, 64); hi_u = t > hi_u ? t : hi_u;
        t = __shfl_xor(lo_v, off, 64); lo_v = t < lo_v ? t : lo_v;
        t = __shfl_xor(hi_v, off, 64); hi_v = t > hi_v ? t : hi_v;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { fred[wave][0] = lo_u; fred[wave][1] = hi_u; fred[wave][2] = lo_v; fred[wave][3] = hi_v; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            lo_u = fred[w][0] < lo_u ? fred[w][0] : lo_u; hi_u = fred[w][1] > hi_u ? fred[w][1] : hi_u;
            lo_v = fred[w][2] < lo_v ? fred[w][2] : lo_v; hi_v = fred[w][3] > hi_v ? fred[w][3] : hi_v;
        }
        bnd[0] = lo_u; bnd[1] = hi_u; bnd[2] = lo_v; bnd[3] = hi_v;
    }
    __syncthreads();
}

struct RastGrid { float u0, v0, su, sv; int GX, GY; };

// the grid over [u0, u1] x [v0, v1]: about one cell per ray, never finer than 1e-4 (the mapped
// coordinates carry ~1e-7 of rounding)
__device__ __forceinline__ RastGrid rast_grid(const unsigned *bnd, int64_t N) {
    RastGrid g;
    const bool any = bnd[0] <= bnd[1] && bnd[2] <= bnd[3];
    const float u0 = any ? dec_f(bnd[0]) : 0.f, u1 = any ? dec_f(bnd[1]) : 0.f;
    const float v0 = any ? dec_f(bnd[2]) : 0.f, v1 = any ? dec_f(bnd[3]) : 0.f;
    const float du = fmaxf(u1 - u0, 1e-6f), dv = fmaxf(v1 - v0, 1e-6f);
    const float aspect = fminf(fmaxf(du / dv, 1.0f / 64.0f), 64.0f);
    const float n = (float)(N < (int64_t)1 << 30 ? N : (int64_t)1 << 30);
    float gx = floorf(sqrtf(n * aspect));
    gx = fminf(fmaxf(gx, 1.0f), fminf(n, du * 1e4f + 1.0f));
    float gy = floorf(n / gx);
    gy = fminf(fmaxf(gy, 1.0f), dv * 1e4f + 1.0f);
    g.GX = (int)gx; g.GY = (int)gy;
    g.u0 = u0; g.v0 = v0;
    g.su = gx / du; g.sv = gy / dv;
    return g;
}

__device__ __forceinline__ int rast_clampi(float x, int hi) {  // floor(x) clamped to [0, hi]
    const float c = fminf(fmaxf(floorf(x), 0.0f), (float)hi);
    return (int)c;
}

__global__ __launch_bounds__(256) void rast_insert_kernel(const float *__restrict__ rays6, int64_t N, RastHdr *__restrict__ h,
                                                         unsigned *__restrict__ head, float4 *__restrict__ nodes,
                                                         const uint4 *__restrict__ part) {
    if (h->ok == 0) return;
    __shared__ unsigned bnd[4];
    rast_fold_bounds(part, bnd);
    const RastGrid g = rast_grid(bnd, N);
    if (blockIdx.x == 0 && threadIdx.x == 0) { h->u0 = g.u0; h->v0 = g.v0; h->su = g.su; h->sv = g.sv; h->GX = g.GX; h->GY = g.GY; }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float dx = rays6[6 * i + 3], dy = rays6[6 * i + 4], dz = rays6[6 * i + 5];
    float A[3], E1[3], E2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { A[k] = h->A[k]; E1[k] = h->E1[k]; E2[k] = h->E2[k]; }
    float u, v;
    if (rast_map(dx, dy, dz, A, E1, E2, u, v) != 1) return;
    const int cx = rast_clampi((u - g.u0) * g.su, g.GX - 1), cy = rast_clampi((v - g.v0) * g.sv, g.GY - 1);
    const unsigned nxt = atomicExch(&head[(size_t)cy * g.GX + cx], (unsigned)i);
    nodes[i] = make_float4(dx, dy, dz, __uint_as_float(nxt));
}

// the rays of one chain against one triangle
__device__ __forceinline__ void rast_fail(RastHdr *h, int why) {
    atomicAnd(&h->ok, 0);
    atomicOr(&h->reason, why);
}

__device__ __forceinline__ void rast_test(const float4 nd, unsigned ray, const float *O, const TriRec &q, unsigned f,
                                          unsigned long long *__restrict__ keys) {
    Ray r;
    r.ox = O[0]; r.oy = O[1]; r.oz = O[2];
    r.dx = nd.x; r.dy = nd.y; r.dz = nd.z;
    const MT m = mt_eval(r, q);
    if (mt_accept(m)) atomicMin(&keys[ray], mt_key(m, f));
}

// The cells of the rays that can meet triangle q (origin O, frame A/E1/E2, grid u0/v0/su/sv, GX x GY): the
// bounding rectangle of its mapped corners, widened by a margin that is PROVEN to hold every ray the oracle's
// fp32 test accepts (DESIGN s4.1 "Margin": derivation; tests/test_ray_gpu.py::test_grid_margin_* measure it).
//   An accepted ray has, in exact arithmetic on the stored operands, barycentric coordinates no farther outside
//   the triangle than the roundings of un, vn, det allow: with u = 2^-24, s = O - v0, L the longest edge, m = e2 x e1
//   its exact hit point lies within  6.5 u (3 |s| + L) L^2 / (|m| cos theta_d) + 2.1 u L  of the triangle; mapped
//   to the plane at unit distance along A (a point at depth w, off axis by |p|, moves by (1 + |p|) / w per unit)
//   and with cos theta_d |P - O| = |s . m| / |m| that is at most
//       u c_a [6.5 (3 |s| + L) L^2 + 2.1 L |m|] / |s . m|,     c_a = (1 + |p|) sqrt(1 + |p|^2),
//   |p| taken at its largest over the rectangle.  TWICE that is used, plus 64 u (1 + |p|^2) for the roundings of
//   the two mappings (ray and corners) and an eighth of a cell for the cell arithmetic.  The margin is never cut
//   off (round 2 capped it at 64 cells).  A triangle seen so nearly edge-on that s . m is rounding (the origin
//   within ~1e-6 rad of its plane) has no bounded image, like one that reaches the plane through O: such triangles
//   are LISTED and tested against every ray by rast_full_kernel.  m == 0 exactly: det is 0 for every ray, nothing
//   can be accepted.
struct RastRect { float fx0, fx1, fy0, fy1, mx, my; int full; };  // unwidened rectangle in cell units, the margins, "every cell"
__device__ __forceinline__ bool rast_rect(const TriRec &q, const float *O, const float *A, const float *E1, const float *E2,
                                          float u0, float v0, float su, float sv, int GX, int GY, int &x0, int &x1, int &y0,
                                          int &y1, RastRect &rr) {
    // corners relative to O (v1, v2 re-derived from the record's edges: one more rounding, far below the margins)
    float X[3][3];
    X[0][0] = q.v0x - O[0]; X[0][1] = q.v0y - O[1]; X[0][2] = q.v0z - O[2];
    X[1][0] = X[0][0] + q.e1x; X[1][1] = X[0][1] + q.e1y; X[1][2] = X[0][2] + q.e1z;
    X[2][0] = X[0][0] + q.e2x; X[2][1] = X[0][1] + q.e2y; X[2][2] = X[0][2] + q.e2z;
    float wmin = 3e38f, wmax = -3e38f, scale = fabsf(O[0]) + fabsf(O[1]) + fabsf(O[2]);
    float pu[3], pv[3], pw[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        pw[k] = X[k][0] * A[0] + X[k][1] * A[1] + X[k][2] * A[2];
        pu[k] = X[k][0] * E1[0] + X[k][1] * E1[1] + X[k][2] * E1[2];
        pv[k] = X[k][0] * E2[0] + X[k][1] * E2[1] + X[k][2] * E2[2];
        wmin = fminf(wmin, pw[k]); wmax = fmaxf(wmax, pw[k]);
        scale = fmaxf(scale, fabsf(X[k][0]) + fabsf(X[k][1]) + fabsf(X[k][2]));
    }
    const float eps = 1e-5f * scale;
    bool live = !(wmax < -eps);  // wholly behind the plane through O: t . (d.A) = w > 0 is impossible
    if (q.mx == 0.0f && q.my == 0.0f && q.mz == 0.0f) live = false;  // det = d . m == 0 for every ray
    x0 = 0; x1 = GX - 1; y0 = 0; y1 = GY - 1;
    rr.fx0 = 0.f; rr.fx1 = (float)GX; rr.fy0 = 0.f; rr.fy1 = (float)GY; rr.mx = rr.my = 0.f; rr.full = 1;
    if ((wmin > eps) && (scale < 3e38f)) {  // otherwise it reaches the plane (or has a NaN / infinite corner): no bounded image
        float umin = 3e38f, umax = -3e38f, vmin = 3e38f, vmax = -3e38f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float iw = 1.0f / pw[k];
            const float a = pu[k] * iw, b = pv[k] * iw;
            umin = fminf(umin, a); umax = fmaxf(umax, a);
            vmin = fminf(vmin, b); vmax = fmaxf(vmax, b);
        }
        const float l1 = sqrtf(q.e1x * q.e1x + q.e1y * q.e1y + q.e1z * q.e1z), l2 = sqrtf(q.e2x * q.e2x + q.e2y * q.e2y + q.e2z * q.e2z);
        const float gx = q.e2x - q.e1x, gy = q.e2y - q.e1y, gz = q.e2z - q.e1z;
        const float L = fmaxf(fmaxf(l1, l2), sqrtf(gx * gx + gy * gy + gz * gz));
        const float ls = sqrtf(X[0][0] * X[0][0] + X[0][1] * X[0][1] + X[0][2] * X[0][2]);
        const float lm = sqrtf(q.mx * q.mx + q.my * q.my + q.mz * q.mz);
        const float sm = fabsf(X[0][0] * q.mx + X[0][1] * q.my + X[0][2] * q.mz);
        const float pa = fmaxf(fabsf(umin), fabsf(umax)), pb = fmaxf(fabsf(vmin), fabsf(vmax));
        const float pm2 = (pa * pa + pb * pb) * 1.01f + 1e-6f, pm = sqrtf(pm2);
        const float ca = (1.0f + pm) * sqrtf(1.0f + pm2);
        const float U = 5.9604645e-8f;
        const float dil = 2.0f * U * ca * (6.5f * (3.0f * ls + L) * L * L + 2.1f * L * lm) / fmaxf(sm, 1e-37f) + 64.0f * U * (1.0f + pm2);
        // (s . m itself carries up to 4 u |s| |m| of rounding: below sixteen times u |s| |m| it says nothing -- the
        // triangle has no bounded image; above, it is within a third of the exact value, which the factor two absorbs)
        if (sm > 16.0f * U * ls * lm && dil * su < 1e9f && dil * sv < 1e9f) {  // (false for NaN / inf too)
            const float mx_ = 0.125f + dil * su, my_ = 0.125f + dil * sv;
            rr.fx0 = (umin - u0) * su; rr.fx1 = (umax - u0) * su; rr.fy0 = (vmin - v0) * sv; rr.fy1 = (vmax - v0) * sv;
            rr.mx = mx_; rr.my = my_; rr.full = 0;
            const float fx0 = rr.fx0 - mx_, fx1 = rr.fx1 + mx_, fy0 = rr.fy0 - my_, fy1 = rr.fy1 + my_;
            if (!(fx1 >= 0.0f) || !(fy1 >= 0.0f) || !(fx0 < (float)GX) || !(fy0 < (float)GY)) live = false;  // beside the rays' rectangle
            x0 = rast_clampi(fx0, GX - 1); x1 = rast_clampi(fx1, GX - 1);
            y0 = rast_clampi(fy0, GY - 1); y1 = rast_clampi(fy1, GY - 1);
        }
    }
    return live;
}

// diagnostics (pedp_debug_rast_rects): the rectangle of every triangle and the continuous cell coordinates of every ray
__global__ void rast_debug_tri_kernel(const float *__restrict__ aos, int64_t F, const RastHdr *__restrict__ h, float *__restrict__ out) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const float *rec = aos + f * PEDP_TRI_STRIDE;
    const TriRec q = {rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], rec[6], rec[7], rec[8], rec[9], rec[10], rec[11]};
    float O[3], A[3], E1[3], E2[3];
    for (int k = 0; k < 3; ++k) { O[k] = h->O[k]; A[k] = h->A[k]; E1[k] = h->E1[k]; E2[k] = h->E2[k]; }
    int x0, x1, y0, y1;
    RastRect rr;
    const bool live = rast_rect(q, O, A, E1, E2, h->u0, h->v0, h->su, h->sv, h->GX, h->GY, x0, x1, y0, y1, rr);
    float *o = out + 12 * f;
    o[0] = live ? 1.f : 0.f; o[1] = (float)rr.full; o[2] = rr.fx0; o[3] = rr.fx1; o[4] = rr.fy0; o[5] = rr.fy1; o[6] = rr.mx; o[7] = rr.my;
    o[8] = (float)x0; o[9] = (float)x1; o[10] = (float)y0; o[11] = (float)y1;
}
__global__ void rast_debug_ray_kernel(const float *__restrict__ rays6, int64_t N, const RastHdr *__restrict__ h, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float A[3], E1[3], E2[3], u = 0.f, v = 0.f;
    for (int k = 0; k < 3; ++k) { A[k] = h->A[k]; E1[k] = h->E1[k]; E2[k] = h->E2[k]; }
    const int kind = rast_map(rays6[6 * i + 3], rays6[6 * i + 4], rays6[6 * i + 5], A, E1, E2, u, v);
    out[3 * i] = (float)kind; out[3 * i + 1] = (u - h->u0) * h->su; out[3 * i + 2] = (v - h->v0) * h->sv;
}

__global__ __launch_bounds__(256) void rast_tri_kernel(const float *__restrict__ aos, int64_t F, RastHdr *__restrict__ h,
                                                      const unsigned *__restrict__ head, const float4 *__restrict__ nodes,
                                                      RastItem *__restrict__ items, unsigned long long *__restrict__ keys,
                                                      unsigned *__restrict__ full_list) {
    if (h->ok == 0) return;
    const int lane = threadIdx.x & 63;
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t fl = f < F ? f : F - 1;  // tail lanes re-run the last triangle and drop it below
    const float *rec = aos + fl * PEDP_TRI_STRIDE;
    const TriRec q = {rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], rec[6], rec[7], rec[8], rec[9], rec[10], rec[11]};
    float O[3], A[3], E1[3], E2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { O[k] = h->O[k]; A[k] = h->A[k]; E1[k] = h->E1[k]; E2[k] = h->E2[k]; }
    const int GX = h->GX, GY = h->GY;
    const float u0 = h->u0, v0 = h->v0, su = h->su, sv = h->sv;
    int x0, x1, y0, y1;
    RastRect rr;
    bool live = rast_rect(q, O, A, E1, E2, u0, v0, su, sv, GX, GY, x0, x1, y0, y1, rr) && f < F;
    if (live && rr.full) {  // no bounded image: every ray, by the kernel made for that (a handful of silhouette triangles)
        const int at = atomicAdd(&h->n_full, 1);
        if (at < RAST_FULL_CAP) full_list[at] = (unsigned)f;
        else rast_fail(h, 8);
        live = false;
    }
    const int w = x1 - x0 + 1, hh = y1 - y0 + 1;
    const long long ncell = live ? (long long)w * hh : 0;
    // larger rectangles: cut into tiles of RAST_TILE x RAST_TILE cells.  One reservation in the item
    // table per wave (exclusive scan of the lanes' tile counts), the tiles written by the whole wave.
    const bool big = ncell > RAST_INLINE;
    unsigned long long bigm = __builtin_amdgcn_ballot_w64(big);
    if (bigm != 0ull) {  // wave-uniform
        const int ntx_l = (w + RAST_TILE - 1) / RAST_TILE, nty_l = (hh + RAST_TILE - 1) / RAST_TILE;
        const int n_l = big ? ntx_l * nty_l : 0;  // <= (N / 256 + ...) tiles: fits an int
        int incl = n_l;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const int total = __shfl(incl, 63, 64);
        int base = 0;
        if (lane == 0) base = atomicAdd(&h->n_items, total);
        base = __shfl(base, 0, 64);
        if ((long long)base + total > RAST_ITEM_CAP) {
            if (lane == 0) rast_fail(h, 8);
            bigm = 0ull;
        }
        const int mine = base + incl - n_l;
        while (bigm != 0ull) {
            const int src = __builtin_ctzll(bigm);
            bigm &= bigm - 1ull;
            const int tf = __shfl((int)fl, src, 64), tx0 = __shfl(x0, src, 64), ty0 = __shfl(y0, src, 64);
            const int tw = __shfl(w, src, 64), th = __shfl(hh, src, 64), at = __shfl(mine, src, 64);
            const int ntx = __shfl(ntx_l, src, 64), n = __shfl(n_l, src, 64);
            for (int j = lane; j < n; j += 64) {
                const int ty = j / ntx, tx = j - ty * ntx;
                RastItem it;
                it.tri = tf;
                it.x0 = tx0 + tx * RAST_TILE; it.y0 = ty0 + ty * RAST_TILE;
                it.w = (tx0 + tw - it.x0) < RAST_TILE ? (tx0 + tw - it.x0) : RAST_TILE;
                it.h = (ty0 + th - it.y0) < RAST_TILE ? (ty0 + th - it.y0) : RAST_TILE;
                it.pad0 = it.pad1 = it.pad2 = 0;
                items[at + j] = it;
            }
        }
    }
    if (ncell == 0 || ncell > RAST_INLINE) return;
    const int nc = (int)ncell;
    constexpr int U = 8;  // cells in flight: all their heads are requested first, then the nodes side by side
    for (int c0 = 0; c0 < nc; c0 += U) {
        unsigned idx[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int c = c0 + j;
            const int yy = c / w, xx = c - yy * w;
            idx[j] = c < nc ? head[(size_t)(y0 + yy) * GX + (x0 + xx)] : RAST_NONE;
        }
        int steps = 0;
        for (;;) {
            unsigned all = idx[0];
#pragma unroll
            for (int j = 1; j < U; ++j) all &= idx[j];
            if (all == RAST_NONE) break;
            float4 nd[U];
#pragma unroll
            for (int j = 0; j < U; ++j) nd[j] = nodes[idx[j] != RAST_NONE ? idx[j] : 0u];
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (idx[j] != RAST_NONE) {
                    rast_test(nd[j], idx[j], O, q, (unsigned)f, keys);
                    idx[j] = __float_as_uint(nd[j].w);
                }
            if (++steps > RAST_CHAIN_MAX) { rast_fail(h, 4); break; }
        }
    }
}

// The listed triangles (no bounded image) against EVERY ray: a thread per ray, the records through scalar loads.
__global__ __launch_bounds__(256) void rast_full_kernel(const float *__restrict__ aos, const float *__restrict__ rays6, int64_t N,
                                                       const RastHdr *__restrict__ h, const unsigned *__restrict__ full_list,
                                                       unsigned long long *__restrict__ keys) {
    if (h->ok == 0) return;
    const int n = h->n_full < RAST_FULL_CAP ? h->n_full : RAST_FULL_CAP;
    if (n == 0) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    Ray r;
    r.ox = rays6[6 * i]; r.oy = rays6[6 * i + 1]; r.oz = rays6[6 * i + 2];
    r.dx = rays6[6 * i + 3]; r.dy = rays6[6 * i + 4]; r.dz = rays6[6 * i + 5];
    unsigned long long best = KEY_MISS;
    for (int k = 0; k < n; ++k) {
        const unsigned f = full_list[k];  // wave-uniform
        const MT m = mt_eval(r, aos + (size_t)f * PEDP_TRI_STRIDE);
        if (mt_accept(m)) {
            const unsigned long long key = mt_key(m, f);
            best = key < best ? key : best;
        }
    }
    if (best != KEY_MISS) atomicMin(&keys[i], best);
}

__global__ __launch_bounds__(256) void rast_item_kernel(const float *__restrict__ aos, RastHdr *__restrict__ h,
                                                       const unsigned *__restrict__ head, const float4 *__restrict__ nodes,
                                                       const RastItem *__restrict__ items, unsigned long long *__restrict__ keys) {
    if (h->ok == 0) return;
    const int n_items = h->n_items < RAST_ITEM_CAP ? h->n_items : RAST_ITEM_CAP;
    const int lane = threadIdx.x & 63;
    const int wave0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
    float O[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) O[k] = h->O[k];
    const int GX = h->GX;
    static_assert(RAST_TILE * RAST_TILE == 4 * 64, "a tile is four cells per lane");
    for (int it = wave0; it < n_items; it += n_waves) {  // wave-uniform
        const RastItem I = items[it];
        const float *rec = aos + (size_t)I.tri * PEDP_TRI_STRIDE;
        const TriRec q = {rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], rec[6], rec[7], rec[8], rec[9], rec[10], rec[11]};
        const int nc = I.w * I.h;
        unsigned idx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // the lane's four cells: all heads first, then the chains side by side
            const int c = lane + 64 * j;
            const int yy = c / I.w, xx = c - yy * I.w;
            idx[j] = c < nc ? head[(size_t)(I.y0 + yy) * GX + (I.x0 + xx)] : RAST_NONE;
        }
        int steps = 0;
        while ((idx[0] & idx[1] & idx[2] & idx[3]) != RAST_NONE) {
            float4 nd[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) nd[j] = nodes[idx[j] != RAST_NONE ? idx[j] : 0u];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (idx[j] != RAST_NONE) {
                    rast_test(nd[j], idx[j], O, q, (unsigned)I.tri, keys);
                    idx[j] = __float_as_uint(nd[j].w);
                }
            if (++steps > RAST_CHAIN_MAX) { rast_fail(h, 4); break; }
        }
    }
}

// ------------------------------------------------------------------ sweep, triangle per lane
constexpr int TPL_BLOCK = 256;
constexpr int TPL_RAYS = 4;

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned lo = __shfl_xor((unsigned)(v & 0xFFFFFFFFull), off, 64);
        unsigned hi = __shfl_xor((unsigned)(v >> 32), off, 64);
        unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(TPL_BLOCK) void ray_sweep_tpl_kernel(
    const float4 *__restrict__ tri, int64_t F_padded, int tris_per_chunk, int n_chunks,
    const float *__restrict__ rays6, int64_t N, unsigned long long *__restrict__ keys) {
    const int b = blockIdx.x;
    const int chunk = b % n_chunks;
    const int64_t rb = b / n_chunks;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t ray0 = (rb * (TPL_BLOCK / 64) + wave) * TPL_RAYS;
    if (ray0 >= N) return;  // wave-uniform
    Ray r[TPL_RAYS];
#pragma unroll
    for (int k = 0; k < TPL_RAYS; ++k) {
        int64_t rl = ray0 + k < N ? ray0 + k : N - 1;  // wave-uniform address: stays in SGPRs
        r[k].ox = rays6[6 * rl + 0]; r[k].oy = rays6[6 * rl + 1]; r[k].oz = rays6[6 * rl + 2];
        r[k].dx = rays6[6 * rl + 3]; r[k].dy = rays6[6 * rl + 4]; r[k].dz = rays6[6 * rl + 5];
    }
    unsigned long long best[TPL_RAYS];
#pragma unroll
    for (int k = 0; k < TPL_RAYS; ++k) best[k] = KEY_MISS;
    int64_t f0 = (int64_t)chunk * tris_per_chunk;
    int64_t f1 = f0 + tris_per_chunk;
    if (f1 > F_padded) f1 = F_padded;
    for (int64_t f = f0 + lane; f < f1; f += 64) {
        const float4 a = tri[3 * f + 0], bq = tri[3 * f + 1], c = tri[3 * f + 2];
        const float rec[12] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int k = 0; k < TPL_RAYS; ++k) {
            MT m = mt_eval(r[k], rec);
            if (mt_accept(m)) {
                unsigned long long key = mt_key(m, (unsigned)f);
                best[k] = key < best[k] ? key : best[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < TPL_RAYS; ++k) {
        unsigned long long v = wave_min_u64(best[k]);  // the wavefront-wide min-t reduction
        if (lane == 0 && ray0 + k < N && v != KEY_MISS) atomicMin(&keys[ray0 + k], v);
    }
}

// ------------------------------------------------------------------ finalize
__global__ __launch_bounds__(256) void ray_finalize_kernel(const float *__restrict__ aos, const float *__restrict__ rays6,
                                    int64_t N, unsigned long long *__restrict__ keys,
                                    float *__restrict__ t_hit, uint32_t *__restrict__ prim_id,
                                    float *__restrict__ uv, const RastHdr *__restrict__ rast, RastHdr *__restrict__ rast_next,
                                    int *__restrict__ rast_status, const f2 *__restrict__ pairs, int groups_total, int reset_keys = 0) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && rast) {  // variant 4: how the cast went, for the host's choice next time; the other header gets its start values
        rast_status[1] = (int)(N & 0x7FFFFFFF);
        rast_status[0] = rast->ok ? 0 : rast->reason;
        __threadfence_system();
        rast_next->ok = 1; rast_next->n_items = 0; rast_next->reason = 0; rast_next->n_full = 0;
    }
    if (i >= N) return;
    if (rast && rast->ok == 0) {
        // the grid could not answer this cast: every triangle for this ray, the exhaustive sweep's loop
        // (what the grid did find is kept: a minimum over more candidates)
        Ray r;
        r.ox = rays6[6 * i + 0]; r.oy = rays6[6 * i + 1]; r.oz = rays6[6 * i + 2];
        r.dx = rays6[6 * i + 3]; r.dy = rays6[6 * i + 4]; r.dz = rays6[6 * i + 5];
        keys[i] = sweep_groups<false>(r, pairs, aos, 0, groups_total, keys[i]);
    }
    if (i >= N) return;
    unsigned long long key = keys[i];
    if (reset_keys) keys[i] = KEY_MISS;   // a ray set's keys start the next cast clean (nothing else clears them)
    unsigned id = (unsigned)(key & 0xFFFFFFFFull);
    if (key == KEY_MISS) {
        t_hit[i] = __uint_as_float(0x7F800000u);
        prim_id[i] = 0xFFFFFFFFu;
        if (uv) { uv[2 * i] = 0.0f; uv[2 * i + 1] = 0.0f; }
        return;
    }
    t_hit[i] = __uint_as_float((unsigned)(key >> 32));
    prim_id[i] = id;
    if (uv) {
        Ray r;
        r.ox = rays6[6 * i + 0]; r.oy = rays6[6 * i + 1]; r.oz = rays6[6 * i + 2];
        r.dx = rays6[6 * i + 3]; r.dy = rays6[6 * i + 4]; r.dz = rays6[6 * i + 5];
        MT m = mt_eval(r, aos + (size_t)id * PEDP_TRI_STRIDE);
        unsigned sg = __float_as_uint(m.det) & 0x80000000u;
        float U = __uint_as_float(__float_as_uint(m.un) ^ sg);
        float V = __uint_as_float(__float_as_uint(m.vn) ^ sg);
        float ad = fabsf(m.det);
        uv[2 * i] = __fdiv_rn(U, ad);
        uv[2 * i + 1] = __fdiv_rn(V, ad);
    }
}

// TriangleMesh.transform (pose_estimation.py:406-409, defect_projection.py:549-550) followed by
// from_legacy's float32 cast (defect_projection.py:245): Open3D forms T * (x, y, z, 1) in
// float64, divides by the fourth component and the tensor mesh stores float32.  One thread per
// vertex, fixed operation order ((T0 x + T1 y) + T2 z) + T3, no contraction.
struct Pose16 { double m[16]; };

__global__ void pose_verts_kernel(const double *__restrict__ v64, int64_t V, Pose16 T, int identity,
                                  float *__restrict__ v32) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const double x = v64[3 * i], y = v64[3 * i + 1], z = v64[3 * i + 2];
    if (identity) {
        v32[3 * i] = (float)x;
        v32[3 * i + 1] = (float)y;
        v32[3 * i + 2] = (float)z;
        return;
    }
    double h[4];
    for (int r = 0; r < 4; ++r)
        h[r] = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(T.m[4 * r], x), __dmul_rn(T.m[4 * r + 1], y)),
                                   __dmul_rn(T.m[4 * r + 2], z)), T.m[4 * r + 3]);
    v32[3 * i] = (float)(h[0] / h[3]);
    v32[3 * i + 1] = (float)(h[1] / h[3]);
    v32[3 * i + 2] = (float)(h[2] / h[3]);
}

// the same T * (x, y, z, 1) / w in float64, kept as float64: what TriangleMesh.transform leaves in `vertices`
__global__ void pose_verts64_kernel(const double *__restrict__ v64, int64_t V, Pose16 T, double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const double x = v64[3 * i], y = v64[3 * i + 1], z = v64[3 * i + 2];
    double h[4];
    for (int r = 0; r < 4; ++r)
        h[r] = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(T.m[4 * r], x), __dmul_rn(T.m[4 * r + 1], y)),
                                   __dmul_rn(T.m[4 * r + 2], z)), T.m[4 * r + 3]);
    out[3 * i] = h[0] / h[3];
    out[3 * i + 1] = h[1] / h[3];
    out[3 * i + 2] = h[2] / h[3];
}

}  // namespace

extern "C" {

// Everything derived from the posed float32 vertices: triangle records, pair records,
// cluster and super-cluster spheres.  Enqueued on the context's stream.
static hipError_t mesh_build_records(pedp_ctx_t c, pedp_mesh_s *m, const float *d_verts, const uint32_t *d_tris) {
    int grid = (int)((m->F_padded + 255) / 256);
    hipLaunchKernelGGL(tri_setup_kernel, dim3(grid), dim3(256), 0, c->stream, d_verts, d_tris, m->F, m->F_padded, m->tri);
    hipLaunchKernelGGL(pair_general_kernel, dim3(grid), dim3(256), 0, c->stream, m->tri, m->F_padded, m->tri2);
    hipLaunchKernelGGL(cluster_sphere_kernel, dim3((unsigned)((m->n_clusters + 255) / 256)), dim3(256), 0, c->stream,
                       m->tri, m->F, m->n_clusters, (float4 *)m->spheres);
    hipLaunchKernelGGL(supercluster_sphere_kernel, dim3((unsigned)((m->n_super + 3) / 4)), dim3(256), 0, c->stream,
                       (const float4 *)m->spheres, m->n_clusters, m->n_super, (float4 *)m->super_spheres);
    return hipGetLastError();
}

static int mesh_alloc(pedp_ctx_t c, int64_t V, const uint32_t *tris, int64_t F, const char *who, pedp_mesh_s **out) {
    PEDP_REQUIRE(V >= 0 && F >= 0 && F < (int64_t)0x7FFFFF00, "%s: sizes out of range", who);
    for (int64_t i = 0; i < 3 * F; ++i)
        PEDP_REQUIRE((int64_t)tris[i] < V, "%s: triangle %lld references vertex %u >= V=%lld", who,
                     (long long)(i / 3), tris[i], (long long)V);
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    pedp_mesh_s *m = new (std::nothrow) pedp_mesh_s();
    if (!m) { pedp_set_error("%s: out of host memory", who); return PEDP_ERR_ALLOC; }
    m->ctx = c;
    m->device = c->device;
    m->V = V;
    m->F = F;
    m->F_padded = ((F + 63) / 64) * 64;
    if (m->F_padded == 0) m->F_padded = 64;
    m->n_clusters = m->F_padded / CL_TRIS;
    m->n_super = (m->n_clusters + 63) / 64;
    hipError_t e = hipMalloc((void **)&m->tri, sizeof(float) * PEDP_TRI_STRIDE * (size_t)m->F_padded);
    if (e == hipSuccess) e = hipMalloc((void **)&m->tri2, sizeof(float) * PAIR_GEN * (size_t)(m->F_padded / 2));
    if (e == hipSuccess) e = hipMalloc((void **)&m->spheres, sizeof(float4) * (size_t)m->n_clusters);
    if (e == hipSuccess) e = hipMalloc((void **)&m->super_spheres, sizeof(float4) * (size_t)m->n_super);
    if (e == hipSuccess) e = hipMalloc((void **)&m->verts32, sizeof(float) * 3 * (size_t)(V ? V : 1));
    if (e == hipSuccess) e = hipMalloc((void **)&m->idx, sizeof(uint32_t) * 3 * (size_t)(F ? F : 1));
    if (e == hipSuccess && F) e = hipMemcpyAsync(m->idx, tris, sizeof(uint32_t) * 3 * (size_t)F, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) {
        pedp_set_error("%s: %s", who, hipGetErrorString(e));
        pedp_mesh_destroy(m);
        return PEDP_ERR_HIP;
    }
    *out = m;
    return PEDP_OK;
}

int pedp_mesh_create(pedp_ctx_t c, const float *verts, int64_t V, const uint32_t *tris, int64_t F,
                     pedp_mesh_t *out) {
    PEDP_REQUIRE(c && out, "pedp_mesh_create: null context/output");
    *out = nullptr;
    PEDP_REQUIRE((verts || V == 0) && (tris || F == 0), "pedp_mesh_create: null arrays");
    pedp_mesh_s *m = nullptr;
    int rc = mesh_alloc(c, V, tris, F, "pedp_mesh_create", &m);
    if (rc) return rc;
    hipError_t e = hipSuccess;
    if (V) e = hipMemcpyAsync(m->verts32, verts, sizeof(float) * 3 * (size_t)V, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = mesh_build_records(c, m, m->verts32, m->idx);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    // a fixed mesh needs neither the vertices nor the indices again
    (void)hipFree(m->verts32);
    (void)hipFree(m->idx);
    m->verts32 = nullptr;
    m->idx = nullptr;
    if (e != hipSuccess) {
        pedp_set_error("pedp_mesh_create: %s", hipGetErrorString(e));
        pedp_mesh_destroy(m);
        return PEDP_ERR_HIP;
    }
    *out = m;
    return PEDP_OK;
}

int pedp_mesh_create_posable(pedp_ctx_t c, const double *verts, int64_t V, const uint32_t *tris, int64_t F,
                             pedp_mesh_t *out) {
    PEDP_REQUIRE(c && out, "pedp_mesh_create_posable: null context/output");
    *out = nullptr;
    PEDP_REQUIRE((verts || V == 0) && (tris || F == 0), "pedp_mesh_create_posable: null arrays");
    pedp_mesh_s *m = nullptr;
    int rc = mesh_alloc(c, V, tris, F, "pedp_mesh_create_posable", &m);
    if (rc) return rc;
    hipError_t e = hipMalloc((void **)&m->verts64, sizeof(double) * 3 * (size_t)(V ? V : 1));
    if (e == hipSuccess && V) e = hipMemcpyAsync(m->verts64, verts, sizeof(double) * 3 * (size_t)V, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // the caller's arrays are free again
    if (e != hipSuccess) {
        pedp_set_error("pedp_mesh_create_posable: %s", hipGetErrorString(e));
        pedp_mesh_destroy(m);
        return PEDP_ERR_HIP;
    }
    *out = m;
    return pedp_mesh_set_pose(m, nullptr);
}

int pedp_mesh_set_pose(pedp_mesh_t m, const double T[16]) {
    PEDP_REQUIRE(m, "pedp_mesh_set_pose: null mesh");
    PEDP_REQUIRE(m->verts64, "pedp_mesh_set_pose: mesh was not created with pedp_mesh_create_posable");
    pedp_ctx_t c = m->ctx;
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    Pose16 P;
    for (int k = 0; k < 16; ++k) P.m[k] = T ? T[k] : ((k % 5) == 0 ? 1.0 : 0.0);
    if (m->V)
        hipLaunchKernelGGL(pose_verts_kernel, dim3((unsigned)((m->V + 255) / 256)), dim3(256), 0, c->stream, m->verts64,
                           m->V, P, T ? 0 : 1, m->verts32);
    PEDP_HIP_CHECK(hipGetLastError());
    PEDP_HIP_CHECK(mesh_build_records(c, m, m->verts32, m->idx));
    return PEDP_OK;
}

int pedp_mesh_posed_vertices(pedp_mesh_t m, const double T[16], int mem, double *out) {
    PEDP_REQUIRE(m && T && out, "pedp_mesh_posed_vertices: null argument");
    PEDP_REQUIRE(m->verts64, "pedp_mesh_posed_vertices: mesh was not created with pedp_mesh_create_posable");
    PEDP_REQUIRE(mem == PEDP_HOST || mem == PEDP_DEVICE, "pedp_mesh_posed_vertices: bad mem flag %d", mem);
    if (m->V == 0) return PEDP_OK;
    pedp_ctx_t c = m->ctx;
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    Pose16 P;
    for (int k = 0; k < 16; ++k) P.m[k] = T[k];
    double *d_out = out;
    if (mem == PEDP_HOST) {
        int st = c->ray_out.reserve(sizeof(double) * 3 * (size_t)m->V);
        if (st) return st;
        d_out = (double *)c->ray_out.ptr;
    }
    hipLaunchKernelGGL(pose_verts64_kernel, dim3((unsigned)((m->V + 255) / 256)), dim3(256), 0, c->stream, m->verts64, m->V, P, d_out);
    PEDP_HIP_CHECK(hipGetLastError());
    if (mem == PEDP_HOST) { int dn_ = pedp_download(c, out, d_out, sizeof(double) * 3 * (size_t)m->V); if (dn_) return dn_; }
    return PEDP_OK;
}

void pedp_mesh_destroy(pedp_mesh_t m) {
    if (!m) return;
    // the context may already be gone (Python collects handles in any order): use the ordinal
    // stored at creation; hipFree itself waits for the device's outstanding work
    (void)hipSetDevice(m->device);
    if (m->tri) (void)hipFree(m->tri);
    if (m->tri2) (void)hipFree(m->tri2);
    if (m->spheres) (void)hipFree(m->spheres);
    if (m->super_spheres) (void)hipFree(m->super_spheres);
    if (m->verts64) (void)hipFree(m->verts64);
    if (m->verts32) (void)hipFree(m->verts32);
    if (m->idx) (void)hipFree(m->idx);
    delete m;
}

int pedp_mesh_size(pedp_mesh_t m, int64_t *V, int64_t *F) {
    PEDP_REQUIRE(m, "pedp_mesh_size: null mesh");
    if (V) *V = m->V;
    if (F) *F = m->F;
    return PEDP_OK;
}

int pedp_raycast_configure(pedp_ctx_t c, int tri_chunks, int variant) {
    PEDP_REQUIRE(c, "pedp_raycast_configure: null context");
    PEDP_REQUIRE(tri_chunks >= 0 && tri_chunks % 8 == 0, "pedp_raycast_configure: tri_chunks must be a multiple of 8");
    PEDP_REQUIRE(variant >= 0 && variant <= 5, "pedp_raycast_configure: variant must be 0..5");
    c->ray_tri_chunks = tri_chunks;
    c->ray_variant = variant;
    return PEDP_OK;
}

int pedp_raycast(pedp_ctx_t c, pedp_mesh_t mesh, const float *rays6, int64_t N, int mem,
                 float *t_hit, uint32_t *prim_id, float *uv) {
    PEDP_REQUIRE(c && mesh, "pedp_raycast: null context/mesh");
    PEDP_REQUIRE(mesh->ctx == c, "pedp_raycast: mesh belongs to another context");
    PEDP_REQUIRE(N >= 0 && N < (int64_t)1 << 40, "pedp_raycast: N out of range");
    PEDP_REQUIRE(mem == PEDP_HOST || mem == PEDP_DEVICE, "pedp_raycast: bad mem flag %d", mem);
    if (N == 0) return PEDP_OK;
    PEDP_REQUIRE(rays6 && t_hit && prim_id, "pedp_raycast: null arrays");
    PEDP_HIP_CHECK(hipSetDevice(c->device));

    const float *d_rays = rays6;
    float *d_t = t_hit, *d_uv = uv;
    uint32_t *d_id = prim_id;
    if (mem == PEDP_HOST) {
        int st = c->ray_in.reserve(sizeof(float) * 6 * (size_t)N);
        if (st) return st;
        st = c->ray_out.reserve(sizeof(float) * 4 * (size_t)N);
        if (st) return st;
        { int up_ = pedp_upload(c, c->ray_in.ptr, rays6, sizeof(float) * 6 * (size_t)N); if (up_) return up_; }
        d_rays = (const float *)c->ray_in.ptr;
        d_t = (float *)c->ray_out.ptr;
        d_id = (uint32_t *)(d_t + N);
        d_uv = uv ? (float *)(d_id + N) : nullptr;
    }
    int st = c->ray_keys.reserve(sizeof(unsigned long long) * (size_t)N);
    if (st) return st;
    unsigned long long *keys = (unsigned long long *)c->ray_keys.ptr;

    int variant = c->ray_variant;
    if (variant == 0) {
        variant = 2;
        if (N >= 16384) {
            // the grid of variant 4 unless an earlier cast of this many rays reported that it could not
            // answer them (origins that differ, rays behind the frame's plane, crowded cells, a full item
            // table): such rays would pay the grid's four kernels AND the exhaustive completion in every
            // call, where variant 3's chunked general-origin sweep is several times faster
            if (c->rast_status && (((volatile int *)c->rast_status)[0] & (1 | 2 | 4 | 8))) c->rast_avoid_n = ((volatile int *)c->rast_status)[1];
            variant = (c->rast_avoid_n == (int)(N & 0x7FFFFFFF)) ? 3 : 4;
        }
    }
    RastHdr *rast = nullptr;
    c->ray_last_variant = variant;
    if (variant != 4) PEDP_HIP_CHECK(hipMemsetAsync(keys, 0xFF, sizeof(unsigned long long) * (size_t)N, c->stream));
    if (variant == 4) {
        PEDP_REQUIRE(N < (int64_t)0x7FFFFFF0 && mesh->F < (int64_t)0x7FFFFFF0, "pedp_raycast: variant 4 takes fewer than 2^31 rays / triangles");
        if (!c->rast_status) {
            PEDP_HIP_CHECK(hipHostMalloc((void **)&c->rast_status, 64, hipHostMallocMapped));
            c->rast_status[0] = 0; c->rast_status[1] = -1;
        }
        int *d_status = nullptr;
        PEDP_HIP_CHECK(hipHostGetDevicePointer((void **)&d_status, c->rast_status, 0));
        const size_t sz_head = align256(sizeof(unsigned) * (size_t)N), sz_nodes = align256(sizeof(float4) * (size_t)N);
        const size_t sz_part = align256(sizeof(uint4) * RAST_BBLOCKS);
        st = c->ray_rast.reserve(512 + sz_part + sz_head + sz_nodes + sizeof(RastItem) * (size_t)RAST_ITEM_CAP + sizeof(unsigned) * RAST_FULL_CAP);
        if (st) return st;
        char *base = (char *)c->ray_rast.ptr;
        // two headers, used in turn: the last kernel of a cast puts the OTHER one back to its start values
        // (it still reads its own), so no kernel ever resets a word that workgroups of the same launch test
        static_assert(sizeof(RastHdr) <= 256, "header slot");
        rast = (RastHdr *)(base + 256 * (c->rast_seq & 1));
        RastHdr *rast_next = (RastHdr *)(base + 256 * ((c->rast_seq + 1) & 1));
        uint4 *part = (uint4 *)(base + 512);
        unsigned *head = (unsigned *)(base + 512 + sz_part);
        float4 *nodes = (float4 *)(base + 512 + sz_part + sz_head);
        RastItem *items = (RastItem *)(base + 512 + sz_part + sz_head + sz_nodes);
        unsigned *full_list = (unsigned *)(items + RAST_ITEM_CAP);
        if (c->rast_hdr_ready != (void *)base) {  // new buffer, or a cast that did not reach its last kernel
            RastHdr h0;
            memset(&h0, 0, sizeof(h0));
            h0.ok = 1;
            PEDP_HIP_CHECK(hipMemcpyAsync(rast, &h0, sizeof(h0), hipMemcpyHostToDevice, c->stream));
            PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
        }
        c->rast_hdr_ready = nullptr;
        PEDP_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
        hipLaunchKernelGGL(rast_bounds_kernel, dim3(RAST_BBLOCKS), dim3(256), 0, c->stream, d_rays, N, rast, keys, head, part);
        hipLaunchKernelGGL(rast_insert_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, d_rays, N, rast, head,
                           nodes, (const uint4 *)part);
        if (mesh->F > 0)
            hipLaunchKernelGGL(rast_tri_kernel, dim3((unsigned)((mesh->F + 255) / 256)), dim3(256), 0, c->stream, mesh->tri,
                               mesh->F, rast, head, nodes, items, keys, full_list);
        hipLaunchKernelGGL(rast_item_kernel, dim3(RAST_ITEM_WAVES / 4), dim3(256), 0, c->stream, mesh->tri, rast, head, nodes,
                           items, keys);
        hipLaunchKernelGGL(rast_full_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, mesh->tri, d_rays, N,
                           (const RastHdr *)rast, (const unsigned *)full_list, keys);
        PEDP_HIP_CHECK(hipGetLastError());
        PEDP_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
        c->ray_timed = true;
        hipLaunchKernelGGL(ray_finalize_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, mesh->tri, d_rays, N,
                           keys, d_t, d_id, d_uv, (const RastHdr *)rast, rast_next, d_status, (const f2 *)mesh->tri2,
                           (int)(mesh->F_padded / (2 * RPL_PAIRS)));
        PEDP_HIP_CHECK(hipGetLastError());
        c->rast_hdr_ready = (void *)base;
        c->rast_seq += 1;
    } else if (variant == 1 || variant == 3 || variant == 5) {
        // aux layout: [flag + bounds + seg info: 256 B][shared pair records][cone records][hist][perm]
        //             [packet masks][packet counts][segment table]
        const int64_t n_packets = (N + 63) / 64;
        const int n_cwords = (int)((mesh->n_clusters + 63) / 64);
        int64_t max_segs = (n_packets * mesh->n_clusters + RLIST - 1) / RLIST + n_packets;
        if (max_segs < 16384) max_segs = 16384;
        PEDP_REQUIRE(max_segs < (int64_t)1 << 26, "pedp_raycast: problem too large for the segment table");
        const size_t sz_tri3 = align256(sizeof(float) * PAIR_SH * (size_t)(mesh->F_padded / 2));
        const size_t sz_cone = align256(sizeof(float4) * 2 * (size_t)(mesh->n_clusters + mesh->n_super));
        const size_t sz_hist = align256(sizeof(unsigned) * BIN_CELLS);
        const size_t sz_perm = 0;  // the direction order lives in its own buffer, kept between calls
        const size_t sz_mask = align256(sizeof(unsigned long long) * (size_t)n_packets * (size_t)n_cwords);
        const size_t sz_pcnt = align256(sizeof(int) * (size_t)n_packets);
        const size_t sz_seg = align256(sizeof(int) * (size_t)max_segs);
        const size_t sz_mf = variant == 1 ? align256((size_t)192 * (size_t)mesh->F_padded) : 0;   // matrix-sweep records: 3 x 32 bf16 per triangle
        PEDP_REQUIRE(mesh->F_padded % 16 == 0, "pedp_raycast: padded triangle count is not a multiple of 16");
        st = c->ray_aux.reserve(256 + sz_tri3 + sz_cone + sz_hist + sz_perm + sz_mask + sz_pcnt + 3 * sz_seg + sz_mf);
        if (st) return st;
        char *aux = (char *)c->ray_aux.ptr;
        int *flag = (int *)aux;
        unsigned *bounds = (unsigned *)(aux + 16);
        int *seg_info = (int *)(aux + 64);
        float *tri3 = (float *)(aux + 256);
        float4 *cones = (float4 *)(aux + 256 + sz_tri3);
        unsigned *hist = (unsigned *)(aux + 256 + sz_tri3 + sz_cone);
        // kept direction order: [stale flag: 256 B][samples][perm]; rebuilt when the ray count or the buffer changes
        const size_t sz_samples = align256(sizeof(float) * 3 * RAY_SAMPLES);
        const void *order_before = c->ray_order.ptr;
        st = c->ray_order.reserve(256 + sz_samples + sizeof(unsigned) * (size_t)N);
        if (st) return st;
        int *stale = (int *)c->ray_order.ptr;
        float *samples = (float *)((char *)c->ray_order.ptr + 256);
        unsigned *perm = (unsigned *)((char *)c->ray_order.ptr + 256 + sz_samples);
        const bool order_kept = c->ray_order.ptr == order_before && c->ray_order_n == N;
        unsigned long long *pmask = (unsigned long long *)(aux + 256 + sz_tri3 + sz_cone + sz_hist + sz_perm);
        int *pk_cnt = (int *)((char *)pmask + sz_mask);
        int *seg_pk = (int *)((char *)pk_cnt + sz_pcnt);
        int *seg_rank0 = (int *)((char *)seg_pk + sz_seg);
        int *seg_n = (int *)((char *)seg_rank0 + sz_seg);
        unsigned short *mf_rec = (unsigned short *)((char *)seg_n + sz_seg);
        PEDP_HIP_CHECK(hipMemsetAsync(flag, 0xFF, sizeof(int), c->stream));
        hipLaunchKernelGGL(origin_check_kernel, dim3(2 * c->num_cus), dim3(256), 0, c->stream, d_rays, N, flag);
        hipLaunchKernelGGL(pair_shared_kernel, dim3((unsigned)((mesh->F_padded + 255) / 256)), dim3(256), 0, c->stream,
                           mesh->tri, mesh->F_padded, d_rays, flag, tri3);
        int64_t ray_blocks = (N + RPL_BLOCK - 1) / RPL_BLOCK;
        int groups_total = (int)(mesh->F_padded / (2 * RPL_PAIRS));
        int n_chunks = c->ray_tri_chunks;
        if (n_chunks == 0) {
            // enough workgroups for >= 4 rounds over the chip, triangle chunks not below 2k
            n_chunks = 8;
            while (ray_blocks * n_chunks < 4 * 8 * (int64_t)c->num_cus && groups_total / (n_chunks * 2) >= 512) n_chunks *= 2;
        }
        int gpc = (groups_total + n_chunks - 1) / n_chunks;
        int64_t grid = ray_blocks * n_chunks;
        PEDP_REQUIRE(grid < (int64_t)0x7FFFFFFF, "pedp_raycast: grid too large");
        PEDP_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
        if (variant == 3) {
            // shared origin: cluster cones, direction binning, culled sweep
            // 0 = "keep the order unless the sampled directions differ", 1 = rebuild (other ray count or buffer)
            PEDP_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)stale, order_kept ? 0 : 1, 1, c->stream));
            c->ray_order_n = N;
            hipLaunchKernelGGL(ray_order_check_kernel, dim3(RAY_SAMPLES / 256), dim3(256), 0, c->stream, d_rays, N, samples, stale,
                               bounds);
            float4 *scones = cones + 2 * mesh->n_clusters;
            hipLaunchKernelGGL(cluster_cone_kernel, dim3((unsigned)((mesh->n_clusters + 255) / 256)), dim3(256), 0, c->stream,
                               (const float4 *)mesh->spheres, mesh->n_clusters, d_rays, flag, cones);
            hipLaunchKernelGGL(cluster_cone_kernel, dim3((unsigned)((mesh->n_super + 255) / 256)), dim3(256), 0, c->stream,
                               (const float4 *)mesh->super_spheres, mesh->n_super, d_rays, flag, scones);
            hipLaunchKernelGGL(ray_bounds_kernel, dim3(c->num_cus), dim3(256), 0, c->stream, d_rays, N, stale, bounds, hist);
            hipLaunchKernelGGL(ray_count_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, d_rays, N, stale,
                               bounds, hist);
            hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(1024), 0, c->stream, stale, hist);
            hipLaunchKernelGGL(ray_scatter_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, d_rays, N, stale,
                               bounds, hist, perm);
            hipLaunchKernelGGL(ray_cull_mask_kernel, dim3((unsigned)((n_packets + 3) / 4)), dim3(256), 0, c->stream,
                               (const float4 *)cones, (const float4 *)scones, (int)mesh->n_clusters, n_cwords, d_rays, perm, N,
                               pmask, pk_cnt, flag);
            hipLaunchKernelGGL(ray_segment_kernel, dim3(1), dim3(1024), 0, c->stream, pk_cnt, (int)n_packets, seg_pk,
                               seg_rank0, seg_n, (int)max_segs, seg_info, flag);
            hipLaunchKernelGGL(ray_sweep_seg_kernel, dim3((unsigned)((max_segs + 3) / 4)), dim3(RPL_BLOCK), 0, c->stream,
                               (const f2 *)tri3, mesh->tri, n_cwords, pmask, seg_pk, seg_rank0, seg_n, seg_info, d_rays,
                               perm, N, keys);
        } else if (variant == 5) {   // round 3's exhaustive kernel: the packed fp32 loop on the vector pipe
            hipLaunchKernelGGL(ray_sweep_rpl_kernel<true>, dim3((unsigned)grid), dim3(RPL_BLOCK), 0, c->stream,
                               (const f2 *)tri3, mesh->tri, groups_total, gpc, n_chunks, d_rays, N, keys, flag);
        } else {                     // exhaustive on the matrix pipe: bf16 filter, exact test on the rare branch
            hipLaunchKernelGGL(mfma_rec_kernel, dim3((unsigned)((mesh->F_padded + 255) / 256)), dim3(256), 0, c->stream, mesh->tri,
                               mesh->F_padded, d_rays, flag, mf_rec);
            const int tg_total = (int)(mesh->F_padded / 16);
            int mf_chunks = n_chunks;   // (its own count: the general-origin kernel behind keeps n_chunks / gpc / grid)
            if (c->ray_tri_chunks == 0 && mf_chunks < 32 && tg_total / 32 >= 64) mf_chunks = 32;   // (measured: 4.35 ms against 4.54 with 8)
            const int tgpc = (tg_total + mf_chunks - 1) / mf_chunks;
            static const int mf_rg = getenv("PEDP_MF_RG") ? atoi(getenv("PEDP_MF_RG")) : 8;   // (experiments: 8 or 16 ray groups per wave)
            const int64_t mf_rays = 16 * (mf_rg == 16 ? 16 : 8) * MF_WAVES;
            const int64_t mf_grid = ((N + mf_rays - 1) / mf_rays) * mf_chunks;
            PEDP_REQUIRE(mf_grid < (int64_t)0x7FFFFFFF, "pedp_raycast: grid too large");
            if (mf_rg == 16)
                hipLaunchKernelGGL(ray_sweep_mfma_kernel<16>, dim3((unsigned)mf_grid), dim3(64 * MF_WAVES), 0, c->stream, (const uint4 *)mf_rec,
                                   mesh->tri, tg_total, tgpc, mf_chunks, d_rays, N, keys, flag);
            else
                hipLaunchKernelGGL(ray_sweep_mfma_kernel<8>, dim3((unsigned)mf_grid), dim3(64 * MF_WAVES), 0, c->stream, (const uint4 *)mf_rec,
                                   mesh->tri, tg_total, tgpc, mf_chunks, d_rays, N, keys, flag);
        }
        hipLaunchKernelGGL(ray_sweep_rpl_kernel<false>, dim3((unsigned)grid), dim3(RPL_BLOCK), 0, c->stream,
                           (const f2 *)mesh->tri2, mesh->tri, groups_total, gpc, n_chunks, d_rays, N, keys, flag);
    } else {
        int64_t rays_per_block = (TPL_BLOCK / 64) * TPL_RAYS;
        int64_t ray_blocks = (N + rays_per_block - 1) / rays_per_block;
        int n_chunks = c->ray_tri_chunks;
        if (n_chunks == 0) {
            n_chunks = 8;
            while (ray_blocks * n_chunks < 4 * 8 * (int64_t)c->num_cus && mesh->F_padded / (n_chunks * 2) >= 1024) n_chunks *= 2;
        }
        int tpc = (int)((mesh->F_padded + n_chunks - 1) / n_chunks);
        tpc = ((tpc + 63) / 64) * 64;
        int64_t grid = ray_blocks * n_chunks;
        PEDP_REQUIRE(grid < (int64_t)0x7FFFFFFF, "pedp_raycast: grid too large");
        PEDP_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
        hipLaunchKernelGGL(ray_sweep_tpl_kernel, dim3((unsigned)grid), dim3(TPL_BLOCK), 0, c->stream,
                           (const float4 *)mesh->tri, mesh->F_padded, tpc, n_chunks, d_rays, N, keys);
    }
    if (variant != 4) {
        PEDP_HIP_CHECK(hipGetLastError());
        PEDP_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
        c->ray_timed = true;
        int64_t grid = (N + 255) / 256;
        hipLaunchKernelGGL(ray_finalize_kernel, dim3((unsigned)grid), dim3(256), 0, c->stream, mesh->tri, d_rays, N,
                           keys, d_t, d_id, d_uv, (const RastHdr *)nullptr, (RastHdr *)nullptr, (int *)nullptr, (const f2 *)nullptr, 0);
        PEDP_HIP_CHECK(hipGetLastError());
    }
    if (mem == PEDP_HOST) {
        { int dn_ = pedp_download(c, t_hit, d_t, sizeof(float) * (size_t)N); if (dn_) return dn_; }
        { int dn_ = pedp_download(c, prim_id, d_id, sizeof(uint32_t) * (size_t)N); if (dn_) return dn_; }
        if (uv) { int dn_ = pedp_download(c, uv, d_uv, sizeof(float) * 2 * (size_t)N); if (dn_) return dn_; }
        PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    return PEDP_OK;
}

// ---------------------------------------------------------------- resident ray sets (include/pedp.h)
}  // extern "C"
struct pedp_rayset_s {
    pedp_ctx_t ctx = nullptr;
    int device = 0;
    int64_t N = 0;
    char *buf = nullptr;        // one allocation: rays, keys, two headers, partial bounds, heads, nodes, items, full list
    float *rays = nullptr;
    unsigned long long *keys = nullptr;
    RastHdr *hdr[2] = {nullptr, nullptr};
    unsigned *head = nullptr;
    float4 *nodes = nullptr;
    RastItem *items = nullptr;
    unsigned *full_list = nullptr;
    int *status = nullptr;      // pinned: [0] why the grid failed in the last cast (0: it did not), [1] ray count
    unsigned seq = 0;           // casts enqueued: picks one of the two headers
    bool grid_ok = false;       // the build found the rays servable; cleared for good by a cast that was not answered by the grid
    int last_variant = 0, last_status = 0;
};
extern "C" {

int pedp_rayset_create(pedp_ctx_t c, const float *rays6, int64_t N, int mem, pedp_rayset_t *out) {
    PEDP_REQUIRE(c && out, "pedp_rayset_create: null context/output");
    *out = nullptr;
    PEDP_REQUIRE(N > 0 && N < (int64_t)0x7FFFFFF0, "pedp_rayset_create: N out of range");
    PEDP_REQUIRE(rays6, "pedp_rayset_create: null rays");
    PEDP_REQUIRE(mem == PEDP_HOST || mem == PEDP_DEVICE, "pedp_rayset_create: bad mem flag %d", mem);
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    pedp_rayset_s *r = new (std::nothrow) pedp_rayset_s();
    if (!r) { pedp_set_error("pedp_rayset_create: out of host memory"); return PEDP_ERR_ALLOC; }
    r->ctx = c; r->device = c->device; r->N = N;
    const size_t sz_rays = align256(sizeof(float) * 6 * (size_t)N), sz_keys = align256(sizeof(unsigned long long) * (size_t)N);
    const size_t sz_part = align256(sizeof(uint4) * RAST_BBLOCKS), sz_head = align256(sizeof(unsigned) * (size_t)N);
    const size_t sz_nodes = align256(sizeof(float4) * (size_t)N), sz_items = align256(sizeof(RastItem) * (size_t)RAST_ITEM_CAP);
    const size_t total = sz_rays + sz_keys + 512 + sz_part + sz_head + sz_nodes + sz_items + sizeof(unsigned) * RAST_FULL_CAP;
    hipError_t e = hipMalloc((void **)&r->buf, total);
    if (e == hipSuccess) e = hipHostMalloc((void **)&r->status, 64, hipHostMallocMapped);
    if (e != hipSuccess) { pedp_set_error("pedp_rayset_create: %s", hipGetErrorString(e)); pedp_rayset_destroy(r); return PEDP_ERR_ALLOC; }
    r->status[0] = 0; r->status[1] = -1;
    char *b = r->buf;
    r->rays = (float *)b; b += sz_rays;
    r->keys = (unsigned long long *)b; b += sz_keys;
    r->hdr[0] = (RastHdr *)b; r->hdr[1] = (RastHdr *)(b + 256); b += 512;
    uint4 *part = (uint4 *)b; b += sz_part;
    r->head = (unsigned *)b; b += sz_head;
    r->nodes = (float4 *)b; b += sz_nodes;
    r->items = (RastItem *)b; b += sz_items;
    r->full_list = (unsigned *)b;
    int rc = PEDP_OK;
    if (mem == PEDP_HOST) rc = pedp_upload(c, r->rays, rays6, sizeof(float) * 6 * (size_t)N);
    else e = hipMemcpyAsync(r->rays, rays6, sizeof(float) * 6 * (size_t)N, hipMemcpyDeviceToDevice, c->stream);
    RastHdr h0;
    memset(&h0, 0, sizeof(h0));
    h0.ok = 1;
    if (rc == PEDP_OK && e == hipSuccess) e = hipMemcpyAsync(r->hdr[0], &h0, sizeof(h0), hipMemcpyHostToDevice, c->stream);
    if (rc == PEDP_OK && e == hipSuccess) {
        // the frame, the bounds and the chains: what pedp_raycast's variant 4 builds per call (the bounds kernel also sets
        // every key to "miss" and every head to "none")
        hipLaunchKernelGGL(rast_bounds_kernel, dim3(RAST_BBLOCKS), dim3(256), 0, c->stream, (const float *)r->rays, N, r->hdr[0], r->keys, r->head, part);
        hipLaunchKernelGGL(rast_insert_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, (const float *)r->rays, N, r->hdr[0],
                           r->head, r->nodes, (const uint4 *)part);
        e = hipGetLastError();
    }
    RastHdr hb;
    if (rc == PEDP_OK && e == hipSuccess) e = hipMemcpyAsync(&hb, r->hdr[0], sizeof(hb), hipMemcpyDeviceToHost, c->stream);
    if (rc == PEDP_OK && e == hipSuccess) e = hipStreamSynchronize(c->stream);   // (also: the caller's array and h0 are free again)
    if (rc != PEDP_OK || e != hipSuccess) {
        if (rc == PEDP_OK) pedp_set_error("pedp_rayset_create: %s", hipGetErrorString(e));
        pedp_rayset_destroy(r);
        return rc != PEDP_OK ? rc : PEDP_ERR_HIP;
    }
    r->grid_ok = hb.ok != 0;
    r->last_status = hb.ok ? 0 : hb.reason;
    if (r->grid_ok) {  // the second header: the same frame and grid, its counters at their start values
        e = hipMemcpyAsync(r->hdr[1], r->hdr[0], sizeof(RastHdr), hipMemcpyDeviceToDevice, c->stream);
        if (e != hipSuccess) { pedp_set_error("pedp_rayset_create: %s", hipGetErrorString(e)); pedp_rayset_destroy(r); return PEDP_ERR_HIP; }
    }
    *out = r;
    return PEDP_OK;
}

void pedp_rayset_destroy(pedp_rayset_t r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    if (r->buf) (void)hipFree(r->buf);
    if (r->status) (void)hipHostFree(r->status);
    delete r;
}

int pedp_raycast_rayset(pedp_ctx_t c, pedp_mesh_t mesh, pedp_rayset_t r, int mem, float *t_hit, uint32_t *prim_id, float *uv) {
    PEDP_REQUIRE(c && mesh && r, "pedp_raycast_rayset: null argument");
    PEDP_REQUIRE(mesh->ctx == c && r->ctx == c, "pedp_raycast_rayset: mesh or ray set belongs to another context");
    PEDP_REQUIRE(mem == PEDP_HOST || mem == PEDP_DEVICE, "pedp_raycast_rayset: bad mem flag %d", mem);
    PEDP_REQUIRE(t_hit && prim_id, "pedp_raycast_rayset: null arrays");
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    const int64_t N = r->N;
    // a cast the grid did not answer (its last kernel completed it exhaustively, correctly but slowly): from now on the
    // other variants, from the resident copy
    if (r->grid_ok && ((volatile int *)r->status)[1] == (int)(N & 0x7FFFFFFF) && ((volatile int *)r->status)[0] != 0) {
        r->grid_ok = false;
        r->last_status = ((volatile int *)r->status)[0];
    }
    if (!r->grid_ok || mesh->F >= (int64_t)0x7FFFFFF0) {
        const int keep = c->ray_variant;
        if (c->ray_variant == 0 || c->ray_variant == 4) c->ray_variant = 3;   // (differing origins: variant 3 takes the general sweep on the device)
        int rc = PEDP_OK;
        if (mem == PEDP_DEVICE) rc = pedp_raycast(c, mesh, r->rays, N, PEDP_DEVICE, t_hit, prim_id, uv);
        else {
            int st = c->ray_out.reserve(sizeof(float) * 4 * (size_t)N);
            if (st) { c->ray_variant = keep; return st; }
            float *d_t = (float *)c->ray_out.ptr;
            uint32_t *d_id = (uint32_t *)(d_t + N);
            float *d_uv = uv ? (float *)(d_id + N) : nullptr;
            rc = pedp_raycast(c, mesh, r->rays, N, PEDP_DEVICE, d_t, d_id, d_uv);
            if (!rc) rc = pedp_download(c, t_hit, d_t, sizeof(float) * (size_t)N);
            if (!rc) rc = pedp_download(c, prim_id, d_id, sizeof(uint32_t) * (size_t)N);
            if (!rc && uv) rc = pedp_download(c, uv, d_uv, sizeof(float) * 2 * (size_t)N);
            if (!rc) PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
        }
        r->last_variant = c->ray_last_variant;
        c->ray_variant = keep;
        return rc;
    }
    float *d_t = t_hit, *d_uv = uv;
    uint32_t *d_id = prim_id;
    if (mem == PEDP_HOST) {
        int st = c->ray_out.reserve(sizeof(float) * 4 * (size_t)N);
        if (st) return st;
        d_t = (float *)c->ray_out.ptr;
        d_id = (uint32_t *)(d_t + N);
        d_uv = uv ? (float *)(d_id + N) : nullptr;
    }
    int *d_status = nullptr;
    PEDP_HIP_CHECK(hipHostGetDevicePointer((void **)&d_status, r->status, 0));
    RastHdr *h = r->hdr[r->seq & 1], *h_next = r->hdr[(r->seq + 1) & 1];
    PEDP_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
    if (mesh->F > 0)
        hipLaunchKernelGGL(rast_tri_kernel, dim3((unsigned)((mesh->F + 255) / 256)), dim3(256), 0, c->stream, mesh->tri, mesh->F, h,
                           (const unsigned *)r->head, (const float4 *)r->nodes, r->items, r->keys, r->full_list);
    hipLaunchKernelGGL(rast_item_kernel, dim3(RAST_ITEM_WAVES / 4), dim3(256), 0, c->stream, mesh->tri, h, (const unsigned *)r->head,
                       (const float4 *)r->nodes, (const RastItem *)r->items, r->keys);
    hipLaunchKernelGGL(rast_full_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, mesh->tri, (const float *)r->rays, N,
                       (const RastHdr *)h, (const unsigned *)r->full_list, r->keys);
    PEDP_HIP_CHECK(hipGetLastError());
    PEDP_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
    c->ray_timed = true;
    hipLaunchKernelGGL(ray_finalize_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, mesh->tri, (const float *)r->rays, N,
                       r->keys, d_t, d_id, d_uv, (const RastHdr *)h, h_next, d_status, (const f2 *)mesh->tri2,
                       (int)(mesh->F_padded / (2 * RPL_PAIRS)), 1);
    PEDP_HIP_CHECK(hipGetLastError());
    r->seq += 1;
    r->last_variant = 4;
    c->ray_last_variant = 4;
    if (mem == PEDP_HOST) {
        { int dn_ = pedp_download(c, t_hit, d_t, sizeof(float) * (size_t)N); if (dn_) return dn_; }
        { int dn_ = pedp_download(c, prim_id, d_id, sizeof(uint32_t) * (size_t)N); if (dn_) return dn_; }
        if (uv) { int dn_ = pedp_download(c, uv, d_uv, sizeof(float) * 2 * (size_t)N); if (dn_) return dn_; }
        PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    return PEDP_OK;
}

int pedp_rayset_last_variant(pedp_rayset_t r, int *variant, int *grid_status) {
    PEDP_REQUIRE(r && variant && grid_status, "pedp_rayset_last_variant: null argument");
    PEDP_HIP_CHECK(hipSetDevice(r->device));
    PEDP_HIP_CHECK(hipStreamSynchronize(r->ctx->stream));
    *variant = r->last_variant;
    *grid_status = r->last_variant == 4 ? ((volatile int *)r->status)[0] : r->last_status;
    return PEDP_OK;
}

int pedp_raycast_last_variant(pedp_ctx_t c, int *variant, int *grid_status) {
    PEDP_REQUIRE(c && variant && grid_status, "pedp_raycast_last_variant: null argument");
    PEDP_REQUIRE(c->ray_last_variant > 0, "pedp_raycast_last_variant: no cast has run on this context");
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    *variant = c->ray_last_variant;
    *grid_status = (c->ray_last_variant == 4 && c->rast_status) ? ((volatile int *)c->rast_status)[0] : 0;
    return PEDP_OK;
}

/* Diagnostics of the triangle-driven ray stage (tests/test_ray_gpu.py::test_grid_margin_*): run a variant-4 cast
 * of the N host rays and return, for every triangle, [live, every-cell, fx0, fx1, fy0, fy1 (the bounding rectangle of
 * its mapped corners in cell units, before widening), mx, my (the margins), x0, x1, y0, y1 (the cells visited)] and,
 * for every ray, [kind (1: mapped), cell coordinate x, y (continuous)]; grid[0..1] = GX, GY, grid[2] = status. */
int pedp_debug_rast_rects(pedp_ctx_t c, pedp_mesh_t mesh, const float *rays6, int64_t N, float *tri_out, float *ray_out, int *grid) {
    PEDP_REQUIRE(c && mesh && rays6 && tri_out && ray_out && grid && N > 0, "pedp_debug_rast_rects: bad argument");
    const int keep = c->ray_variant;
    c->ray_variant = 4;
    std::vector<float> t((size_t)N);
    std::vector<uint32_t> id((size_t)N);
    const int rc = pedp_raycast(c, mesh, rays6, N, PEDP_HOST, t.data(), id.data(), nullptr);
    c->ray_variant = keep;
    if (rc) return rc;
    // the header the cast used (the finalize kernel reset the OTHER one) and the rays it uploaded are still in place
    const RastHdr *h = (const RastHdr *)((char *)c->ray_rast.ptr + 256 * ((c->rast_seq + 1) & 1));
    float *d_tri = nullptr, *d_ray = nullptr;
    PEDP_HIP_CHECK(hipMalloc((void **)&d_tri, sizeof(float) * 12 * (size_t)(mesh->F > 0 ? mesh->F : 1)));
    PEDP_HIP_CHECK(hipMalloc((void **)&d_ray, sizeof(float) * 3 * (size_t)N));
    if (mesh->F > 0)
        hipLaunchKernelGGL(rast_debug_tri_kernel, dim3((unsigned)((mesh->F + 255) / 256)), dim3(256), 0, c->stream, mesh->tri, mesh->F, h, d_tri);
    hipLaunchKernelGGL(rast_debug_ray_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, (const float *)c->ray_in.ptr, N, h, d_ray);
    RastHdr hh;
    hipError_t e = hipMemcpyAsync(tri_out, d_tri, sizeof(float) * 12 * (size_t)mesh->F, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ray_out, d_ray, sizeof(float) * 3 * (size_t)N, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&hh, h, sizeof(hh), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_tri);
    (void)hipFree(d_ray);
    PEDP_HIP_CHECK(e);
    grid[0] = hh.GX; grid[1] = hh.GY; grid[2] = c->rast_status ? c->rast_status[0] : -1;
    return PEDP_OK;
}

/* Diagnostics of the matrix-pipe filter of the exhaustive sweep: for N host rays of ONE origin the filter's score of every
 * (ray, triangle) pair (>= 0: the pair goes to the exact test) and the slack inside it, N x F float32 each, row-major. */
int pedp_debug_mfma_scores(pedp_ctx_t c, pedp_mesh_t mesh, const float *rays6, int64_t N, float *score, float *slack) {
    PEDP_REQUIRE(c && mesh && rays6 && score && slack && N > 0 && mesh->F > 0, "pedp_debug_mfma_scores: bad argument");
    PEDP_REQUIRE(mesh->ctx == c && N * mesh->F <= ((int64_t)1 << 27), "pedp_debug_mfma_scores: foreign mesh, or more than 2^27 pairs");
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    float *d_rays = nullptr, *d_sc = nullptr, *d_sl = nullptr;
    unsigned short *d_rec = nullptr;
    int *d_flag = nullptr;
    const size_t pairs = (size_t)N * (size_t)mesh->F;
    hipError_t e = hipMalloc((void **)&d_rays, sizeof(float) * 6 * (size_t)N);
    if (e == hipSuccess) e = hipMalloc((void **)&d_sc, sizeof(float) * pairs);
    if (e == hipSuccess) e = hipMalloc((void **)&d_sl, sizeof(float) * pairs);
    if (e == hipSuccess) e = hipMalloc((void **)&d_rec, (size_t)192 * (size_t)mesh->F_padded);
    if (e == hipSuccess) e = hipMalloc((void **)&d_flag, sizeof(int));
    if (e == hipSuccess) e = hipMemcpyAsync(d_rays, rays6, sizeof(float) * 6 * (size_t)N, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0xFF, sizeof(int), c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(mfma_rec_kernel, dim3((unsigned)((mesh->F_padded + 255) / 256)), dim3(256), 0, c->stream, mesh->tri, mesh->F_padded,
                           (const float *)d_rays, (const int *)d_flag, d_rec);
        hipLaunchKernelGGL(mfma_debug_kernel, dim3((unsigned)((N + 15) / 16)), dim3(64), 0, c->stream, (const uint4 *)d_rec,
                           (int)(mesh->F_padded / 16), (const float *)d_rays, N, mesh->F, d_sc, d_sl);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(score, d_sc, sizeof(float) * pairs, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(slack, d_sl, sizeof(float) * pairs, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_rays); (void)hipFree(d_sc); (void)hipFree(d_sl); (void)hipFree(d_rec); (void)hipFree(d_flag);
    PEDP_HIP_CHECK(e);
    return PEDP_OK;
}

int pedp_raycast_last_sweep_ms(pedp_ctx_t c, float *ms) {
    PEDP_REQUIRE(c && ms, "pedp_raycast_last_sweep_ms: null argument");
    PEDP_REQUIRE(c->ray_timed, "pedp_raycast_last_sweep_ms: no sweep has run on this context");
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    PEDP_HIP_CHECK(hipEventSynchronize(c->ev1));
    PEDP_HIP_CHECK(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return PEDP_OK;
}

}  // extern "C"
