// ICP (registration_icp) for gfx950 (MI355X).
//
// Replaces o3d.pipelines.registration.registration_icp as called by
// src/pose_estimation.py:519-521 (refine_registration) and :654-660 (z search, one
// iteration).  Semantics contract: oracle/icp.c (float64; exact nearest neighbour,
// strict d^2 < r^2, point-to-plane 6x6 / point-to-point Umeyama update, Open3D's
// convergence rule).
//
// Kernels per correspondence pass (all on the context's stream, no host round trip: a
// device-side `done` flag turns the remaining passes into no-ops):
//   icp_transform_pack   P <- U * P in float64 (the oracle's operation order).  Points farther
//                        than r from the target's bounding box cannot be inliers and are
//                        dropped here; the others are compacted (wave-aggregated atomic) into
//                        the float32 MFMA operand array, (-2x', -2y', -2z', 1) in coordinates
//                        centred on the target centroid.
//   nn_sweep             the one dense contraction.  v_mfma_f32_16x16x4_f32 evaluates
//                        g(i,j) = |t'_j|^2 - 2 s'_i . t'_j for 16 target x 16 scene
//                        points per instruction (A = target tile (x,y,z,|t|^2), B = scene
//                        block); a wave keeps 8 scene blocks (128 points) as B operands in
//                        registers and streams a chunk of target tiles as coalesced 256-B
//                        fragments; the grid is (scene blocks) x (target chunks), sized on
//                        the device from the candidate count so small problems still fill
//                        the chip.  The loop is software-pipelined: the MFMA of tile k+1 is
//                        issued before the 6-op VALU epilogue of tile k (best tile value,
//                        its tile, second-best tile value per lane), so one wave alone keeps
//                        the matrix pipe busy.
//   nn_select            exact selection from the per-(point, lane group, chunk) triples.
//                        fp32 g is only a FILTER: with a proven error bound eps_i the true
//                        nearest neighbour lies in the tiles whose value is within
//                        2 eps_i of the minimum; those <= 16 points are re-scored in
//                        float64 in the oracle's exact formula; if a second tile of one
//                        lane group is inside the window the point goes to nn_fallback.
//   nn_fallback          exact float64 brute force for the (rare) ambiguous points.
//   icp_accumulate       float64 J^T J / J^T r (or Umeyama moments), fixed partition and
//                        fixed reduction tree: run-to-run bit-stable.
//   icp_reduce + icp_solve   29-double packet (optionally summed over ranks by the
//                        caller's hook), pivoted LDLT 6x6 / Jacobi SVD 3x3, T <- U * T,
//                        fitness / rmse / convergence.
#include "pedp_internal.h"
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <new>

#ifndef PEDP_NN_EXPERIMENT
#define PEDP_NN_EXPERIMENT 0
#endif

#ifndef PEDP_ICP_STAMPS
#define PEDP_ICP_STAMPS 0
#endif
#if PEDP_ICP_STAMPS
// Diagnostic build only (tools/icp_stamps.py): s_memtime at the phase boundaries of the fused
// pass's kernels, written to a buffer nothing else reads.  [kernel 0..2][workgroup or wave][8]
__device__ long long g_icp_stamps[3][4096][8];
#define PEDP_STAMP(kern, unit, slot)                                                   \
    do {                                                                               \
        if ((unit) < 4096) g_icp_stamps[kern][unit][slot] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
extern "C" int pedp_debug_icp_stamps(long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_icp_stamps), sizeof(long long) * 3 * 4096 * 8) == hipSuccess ? 0 : -3;
}
// s_memrealtime (100 MHz, one clock for the whole device) per pass and workgroup of the pass kernel:
// [0] entry, [1] state read, [2] chunks done, [3] ticket returned, [4] pass closed (last workgroup only)
__device__ long long g_icp_rt[32][512][8];
// per wave of the pass kernel (last pass that ran): [0] start [1] slots ready [2] culled+swept [3] selected [4] sums done (s_memtime),
// [5] words << 32 | batches << 16 | wide << 8 | slots, [6] tiles
__device__ long long g_icp_wave[512][8][16];
#define PEDP_WV(slot, val)                                                                         \
    do {                                                                                           \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 512 && blockIdx.y == 0)                        \
            g_icp_wave[blockIdx.x][threadIdx.x >> 6][slot] = (long long)(val);                     \
    } while (0)
extern "C" int pedp_debug_icp_wave(long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_icp_wave), sizeof(long long) * 512 * 8 * 16) == hipSuccess ? 0 : -3;
}
#define PEDP_RT(pass, slot)                                                                     \
    do {                                                                                        \
        if (threadIdx.x == 0 && (pass) < 32 && blockIdx.x < 512 && blockIdx.y == 0)             \
            g_icp_rt[pass][blockIdx.x][slot] = (long long)__builtin_amdgcn_s_memrealtime();     \
    } while (0)
extern "C" int pedp_debug_icp_rt(long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_icp_rt), sizeof(long long) * 32 * 512 * 8) == hipSuccess ? 0 : -3;
}
#else
#define PEDP_STAMP(kern, unit, slot) do {} while (0)
#define PEDP_RT(pass, slot) do {} while (0)
#define PEDP_WV(slot, val) do {} while (0)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PACKET = 29;  // doubles per partial-sum packet
constexpr int NN_SB = 8;    // scene blocks of 16 points per wave
constexpr int NN_WAVES = 4;
constexpr int NN_PTS_PER_WG = NN_SB * 16 * NN_WAVES;  // 512
constexpr int NN_TU = 4;    // rows are padded to multiples of 16 * NN_TU (= the largest unit)
// A target UNIT is QT MFMA tiles (16 QT rows): the granularity of culling, of the sweep's
// fold-and-compare epilogue and of the exact re-scoring.  QT = 1 inside a registration with a
// finite radius (finest culling), QT = 4 for dense sweeps (3 instead of 6 VALU ops per MFMA).
constexpr int NN_LIST_TILES = 2048; // most MFMA tiles one sweep wave walks (its unit list lives in LDS)
constexpr int SEG_MIN_TILES = 64;   // MFMA tiles per sweep segment at least
constexpr int CULL_WORDS = 8;       // 64-unit mask words one cull wave fills
constexpr int SORT_BITS = 16;         // spatial sort: 65536^3 Hilbert-ordered cells over the cloud's own bounding box
constexpr int SORT_KEY_BITS = 3 * SORT_BITS + 1;  // + the bucket of points without a cell (non-finite coordinates)
constexpr int NN_TILE_PAD = 2 * NN_TU;  // readable pad tiles behind the last real tile
constexpr int ACC_BLOCKS = 256;
constexpr int ACC_THREADS = 256;

struct IcpState {
    double T[16];
    double upd[16];
    double fitness, rmse, prev_fitness, prev_rmse;
    double centroid[3];
    int done;
    int iters;
    int fb_count;
    int n_cand;   // slots of the compacted candidate list this pass (128 per scene block)
    int n_blocks; // scene blocks (one per transform wave with at least one candidate)
    int n_segs, seg_len;       // sweep segments of this pass and their length in tiles
    long long sum_tiles;       // surviving (scene block, target tile) pairs, summed over passes
    long long sum_cand;  // statistics over the passes of this registration
    long long sum_fb;
    // fused pass: the live chunk set is rebuilt from the whole scene when `rebuild` is set (pass 0,
    // and whenever the accumulated motion could have carried an outside point into reach)
    int rebuild;
    int n_rebuilds;
    int n_live;                // entries of the live list (written by icp_finish_kernel)
    // parameters of this registration that the fused pass reads from here rather than from kernel
    // arguments, so that one captured graph serves start poses with different radii and criteria
    double r2, r2cut, r2live;  // r^2; rounding-safe r^2 of the box test; (r + margin)^2 of the live test
    double reachE, margin;     // motion bound: r + margin + rho, margin
    double rel_fitness, rel_rmse, n_source;
    float r1, r_search, wide_radius, r2f;
    int pass, max_iter;        // the pass the fused kernels are in (advanced by icp_finish_kernel), and the limit
    double mu_theta, mu_tau;   // sum of |R - I|_F and of |t + (R - I) c| since the last rebuild
    // tickets and sign-offs are counted on from launch to launch (nothing to reset at the end of a pass): what the
    // counters read when this launch began
    unsigned ticket_base, idle_base;
    int n_planned;             // passes of this registration that ran under a visit plan (statistics)
    int nonce;                 // of this registration (<< 16 in the tags of the visit plan: entries of an earlier registration never match)
    double T_init[16];         // the start transformation: slot 0 of the update history (arrives with the state, no copy of its own)
};

__device__ __forceinline__ double dmul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double dadd(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ double dsub(double a, double b) { return __dsub_rn(a, b); }

// the oracle's dist2(): (dx*dx + dy*dy) + dz*dz, no FMA
__device__ __forceinline__ double dist2(double ax, double ay, double az, double bx, double by, double bz) {
    double dx = dsub(ax, bx), dy = dsub(ay, by), dz = dsub(az, bz);
    return dadd(dadd(dmul(dx, dx), dmul(dy, dy)), dmul(dz, dz));
}

// ------------------------------------------------------------------ spatial order of a cloud
// Stable sort by the Hilbert-curve index of the point's cell in a 65536^3 grid over the cloud's OWN
// bounding box: consecutive entries of `perm` are neighbours in space, so 128-point scene chunks and
// 16-point target tiles are compact and their bounding spheres are small.  Inside a cell the points
// keep ascending index (stable radix sort).  The order -- and with it the order of every float64 sum
// of a registration -- is a function of the cloud's data alone: not of the registration that first
// touched the handle, not of its start pose (round 2 laid 256^3 cells over the region the target
// could reach from the first start pose, so two handles of the same data could sum in different
// orders).  48 bits keep the cells far below the point spacing even when one stray point stretches
// the box a hundredfold.
// 3-D Hilbert index of cell (x, y, z), SORT_BITS bits per axis (Skilling, "Programming the
// Hilbert curve", 2004: axes -> transpose, then bit interleave).  Unlike Morton order, points
// that are consecutive along the curve are always neighbours in space, so no 128-point scene
// block or 16-point target tile straddles a long jump (such blocks would defeat the culling).
__device__ __forceinline__ unsigned long long hilbert3(unsigned x, unsigned y, unsigned z) {
    unsigned X[3] = {x, y, z};
    const unsigned M = 1u << (SORT_BITS - 1);
    for (unsigned Q = M; Q > 1; Q >>= 1) {
        const unsigned Pm = Q - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= Pm;
            else { const unsigned t = (X[0] ^ X[i]) & Pm; X[0] ^= t; X[i] ^= t; }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    unsigned t = 0;
    for (unsigned Q = M; Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    unsigned long long h = 0;
#pragma unroll
    for (int b = SORT_BITS - 1; b >= 0; --b)
        h = (h << 3) | (unsigned long long)((((X[0] >> b) & 1u) << 2) | (((X[1] >> b) & 1u) << 1) | ((X[2] >> b) & 1u));
    return h;
}
// Cell of point i; a point without a cell (a non-finite coordinate) goes to the extra bucket behind
// the curve.
__device__ __forceinline__ unsigned long long point_cell(const double *__restrict__ pts, int64_t i, double lox, double loy,
                                                         double loz, double sx, double sy, double sz) {
    const double fx = (pts[3 * i] - lox) * sx, fy = (pts[3 * i + 1] - loy) * sy, fz = (pts[3 * i + 2] - loz) * sz;
    const double top = (double)(1 << SORT_BITS);
    if (!(fx >= 0.0 && fx < top && fy >= 0.0 && fy < top && fz >= 0.0 && fz < top)) return 1ull << (3 * SORT_BITS);
    return hilbert3((unsigned)(int)fx, (unsigned)(int)fy, (unsigned)(int)fz);
}
// the box is read where it was made: on the device (lo xyz, hi xyz)
__global__ void cell_key_kernel(const double *__restrict__ pts, int64_t N, const double *__restrict__ box,
                                unsigned long long *__restrict__ key) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double sc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double ext = box[3 + k] - box[k];
        sc[k] = ext > 0.0 && isfinite(ext) ? (double)(1 << SORT_BITS) / ext * (1.0 - 1e-9) : 0.0;
    }
    key[i] = point_cell(pts, i, box[0], box[1], box[2], sc[0], sc[1], sc[2]);
}
// Bounding spheres (centre xyz, radius; float64, the cloud's own frame) of the eight runs of 16 consecutive
// points of every 128-point chunk of the spatial order: a rebuild pass of a registration asks the spheres,
// moved by the pose so far, which chunks can be near the target at all, and touches only those chunks'
// points.  (One sphere per chunk let through twice as many chunks as are live; with eight the workgroups of a
// rebuild pass mostly get one chunk each.)  One wave per chunk: lane = point of a half, 16 lanes = a run.
__global__ __launch_bounds__(64) void chunk_sphere_kernel(const double *__restrict__ pts, const int32_t *__restrict__ perm,
                                                          int64_t N, double *__restrict__ sph /* [chunk][8][4] */) {
    const int64_t chunk = blockIdx.x;
    const int lane = threadIdx.x;
    const double big = 1.7976931348623157e308;
    for (int h = 0; h < 2; ++h) {
        double lo[3] = {big, big, big}, hi[3] = {-big, -big, -big};
        const int64_t k = chunk * 128 + h * 64 + lane;
        if (k < N) {
            const int64_t i = perm[k];
            for (int c = 0; c < 3; ++c) lo[c] = hi[c] = pts[3 * i + c];
        }
        for (int c = 0; c < 3; ++c)
            for (int off = 8; off >= 1; off >>= 1) {
                const double l2 = __shfl_xor(lo[c], off, 64), h2 = __shfl_xor(hi[c], off, 64);
                lo[c] = l2 < lo[c] ? l2 : lo[c];
                hi[c] = h2 > hi[c] ? h2 : hi[c];
            }
        if ((lane & 15) == 0) {
            double *o = sph + ((chunk * 8) + h * 4 + (lane >> 4)) * 4;
            if (!(hi[0] >= lo[0])) {  // a run behind the cloud's last point: matches nothing
                o[0] = o[1] = o[2] = 0.0;
                o[3] = -1.0;
            } else {
                double m[3], r2 = 0.0;
                for (int c = 0; c < 3; ++c) {
                    m[c] = 0.5 * (lo[c] + hi[c]);
                    const double e = hi[c] - m[c];
                    r2 += e * e;
                }
                // non-finite coordinates give a non-finite sphere: such a chunk is never skipped
                o[0] = m[0]; o[1] = m[1]; o[2] = m[2];
                o[3] = sqrt(r2) * (1.0 + 1e-12) + 1e-300;
            }
        }
    }
}

// ------------------------------------------------------------------ target preparation
// Sorted target operand: row k holds point perm[k] as float4 (x', y', z', |t'|^2), centred on
// c; pad rows can never win.  One bounding sphere per 64-row unit (centred coordinates);
// radius < 0 marks a unit without real points.
__global__ void pack_target_kernel(const double *__restrict__ pts, const int32_t *__restrict__ perm, int64_t N,
                                   int64_t N_pad, double cx, double cy, double cz, float4 *__restrict__ out) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N_pad) return;
    if (k >= N) { out[k] = make_float4(0.f, 0.f, 0.f, 1e30f); return; }
    const int64_t i = perm[k];
    float x = (float)(pts[3 * i] - cx), y = (float)(pts[3 * i + 1] - cy), z = (float)(pts[3 * i + 2] - cz);
    double w = (double)x * x + (double)y * y + (double)z * z;
    out[k] = make_float4(x, y, z, (float)w);
}
// sorted float64 rows of the target (row k = point perm[k]): x y z nx ny nz, so that the exact
// re-scoring reads a candidate row -- and with it what the winner contributes to the sums -- with
// one contiguous 48-byte load instead of perm -> point -> normal
__global__ void sort_rows_kernel(const double *__restrict__ pts, const double *__restrict__ nrm, const int32_t *__restrict__ perm,
                                 int64_t N, int64_t N_pad, double *__restrict__ out) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N_pad) return;
    const bool real = k < N;
    const int64_t i = real ? perm[k] : 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        out[6 * k + c] = real ? pts[3 * i + c] : 0.0;
        out[6 * k + 3 + c] = real && nrm ? nrm[3 * i + c] : 0.0;
    }
}
__global__ void tile_sphere_kernel(const float4 *__restrict__ t4, int64_t N, int64_t n_units_all, int UNIT_ROWS,
                                   float4 *__restrict__ sph) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_units_all) return;
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    int n = 0;
    for (int r = 0; r < UNIT_ROWS; ++r) {
        int64_t k = t * UNIT_ROWS + r;
        if (k >= N) break;
        const float4 p = t4[k];
        lo[0] = fminf(lo[0], p.x); hi[0] = fmaxf(hi[0], p.x);
        lo[1] = fminf(lo[1], p.y); hi[1] = fmaxf(hi[1], p.y);
        lo[2] = fminf(lo[2], p.z); hi[2] = fmaxf(hi[2], p.z);
        ++n;
    }
    if (n == 0) { sph[t] = make_float4(0.f, 0.f, 0.f, -1.f); return; }
    const float cx = 0.5f * (lo[0] + hi[0]), cy = 0.5f * (lo[1] + hi[1]), cz = 0.5f * (lo[2] + hi[2]);
    float r2 = 0.f;
    for (int r = 0; r < n; ++r) {
        const float4 p = t4[t * UNIT_ROWS + r];
        const float dx = p.x - cx, dy = p.y - cy, dz = p.z - cz;
        r2 = fmaxf(r2, dx * dx + dy * dy + dz * dz);
    }
    sph[t] = make_float4(cx, cy, cz, sqrtf(r2) * 1.0001f + 1e-6f * (fabsf(cx) + fabsf(cy) + fabsf(cz)) + 1e-30f);
}

// ------------------------------------------------------------------ transform + pack
// mode 0: P <- T * src (first pass; T = init), mode 1: P <- upd * P.
// A wave takes 128 consecutive points of the scene's spatial order (two per lane).  Points
// farther than r from the target's bounding box (lo, hi) have no neighbour within r
// (d_nn >= d_box) and are written off as "no correspondence" here.  If any point of the wave
// survives, the wave claims one 128-slot scene block (atomic counter), compacts its survivors
// into it, pads the rest with dummies (list = -1) and stores the block's bounding sphere --
// so every scene block of the sweep is one compact patch of space.
struct PackPoint {
    bool cand;
    int i;
    float sx, sy, sz;
};
__device__ __forceinline__ PackPoint pack_one(const IcpState *__restrict__ st, int mode, const double *__restrict__ src,
                                              double *__restrict__ P, const int32_t *__restrict__ perm, int64_t k,
                                              int64_t N, int32_t *__restrict__ idx_out, double *__restrict__ d2_out,
                                              double r2cut, double lox, double loy, double loz, double hix, double hiy,
                                              double hiz) {
    PackPoint o;
    o.cand = false; o.i = -1; o.sx = o.sy = o.sz = 0.f;
    if (k >= N) return o;
    const int64_t i = perm[k];
    const double *M = mode == 0 ? st->T : st->upd;
    const double *in = mode == 0 ? src : P;
    double x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
    double nx = dadd(dadd(dadd(dmul(M[0], x), dmul(M[1], y)), dmul(M[2], z)), M[3]);
    double ny = dadd(dadd(dadd(dmul(M[4], x), dmul(M[5], y)), dmul(M[6], z)), M[7]);
    double nz = dadd(dadd(dadd(dmul(M[8], x), dmul(M[9], y)), dmul(M[10], z)), M[11]);
    P[3 * i] = nx; P[3 * i + 1] = ny; P[3 * i + 2] = nz;
    double ex = fmax(fmax(lox - nx, nx - hix), 0.0), ey = fmax(fmax(loy - ny, ny - hiy), 0.0),
           ez = fmax(fmax(loz - nz, nz - hiz), 0.0);
    o.cand = (ex * ex + ey * ey + ez * ez) <= r2cut;  // r2cut = r^2 (1 + 1e-12): rounding-safe
    o.i = (int)i;
    o.sx = (float)(nx - st->centroid[0]); o.sy = (float)(ny - st->centroid[1]); o.sz = (float)(nz - st->centroid[2]);
    if (!o.cand) {
        idx_out[i] = -1;
        d2_out[i] = __longlong_as_double(0x7FF0000000000000ll);
    }
    return o;
}
__device__ __forceinline__ void pack_store(const PackPoint &p, int slot, float4 *__restrict__ B, float *__restrict__ eps,
                                           float *__restrict__ S, int32_t *__restrict__ list, float Tn, float T2, float r1,
                                           float mi_factor) {
    B[slot] = make_float4(-2.0f * p.sx, -2.0f * p.sy, -2.0f * p.sz, 1.0f);
    // Error bound of the fp32 surrogate relative to the float64 distance, for points whose
    // nearest neighbour is closer than r1 (see DESIGN.md "NN filter bound"):
    //   eps = 2^-23 * (5 * (2*|s'|_1*Tn + T2) + 2*min(r1, |s'|_1 + Tn)*(Tn + |s'|_1))
    float s1 = fabsf(p.sx) + fabsf(p.sy) + fabsf(p.sz);
    float Mi = 2.0f * s1 * Tn + T2;
    // (mi_factor: 5 for the fp32 MFMA's four rounded products and three sums; 34 for the bf16 form's thirty exact products
    // and up to thirty-one fp32 additions inside the matrix pipe, each charged a full ulp of the largest partial sum)
    eps[slot] = 1.1920929e-7f * (mi_factor * Mi + 2.0f * fminf(r1, s1 + Tn) * (Tn + s1)) * 1.0001f;
    S[slot] = p.sx * p.sx + p.sy * p.sy + p.sz * p.sz;
    list[slot] = p.i;
}
__global__ __launch_bounds__(256) void icp_transform_pack_kernel(
    IcpState *__restrict__ st, int mode, const double *__restrict__ src, double *__restrict__ P,
    const int32_t *__restrict__ perm, int64_t N, float4 *__restrict__ B, float *__restrict__ eps, float *__restrict__ S,
    int32_t *__restrict__ list, float4 *__restrict__ blk_sph, int32_t *__restrict__ idx_out,
    double *__restrict__ d2_out, float Tn, float T2, float r1, double r2cut, double lox, double loy, double loz,
    double hix, double hiy, double hiz, float mi_factor) {
    if (st->done) return;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t k0 = wave * 128 + lane;
    const PackPoint p0 = pack_one(st, mode, src, P, perm, k0, N, idx_out, d2_out, r2cut, lox, loy, loz, hix, hiy, hiz);
    const PackPoint p1 = pack_one(st, mode, src, P, perm, k0 + 64, N, idx_out, d2_out, r2cut, lox, loy, loz, hix, hiy, hiz);
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(p0.cand), m1 = __builtin_amdgcn_ballot_w64(p1.cand);
    const int c0 = __builtin_popcountll(m0), cnt = c0 + __builtin_popcountll(m1);
    if (cnt == 0) return;  // wave-uniform
    int blk = 0;
    if (lane == 0) blk = atomicAdd(&st->n_blocks, 1);
    blk = __builtin_amdgcn_readfirstlane(blk);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int base = blk * 128;
    if (p0.cand) pack_store(p0, base + __builtin_popcountll(m0 & lt), B, eps, S, list, Tn, T2, r1, mi_factor);
    if (p1.cand) pack_store(p1, base + c0 + __builtin_popcountll(m1 & lt), B, eps, S, list, Tn, T2, r1, mi_factor);
    for (int s = cnt + lane; s < 128; s += 64) {  // dummies: never inliers, never selected
        B[base + s] = make_float4(0.f, 0.f, 0.f, 1.f);
        eps[base + s] = 0.f;
        S[base + s] = 3e38f;
        list[base + s] = -1;
    }
    // Bounding spheres of the block's eight 16-slot sub-blocks (centred fp32 coordinates).
    // Consecutive NON-EMPTY cells of the space-filling curve can be far apart (the curve
    // leaves a surface and re-enters it elsewhere), so one sphere per 128 slots can be huge;
    // per sub-block the sweep keeps a target tile only if it is near SOME sub-block.
    __shared__ float stage[4][3][128];
    float (*sg)[128] = stage[threadIdx.x >> 6];
    if (p0.cand) { const int sl = __builtin_popcountll(m0 & lt); sg[0][sl] = p0.sx; sg[1][sl] = p0.sy; sg[2][sl] = p0.sz; }
    if (p1.cand) { const int sl = c0 + __builtin_popcountll(m1 & lt); sg[0][sl] = p1.sx; sg[1][sl] = p1.sy; sg[2][sl] = p1.sz; }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed
    const float big = 3e38f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int sl = half * 64 + lane;
        const bool real = sl < cnt;
        const float x = real ? sg[0][sl] : 0.f, y = real ? sg[1][sl] : 0.f, z = real ? sg[2][sl] : 0.f;
        float lx = real ? x : big, hx = real ? x : -big, ly = real ? y : big, hy = real ? y : -big, lz = real ? z : big,
              hz = real ? z : -big;
#pragma unroll
        for (int off = 1; off <= 8; off <<= 1) {
            lx = fminf(lx, __shfl_xor(lx, off, 64)); hx = fmaxf(hx, __shfl_xor(hx, off, 64));
            ly = fminf(ly, __shfl_xor(ly, off, 64)); hy = fmaxf(hy, __shfl_xor(hy, off, 64));
            lz = fminf(lz, __shfl_xor(lz, off, 64)); hz = fmaxf(hz, __shfl_xor(hz, off, 64));
        }
        if ((lane & 15) == 0) {
            float4 sp = make_float4(0.f, 0.f, 0.f, -1.f);  // empty sub-block: matches nothing
            if (hx >= lx) {
                const float cx = 0.5f * (lx + hx), cy = 0.5f * (ly + hy), cz = 0.5f * (lz + hz);
                const float ex = hx - cx, ey = hy - cy, ez = hz - cz;
                sp = make_float4(cx, cy, cz, sqrtf(ex * ex + ey * ey + ez * ez) * 1.0001f +
                                                 1e-6f * (fabsf(cx) + fabsf(cy) + fabsf(cz)) + 1e-30f);
            }
            blk_sph[(size_t)blk * NN_SB + (sl >> 4)] = sp;
        }
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------ NN sweep (MFMA)
__device__ __forceinline__ void lexmin(double &d, int &j, double od, int oj) {
    if (od < d || (od == d && oj < j)) { d = od; j = oj; }
}

// ---- 1. cull: which target tiles can matter for which scene block (bit mask per block) ----
// Tile t survives for a block iff for some 16-slot sub-block |c_sub - c_tile| <= r + rad_sub +
// rad_tile (bounding spheres, margins included).  A skipped tile has all its points farther
// than r from all points of the block, so it cannot contain the nearest neighbour of an
// INLIER; for a point without any neighbour within r the answer is "no correspondence"
// whichever tiles were visited.  One wave per (block, CULL_WORDS x 64 tiles): lane l tests tile
// base + l, the ballot IS the mask word.  With r = infinity (pedp_nn) every bit is set: the
// dense all-pairs sweep.
__global__ __launch_bounds__(256) void nn_cull_kernel(const IcpState *__restrict__ st, const float4 *__restrict__ tile_sph,
                                                      int n_tiles, int n_words, const float4 *__restrict__ blk_sph,
                                                      float r_search, unsigned long long *__restrict__ mask,
                                                      int32_t *__restrict__ blk_cnt) {
    if (st->done) return;
    const int lane = threadIdx.x & 63;
    const int wg = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int groups = (n_words + CULL_WORDS - 1) / CULL_WORDS;
    const int blk = wg / groups, grp = wg - blk * groups;
    if (blk >= st->n_blocks) return;  // wave-uniform
    float4 bs[NN_SB];
#pragma unroll
    for (int sb = 0; sb < NN_SB; ++sb) bs[sb] = blk_sph[(size_t)blk * NN_SB + sb];
    int w1 = (grp + 1) * CULL_WORDS;
    if (w1 > n_words) w1 = n_words;
    int cnt = 0;
    for (int wi = grp * CULL_WORDS; wi < w1; ++wi) {
        const int t = wi * 64 + lane;
        const float4 ts = tile_sph[t < n_tiles ? t : 0];
        bool keep = false;
#pragma unroll
        for (int sb = 0; sb < NN_SB; ++sb) {
            const float dx = ts.x - bs[sb].x, dy = ts.y - bs[sb].y, dz = ts.z - bs[sb].z;
            const float lim = r_search + bs[sb].w + ts.w;
            keep |= (bs[sb].w >= 0.f) & !((dx * dx + dy * dy + dz * dz) > lim * lim * 1.00001f + 1e-6f);
        }
        keep = keep && (t < n_tiles) && (ts.w >= 0.f);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (lane == 0) mask[(size_t)blk * n_words + wi] = m;
        cnt += __builtin_popcountll(m);
    }
    if (lane == 0 && cnt > 0) atomicAdd(&blk_cnt[blk], cnt);
}

// ---- 2. segments: cut every block's survivor list into pieces of seg_len tiles ----
// One workgroup.  seg_len is chosen so that all pieces fit the segment table (max_segs) and
// is at least SEG_MIN: heavy blocks simply get more pieces, so every sweep wave has the same
// amount of work whatever the spatial distribution.
__global__ __launch_bounds__(1024) void nn_segment_kernel(IcpState *__restrict__ st, int32_t *__restrict__ blk_cnt,
                                                          int32_t *__restrict__ blk_segstart, int32_t *__restrict__ seg_blk,
                                                          int32_t *__restrict__ seg_rank0, int32_t *__restrict__ seg_n,
                                                          int max_segs, int SEG_MIN, int NN_LIST) {
    if (st->done) return;
    __shared__ long long red[16];
    __shared__ int scan[1024];
    __shared__ long long total_s;
    const int nb = st->n_blocks, tid = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int b0 = tid * per, b1 = (b0 + per < nb) ? b0 + per : nb;
    long long loc = 0;
    for (int b = b0; b < b1; ++b) loc += blk_cnt[b];
    long long v = loc;
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
    const long long total = total_s;
    long long room = (long long)max_segs - nb;
    if (room < 1) room = 1;
    long long sl = (total + room - 1) / room;
    if (sl < SEG_MIN) sl = SEG_MIN;
    sl = (sl + NN_TU - 1) / NN_TU * NN_TU;
    if (sl > NN_LIST) sl = NN_LIST;  // cannot happen: the host sizes max_segs for the dense case
    const int seg_len = (int)sl;
    int mine = 0;
    for (int b = b0; b < b1; ++b) mine += (blk_cnt[b] + seg_len - 1) / seg_len;
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
        const int c = blk_cnt[b];
        blk_cnt[b] = 0;  // ready for the next pass
        blk_segstart[b] = at;
        // the block's pieces are made equally long (a 782-unit list is cut 392 + 390, not 424 + 358)
        const int pieces = (c + seg_len - 1) / seg_len;
        const int piece = pieces > 0 ? ((c + pieces - 1) / pieces + NN_TU - 1) / NN_TU * NN_TU : seg_len;
        for (int r0 = 0, k = 0; k < pieces; r0 += piece, ++k) {
            if (at < max_segs) { seg_blk[at] = b; seg_rank0[at] = r0; seg_n[at] = (c - r0 < piece) ? c - r0 : piece; }
            ++at;
        }
    }
    if (tid == 1023) {
        blk_segstart[nb] = scan[1023];
        st->n_segs = scan[1023] < max_segs ? scan[1023] : max_segs;
        st->seg_len = seg_len;
        st->sum_tiles += total;
    }
}

// ---- 3. sweep ----
// Ranks [r0, r0 + n_s) of a block's surviving-unit mask, expanded into an LDS list by one wave
// (prefix popcount over the mask words), followed by 2 G pad units (rows that can never win: the
// last group is filled up with them and the prefetch of the trip after it reads them).
template <int G>
__device__ __forceinline__ void expand_ranks(const unsigned long long *__restrict__ mw, int n_words, int r0, int n_s,
                                             unsigned pad_unit, unsigned *__restrict__ mine, int lane) {
    int running = 0;
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
            const unsigned tile0 = (unsigned)(wg + lane) * 64u;
            while (word != 0ull) {
                const int bit = __builtin_ctzll(word);
                word &= word - 1ull;
                if (rank >= r0 && rank < r0 + n_s) mine[rank - r0] = tile0 + (unsigned)bit;
                ++rank;
            }
        }
        running += __shfl(incl, 63, 64);
    }
    if (lane < 2 * G) mine[n_s + lane] = pad_unit;
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes have landed
}

// The MFMA loop of one wave over the n_s units of its LDS list: keeps, per lane and scene
// sub-block, the best unit value b1, its unit t1 and the second-best unit value b2 (running
// across calls).  Per unit (QT MFMA tiles = 16 QT target rows) the values a lane sees are folded
// with two v_min3 per MFMA, and only once per unit the running triple is updated.  The matrix
// pipe works on the next tile while the VALU folds this one (software pipeline), and the A
// operands of a whole group of G units are fetched one group ahead: G QT x NN_SB MFMAs cover the
// load latency.
template <int QT, int G>
__device__ __forceinline__ void sweep_list(const unsigned *__restrict__ mine, int n_s,
                                           const float *__restrict__ tgtf, int frag, const float (&b)[NN_SB],
                                           float (&b1)[NN_SB], int (&t1)[NN_SB], float (&b2)[NN_SB]) {
    if (n_s <= 0) return;
    constexpr int U = G * QT;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    float vq[NN_SB];
#pragma unroll
    for (int sb = 0; sb < NN_SB; ++sb) vq[sb] = __uint_as_float(0x7F800000u);
    float a[U];
    unsigned units[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        units[g] = mine[g];
#pragma unroll
        for (int u = 0; u < QT; ++u) a[g * QT + u] = tgtf[((size_t)units[g] * QT + u) * 64 + frag];
    }
    f32x4 acc[NN_SB];
#pragma unroll
    for (int sb = 0; sb < NN_SB; ++sb) acc[sb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[sb], zero, 0, 0, 0);
    for (int k = 0; k < n_s; k += G) {
        float an[U];
        unsigned units_n[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            units_n[g] = mine[k + G + g];  // pad units follow the last real one
#pragma unroll
            for (int u = 0; u < QT; ++u) an[g * QT + u] = tgtf[((size_t)units_n[g] * QT + u) * 64 + frag];
        }
#pragma unroll
        for (int t = 0; t < U; ++t) {
            const int u = t % QT;
            const unsigned unit = units[t / QT];
            const float a_next = (t + 1 < U) ? a[t + 1] : an[0];
#pragma unroll
            for (int sb = 0; sb < NN_SB; ++sb) {
                f32x4 nxt = __builtin_amdgcn_mfma_f32_16x16x4f32(a_next, b[sb], zero, 0, 0, 0);
                const f32x4 cur = acc[sb];
#if PEDP_NN_EXPERIMENT == 1   /* MFMA only (wrong results): pure matrix-pipe rate of this loop shape */
                b1[sb] = fminf(b1[sb], cur[0]);
                acc[sb] = nxt;
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
#else
                // two v_min3 per MFMA (linear nesting is what the compiler turns into v_min3)
                if (u == 0) vq[sb] = fminf(fminf(fminf(cur[0], cur[1]), cur[2]), cur[3]);
                else vq[sb] = fminf(fminf(fminf(fminf(vq[sb], cur[0]), cur[1]), cur[2]), cur[3]);
                if (u == QT - 1) {
                    const float v = vq[sb];
                    t1[sb] = v < b1[sb] ? (int)unit : t1[sb];
                    b2[sb] = __builtin_amdgcn_fmed3f(b1[sb], b2[sb], v);  // b1 <= b2: new second best
                    b1[sb] = fminf(b1[sb], v);
                }
                acc[sb] = nxt;
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // 1 MFMA
                if (u == QT - 1) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // then its VALU ops
                else __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
#endif
            }
        }
#pragma unroll
        for (int t = 0; t < U; ++t) a[t] = an[t];
#pragma unroll
        for (int g = 0; g < G; ++g) units[g] = units_n[g];
    }
}

// One wave per segment.
// Triples of segment s: tr_b1 / tr_t1 / tr_b2 [(s * 4 + q) * 128 + slot in block]
template <int QT, int G>
__global__ __launch_bounds__(NN_WAVES * 64) void nn_sweep_kernel(
    const IcpState *__restrict__ st, const float *__restrict__ tgtf /* (n_tiles + pad) x 64, sorted */, int n_tiles,
    int n_words, const unsigned long long *__restrict__ mask, const int32_t *__restrict__ seg_blk,
    const int32_t *__restrict__ seg_rank0, const int32_t *__restrict__ seg_n, const float *__restrict__ srcf /* slots x 4 */,
    float *__restrict__ tr_b1, int32_t *__restrict__ tr_t1, float *__restrict__ tr_b2) {
    __shared__ unsigned surv[NN_WAVES][NN_LIST_TILES / QT + 2 * G];
    if (st->done) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int seg = blockIdx.x * NN_WAVES + wv;
    if (seg >= st->n_segs) return;  // wave-uniform
    const int blk = seg_blk[seg], r0 = seg_rank0[seg], n_s = seg_n[seg];
    const int64_t base = (int64_t)blk * (NN_SB * 16);
    const int frag = (lane & 15) * 4 + (lane >> 4);  // float offset inside a 16-point tile
    unsigned *mine = surv[wv];
    expand_ranks<G>(mask + (size_t)blk * n_words, n_words, r0, n_s, (unsigned)n_tiles, mine, lane);

    float b[NN_SB];
#pragma unroll
    for (int sb = 0; sb < NN_SB; ++sb) b[sb] = srcf[(base + sb * 16) * 4 + frag];
    float b1[NN_SB], b2[NN_SB];
    int t1[NN_SB];
#pragma unroll
    for (int sb = 0; sb < NN_SB; ++sb) { b1[sb] = __uint_as_float(0x7F800000u); b2[sb] = b1[sb]; t1[sb] = n_tiles; }
    sweep_list<QT, G>(mine, n_s, tgtf, frag, b, b1, t1, b2);
    const int q = lane >> 4, j = lane & 15;
#pragma unroll
    for (int sb = 0; sb < NN_SB; ++sb) {
        const size_t o = ((size_t)seg * 4 + q) * (NN_SB * 16) + (size_t)(sb * 16 + j);
        tr_b1[o] = b1[sb];
        tr_t1[o] = t1[sb];
        tr_b2[o] = b2[sb];
    }
}

// ---- 3b. the dense sweep on the bf16 matrix pipe (units of QT = 4 tiles: pedp_nn, large radii, the exhaustive
// configuration).  g(i, j) = |t'_j|^2 - 2 s'_i . t'_j is one K = 32 contraction of v_mfma_f32_16x16x32_bf16 over EXACT
// three-way bf16 pieces (truncation splits: 3 x 8 bits carry an fp32 mantissa): slot k = 9 c + 3 i + j holds piece i of the
// model's t'_c against piece j of the scene's -2 s'_c (27 slots), slots 27..29 the pieces of |t'|^2 against 1, slots 30,
// 31 zero.  Every product is exact; what is left of the error is the pipe's fp32 accumulation of thirty terms, which
// the slot's bound eps charges at a full ulp of the largest partial sum per addition (mi_factor 34 instead of the fp32
// form's 5): g is still a FILTER, winners are re-scored in float64 exactly as before.  16 cycles per MFMA instead of the
// f32-input form's 32, and vector instructions issue beside it (8 of the 16 cycles are free): the fold's three VALU
// operations per MFMA fit.  Same lanes, same triples, same selection and fallback kernels as nn_sweep_kernel.
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
union BfFrag { bf8v v; unsigned short h[8]; uint4 q; };
__device__ __forceinline__ void split3_bf16(float x, unsigned short (&out)[3]) {  // x = hi + mid + lo exactly (truncation)
    const float hi = __uint_as_float(__float_as_uint(x) & 0xFFFF0000u);
    const float r1 = __fsub_rn(x, hi);
    const float mid = __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
    const float lo = __fsub_rn(r1, mid);   // eight significant bits at most: a bf16
    out[0] = (unsigned short)(__float_as_uint(hi) >> 16);
    out[1] = (unsigned short)(__float_as_uint(mid) >> 16);
    out[2] = (unsigned short)(__float_as_uint(lo) >> 16);
}
// A operand: per 16-row tile 64 lanes x 16 B, lane l = (row l & 15, k group l >> 4) holds its eight slots
__global__ void pack_target_bf16_kernel(const float4 *__restrict__ tgt4, int64_t n_rows, uint4 *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * 4) return;
    const int64_t tile = t >> 6;
    const int lane = (int)(t & 63), row = lane & 15, q = lane >> 4;
    const float4 v = tgt4[tile * 16 + row];
    unsigned short pc[4][3];
    split3_bf16(v.x, pc[0]); split3_bf16(v.y, pc[1]); split3_bf16(v.z, pc[2]); split3_bf16(v.w, pc[3]);
    BfFrag f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * q + e;
        f.h[e] = k < 27 ? pc[k / 9][(k % 9) / 3] : (k < 30 ? pc[3][k - 27] : (unsigned short)0);
    }
    out[t] = f.q;
}
#if PEDP_NN_EXPERIMENT == 2   /* timing experiment (wrong results): every wave reads the same unit -- the operand out of L1 */
#define PEDP_BF_UNIT(u) ((u) & 1u)
#else
#define PEDP_BF_UNIT(u) (u)
#endif
template <int QT, int G>
__device__ __forceinline__ void sweep_list_bf16(const unsigned *__restrict__ mine, int n_s, const uint4 *__restrict__ tgtb, int lane,
                                                const bf8v (&b)[NN_SB], float (&b1)[NN_SB], int (&t1)[NN_SB], float (&b2)[NN_SB]) {
    if (n_s <= 0) return;
    constexpr int U = G * QT;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    float vq[NN_SB];
#pragma unroll
    for (int sb = 0; sb < NN_SB; ++sb) vq[sb] = __uint_as_float(0x7F800000u);
    BfFrag a[U];
    unsigned units[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        units[g] = mine[g];
#pragma unroll
        for (int u = 0; u < QT; ++u) a[g * QT + u].q = tgtb[((size_t)PEDP_BF_UNIT(units[g]) * QT + u) * 64 + lane];
    }
    f32x4 acc[NN_SB];
#pragma unroll
    for (int sb = 0; sb < NN_SB; ++sb) acc[sb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[sb], zero, 0, 0, 0);
    for (int k = 0; k < n_s; k += G) {
        BfFrag an[U];
        unsigned units_n[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            units_n[g] = mine[k + G + g];  // pad units follow the last real one
#pragma unroll
            for (int u = 0; u < QT; ++u) an[g * QT + u].q = tgtb[((size_t)PEDP_BF_UNIT(units_n[g]) * QT + u) * 64 + lane];
        }
#pragma unroll
        for (int t = 0; t < U; ++t) {
            const int u = t % QT;
            const unsigned unit = units[t / QT];
            const bf8v a_next = (t + 1 < U) ? a[t + 1].v : an[0].v;
#pragma unroll
            for (int sb = 0; sb < NN_SB; ++sb) {
                f32x4 nxt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_next, b[sb], zero, 0, 0, 0);
                const f32x4 cur = acc[sb];
                if (u == 0) vq[sb] = fminf(fminf(fminf(cur[0], cur[1]), cur[2]), cur[3]);
                else vq[sb] = fminf(fminf(fminf(fminf(vq[sb], cur[0]), cur[1]), cur[2]), cur[3]);
                if (u == QT - 1) {
                    const float v = vq[sb];
                    t1[sb] = v < b1[sb] ? (int)unit : t1[sb];
                    b2[sb] = __builtin_amdgcn_fmed3f(b1[sb], b2[sb], v);
                    b1[sb] = fminf(b1[sb], v);
                }
                acc[sb] = nxt;
#if PEDP_NN_EXPERIMENT != 3
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // 1 MFMA
                if (u == QT - 1) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // then its VALU ops
                else __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
#endif
            }
        }
#pragma unroll
        for (int t = 0; t < U; ++t) a[t] = an[t];
#pragma unroll
        for (int g = 0; g < G; ++g) units[g] = units_n[g];
    }
}
// One wave per segment, like nn_sweep_kernel.  (Measured and not kept: the four waves of a workgroup taking the same piece
// of four consecutive blocks and sharing its A operands through LDS, double-buffered, one barrier per 64 MFMAs of every
// wave -- a quarter of the L2 traffic, 9.2 -> 2.3 GB per sweep, but 1.06-1.22 ms against this kernel's 0.84: the barrier,
// the exposed LDS read at the head of every group and the lost cross-group pipelining cost more than the L2 gave back.)
template <int QT, int G>
__global__ __launch_bounds__(NN_WAVES * 64) void nn_sweep_bf16_kernel(
    const IcpState *__restrict__ st, const uint4 *__restrict__ tgtb /* (n_tiles + pad) x 64 lanes x 16 B */, int n_tiles,
    int n_words, const unsigned long long *__restrict__ mask, const int32_t *__restrict__ seg_blk,
    const int32_t *__restrict__ seg_rank0, const int32_t *__restrict__ seg_n, const float4 *__restrict__ src4 /* slots */,
    float *__restrict__ tr_b1, int32_t *__restrict__ tr_t1, float *__restrict__ tr_b2) {
    __shared__ unsigned surv[NN_WAVES][NN_LIST_TILES / QT + 2 * G];
    if (st->done) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int seg = blockIdx.x * NN_WAVES + wv;
    if (seg >= st->n_segs) return;  // wave-uniform
    const int blk = seg_blk[seg], r0 = seg_rank0[seg], n_s = seg_n[seg];
    const int64_t base = (int64_t)blk * (NN_SB * 16);
    unsigned *mine = surv[wv];
    expand_ranks<G>(mask + (size_t)blk * n_words, n_words, r0, n_s, (unsigned)n_tiles, mine, lane);
    const int q = lane >> 4, j = lane & 15;
    bf8v b[NN_SB];
#pragma unroll
    for (int sb = 0; sb < NN_SB; ++sb) {   // the B operand: pieces of the slot's (-2 x', -2 y', -2 z'), ones for |t'|^2's slots
        const float4 sv = src4[base + sb * 16 + j];
        unsigned short ps[3][3];
        split3_bf16(sv.x, ps[0]); split3_bf16(sv.y, ps[1]); split3_bf16(sv.z, ps[2]);
        BfFrag f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * q + e;
            f.h[e] = k < 27 ? ps[k / 9][k % 3] : (k < 30 ? (unsigned short)0x3F80 : (unsigned short)0);
        }
        b[sb] = f.v;
    }
    float b1[NN_SB], b2[NN_SB];
    int t1[NN_SB];
#pragma unroll
    for (int sb = 0; sb < NN_SB; ++sb) { b1[sb] = __uint_as_float(0x7F800000u); b2[sb] = b1[sb]; t1[sb] = n_tiles; }
    sweep_list_bf16<QT, G>(mine, n_s, tgtb, lane, b, b1, t1, b2);
#pragma unroll
    for (int sb = 0; sb < NN_SB; ++sb) {
        const size_t o = ((size_t)seg * 4 + q) * (NN_SB * 16) + (size_t)(sb * 16 + j);
        tr_b1[o] = b1[sb];
        tr_t1[o] = t1[sb];
        tr_b2[o] = b2[sb];
    }
}

// Diagnostics (tests/test_icp_gpu.py measures the bf16 form's error against float64): g of every (scene row, model row)
// pair as the sweep's MFMA produces it -- one wave per (16 scene rows, 16 model rows), the same operand packing
__global__ __launch_bounds__(64) void nn_bf16_debug_kernel(const uint4 *__restrict__ tgtb, const float4 *__restrict__ src4, int n_src16,
                                                           int n_tgt, float *__restrict__ g_out) {
    const int lane = threadIdx.x, q = lane >> 4, j = lane & 15;
    const int sb = blockIdx.x % n_src16, tile = blockIdx.x / n_src16;
    const float4 sv = src4[(size_t)sb * 16 + j];
    unsigned short ps[3][3];
    split3_bf16(sv.x, ps[0]); split3_bf16(sv.y, ps[1]); split3_bf16(sv.z, ps[2]);
    BfFrag f, a;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * q + e;
        f.h[e] = k < 27 ? ps[k / 9][k % 3] : (k < 30 ? (unsigned short)0x3F80 : (unsigned short)0);
    }
    a.q = tgtb[(size_t)tile * 64 + lane];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, f.v, zero, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {   // D: column = scene slot j, row = model row 4 q + e of the tile
        const int row = tile * 16 + 4 * q + e;
        if (row < n_tgt) g_out[((size_t)sb * 16 + j) * n_tgt + row] = r[e];
    }
}

// ---- 4. exact selection, four threads per slot (thread gl of a slot reads lane group gl of
// every segment of its block): window = min b1 + 2 eps; every (segment, lane group) whose best
// tile is inside the window has its 4 rows re-scored in float64 (the oracle's formula,
// lexicographic (d^2, index) min); a second tile inside the window sends the slot to
// nn_fallback.
template <int QT>
__global__ __launch_bounds__(256) void nn_select_kernel(
    IcpState *__restrict__ st, const int32_t *__restrict__ blk_segstart, const float *__restrict__ tr_b1,
    const int32_t *__restrict__ tr_t1, const float *__restrict__ tr_b2, const double *__restrict__ tgt,
    const int32_t *__restrict__ tperm /* sorted row -> target index */, int64_t Nt, const double *__restrict__ P,
    const float *__restrict__ eps, const float *__restrict__ S, const int32_t *__restrict__ list, float r2f,
    int32_t *__restrict__ idx_out, double *__restrict__ d2_out, int32_t *__restrict__ fb_list) {
    if (st->done) return;
    const int count = st->n_blocks * (NN_SB * 16);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = tid >> 2, gl = tid & 3;
    if ((tid & ~63) >= 4 * count) return;  // whole wave beyond the list
    const int kk = k < count ? k : 0;
    const int i = k < count ? list[kk] : -1;
    const bool live = i >= 0;  // dummies carry -1
    const int blk = kk >> 7, slot = kk & 127;
    const int s0 = blk_segstart[blk], s1 = blk_segstart[blk + 1];
    // everything the slot needs later is requested now, ahead of the dependent loads below
    const float e = eps[kk], Si = S[kk];
    const int64_t ip = live ? i : 0;
    const double px = P[3 * ip], py = P[3 * ip + 1], pz = P[3 * ip + 2];
    const float inf = __uint_as_float(0x7F800000u);
    float m = inf, sm = inf, m2 = inf;  // own best b1, own second-best b1, best b2
    int mt = 0;                         // unit of the own best
    for (int sg = s0; sg < s1; sg += 4) {  // four segments per trip: twelve loads in flight
        float v1[4], v2[4];
        int vt[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = sg + u < s1;
            const size_t o = ((size_t)(in ? sg + u : s1 - 1) * 4 + gl) * (NN_SB * 16) + slot;
            v1[u] = tr_b1[o];
            v2[u] = tr_b2[o];
            vt[u] = tr_t1[o];
            if (!in) { v1[u] = inf; v2[u] = inf; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sm = v1[u] < m ? m : fminf(sm, v1[u]);
            mt = v1[u] < m ? vt[u] : mt;
            m = fminf(m, v1[u]);
            m2 = fminf(m2, v2[u]);
        }
    }
    float mg = fminf(m, __shfl_xor(m, 1, 64)); mg = fminf(mg, __shfl_xor(mg, 2, 64));
    m2 = fminf(m2, __shfl_xor(m2, 1, 64)); m2 = fminf(m2, __shfl_xor(m2, 2, 64));
    const bool maybe = mg + Si <= r2f + 4.0f * e + 4.8e-7f * Si;  // else certainly farther than r
    const float win = mg + 2.0f * e;
    double bd = __longlong_as_double(0x7FF0000000000000ll);
    int bj = 0x7FFFFFFF;
    auto rescore = [&](int unit) {
        const int64_t row0 = (int64_t)unit * (16 * QT) + 4 * gl;  // lane group gl: rows 4gl..4gl+3 of each tile
        for (int u = 0; u < QT; ++u) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + 16 * u + r;
                if (row < Nt) {
                    const int64_t j = tperm[row];
                    lexmin(bd, bj, dist2(px, py, pz, tgt[3 * j], tgt[3 * j + 1], tgt[3 * j + 2]), (int)j);
                }
            }
        }
    };
    if (live && maybe) {
        if (sm <= win) {  // rare: several of this thread's segments have a unit inside the window
            for (int sg = s0; sg < s1; ++sg) {
                const size_t o = ((size_t)sg * 4 + gl) * (NN_SB * 16) + slot;
                if (tr_b1[o] <= win) rescore(tr_t1[o]);
            }
        } else if (m <= win) {
            rescore(mt);
        }
    }
#pragma unroll
    for (int off = 1; off <= 2; off <<= 1) {
        const double od = __shfl_xor(bd, off, 64);
        const int oj = __shfl_xor(bj, off, 64);
        lexmin(bd, bj, od, oj);
    }
    if (live && gl == 0) {
        if (!maybe) {
            idx_out[i] = -1;
            d2_out[i] = __longlong_as_double(0x7FF0000000000000ll);
        } else {
            idx_out[i] = bj;
            d2_out[i] = bd;
            if (m2 <= win) fb_list[atomicAdd(&st->fb_count, 1)] = kk;  // ambiguous: exact search decides
        }
    }
}

// ---- 5. ambiguous slots: exact float64 search, one workgroup (FB_WAVES waves) per slot.
// Candidate tiles are the block's surviving tiles (mask) that also come within r of THIS point
// (lane-parallel sphere test per non-empty mask word); wave w takes the mask words w, w +
// FB_WAVES, ... so the chain of dependent loads per slot is n_words / FB_WAVES long; the waves'
// results meet in LDS.  Rows are scanned 64 at a time.
constexpr int FB_WAVES = 16;
template <int QT>
__global__ __launch_bounds__(FB_WAVES * 64) void nn_fallback_kernel(const IcpState *__restrict__ st,
                                                                   const int32_t *__restrict__ fb_list,
                                                                   const int32_t *__restrict__ list,
                                                                   const unsigned long long *__restrict__ mask, int n_words,
                                                                   const float4 *__restrict__ tile_sph, float r_search,
                                                                   const double *__restrict__ tgt,
                                                                   const int32_t *__restrict__ tperm, int64_t Nt,
                                                                   const double *__restrict__ P,
                                                                   int32_t *__restrict__ idx_out,
                                                                   double *__restrict__ d2_out) {
    if (st->done) return;
    __shared__ double red_d[FB_WAVES];
    __shared__ int red_j[FB_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = st->fb_count;
    const double cx = st->centroid[0], cy = st->centroid[1], cz = st->centroid[2];
    for (int w = blockIdx.x; w < n; w += gridDim.x) {
        const int kk = fb_list[w];
        const int i = list[kk];
        const unsigned long long *mw = mask + (size_t)(kk >> 7) * n_words;
        const double px = P[3 * (int64_t)i], py = P[3 * (int64_t)i + 1], pz = P[3 * (int64_t)i + 2];
        const float sx = (float)(px - cx), sy = (float)(py - cy), sz = (float)(pz - cz);
        const float slack = 1e-5f * (fabsf(sx) + fabsf(sy) + fabsf(sz)) + 1e-6f;  // fp32 rounding of the centred point
        double bd = __longlong_as_double(0x7FF0000000000000ll);
        int bj = 0x7FFFFFFF;
        for (int wi = wave; wi < n_words; wi += FB_WAVES) {
            const unsigned long long word = mw[wi];  // wave-uniform
            if (word == 0ull) continue;
            bool keep = false;
            if ((word >> lane) & 1ull) {
                const float4 ts = tile_sph[wi * 64 + lane];
                const float dx = ts.x - sx, dy = ts.y - sy, dz = ts.z - sz;
                const float lim = r_search + ts.w + slack;
                keep = !((dx * dx + dy * dy + dz * dz) > lim * lim * 1.00001f + 1e-6f);
            }
            unsigned long long near = __builtin_amdgcn_ballot_w64(keep);
            while (near != 0ull) {  // wave-uniform: 64 lanes = 64 rows = 64 / (16 QT) units per trip
                constexpr int UPT = 64 / (16 * QT);  // units per trip
                int unit = -1;
#pragma unroll
                for (int g = 0; g < UPT; ++g) {
                    if (near != 0ull) {
                        const int bit = __builtin_ctzll(near);
                        near &= near - 1ull;
                        if (g == lane / (16 * QT)) unit = wi * 64 + bit;
                    }
                }
                const int64_t row = (int64_t)unit * (16 * QT) + (lane % (16 * QT));
                if (unit >= 0 && row < Nt) {
                    const int64_t j = tperm[row];
                    lexmin(bd, bj, dist2(px, py, pz, tgt[3 * j], tgt[3 * j + 1], tgt[3 * j + 2]), (int)j);
                }
            }
        }
#pragma unroll
        for (int off = 1; off <= 32; off <<= 1) {
            double od = __shfl_xor(bd, off, 64);
            int oj = __shfl_xor(bj, off, 64);
            lexmin(bd, bj, od, oj);
        }
        if (lane == 0) { red_d[wave] = bd; red_j[wave] = bj; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int q = 1; q < FB_WAVES; ++q) lexmin(bd, bj, red_d[q], red_j[q]);
            idx_out[i] = bj;
            d2_out[i] = bd;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ accumulate

// Packet layout.  point-to-plane: [0..20] upper triangle of J J^T (row-major), [21..26] J r,
// [27] sum d^2, [28] count.  point-to-point: [0..2] sum (s-c), [3..5] sum (t-c),
// [6..14] sum (t-c)(s-c)^T, [27], [28] as above.
__global__ __launch_bounds__(ACC_THREADS) void icp_accumulate_kernel(
    const IcpState *__restrict__ st, int estimator, const double *__restrict__ P, int64_t Ns,
    const double *__restrict__ tgt, const double *__restrict__ nrm, int32_t *__restrict__ idx,
    const double *__restrict__ d2, double r2, double *__restrict__ partials /* ACC_BLOCKS x PACKET */) {
    if (st->done) return;
    double acc[PACKET];
#pragma unroll
    for (int k = 0; k < PACKET; ++k) acc[k] = 0.0;
    const double cx = st->centroid[0], cy = st->centroid[1], cz = st->centroid[2];
    for (int64_t i = (int64_t)blockIdx.x * ACC_THREADS + threadIdx.x; i < Ns; i += (int64_t)ACC_BLOCKS * ACC_THREADS) {
        int j = idx[i];
        if (j < 0) continue;
        double dd = d2[i];
        if (!(dd < r2)) { idx[i] = -1; continue; }  // strict, as SearchHybrid's lower_bound
        double sx = P[3 * i], sy = P[3 * i + 1], sz = P[3 * i + 2];
        double tx = tgt[3 * (int64_t)j], ty = tgt[3 * (int64_t)j + 1], tz = tgt[3 * (int64_t)j + 2];
        if (estimator == PEDP_POINT_TO_PLANE) {
            double nx = nrm[3 * (int64_t)j], ny = nrm[3 * (int64_t)j + 1], nz = nrm[3 * (int64_t)j + 2];
            double r = (sx - tx) * nx + (sy - ty) * ny + (sz - tz) * nz;
            double J[6] = {sy * nz - sz * ny, sz * nx - sx * nz, sx * ny - sy * nx, nx, ny, nz};
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int c = a; c < 6; ++c) acc[k++] += J[a] * J[c];
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[21 + a] += J[a] * r;
        } else {
            double s[3] = {sx - cx, sy - cy, sz - cz}, t[3] = {tx - cx, ty - cy, tz - cz};
#pragma unroll
            for (int a = 0; a < 3; ++a) { acc[a] += s[a]; acc[3 + a] += t[a]; }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[6 + 3 * a + c] += t[a] * s[c];
        }
        acc[27] += dd;
        acc[28] += 1.0;
    }
    __shared__ double sh[ACC_THREADS / 64][PACKET];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < PACKET; ++k) {
        double v = wave_sum(acc[k]);
        if (lane == 0) sh[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < PACKET) {
        double v = 0.0;
        for (int w = 0; w < ACC_THREADS / 64; ++w) v += sh[w][threadIdx.x];
        partials[(size_t)blockIdx.x * PACKET + threadIdx.x] = v;
    }
}

__global__ void icp_reduce_kernel(const IcpState *__restrict__ st, const double *__restrict__ partials,
                                  double *__restrict__ packet) {
    if (st->done) return;
    if (threadIdx.x < PACKET) {
        double v = 0.0;
        for (int b = 0; b < ACC_BLOCKS; ++b) v += partials[(size_t)b * PACKET + threadIdx.x];
        packet[threadIdx.x] = v;
    }
}

// ------------------------------------------------------------------ solve (one thread)
__device__ void mat4_mul_dev(const double *A, const double *B, double *C) {
    double R[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
            R[4 * i + j] = s;
        }
    for (int k = 0; k < 16; ++k) C[k] = R[k];
}

__device__ void ident4(double *T) {
    for (int k = 0; k < 16; ++k) T[k] = 0.0;
    T[0] = T[5] = T[10] = T[15] = 1.0;
}

// Eigen-style LDLT (left-looking; pivot = largest remaining original diagonal entry), as
// in oracle/icp.c pedp_oracle_solve6_ldlt.
__device__ bool solve6_ldlt(const double *Ain, const double *b, double *x) {
    const int n = 6;
    // run by one thread; the pivoting indexes these arrays at run time, so they live in LDS
    // (private arrays with dynamic indices would go to scratch memory: ~10x the latency)
    __shared__ double A[6][6], tmp[6], y[6];
    __shared__ int tr[6];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) A[i][j] = Ain[n * i + j];
    for (int k = 0; k < n; ++k) {
        int p = k;
        double big = fabs(A[k][k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(A[i][i]) > big) { big = fabs(A[i][i]); p = i; }
        tr[k] = p;
        if (p != k) {
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
    for (int i = 0; i < n; ++i) y[i] = b[i];
    for (int k = 0; k < n; ++k)
        if (tr[k] != k) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) y[i] -= A[i][j] * y[j];
    for (int i = 0; i < n; ++i) {
        if (fabs(A[i][i]) > 2.2250738585072014e-308) y[i] /= A[i][i];
        else y[i] = 0.0;
    }
    for (int i = n - 1; i >= 0; --i)
        for (int j = i + 1; j < n; ++j) y[i] -= A[j][i] * y[j];
    for (int k = n - 1; k >= 0; --k)
        if (tr[k] != k) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    bool ok = true;
    for (int i = 0; i < n; ++i) {
        x[i] = y[i];
        if (!(y[i] == y[i]) || isinf(y[i])) ok = false;
    }
    return ok;
}

__device__ void vec6_to_T(const double *x, double *T) {
    double ca, sa, cb, sb, cc, sc;  // one argument reduction per angle
    sincos(x[0], &sa, &ca);
    sincos(x[1], &sb, &cb);
    sincos(x[2], &sc, &cc);
    ident4(T);
    T[0] = cc * cb;  T[1] = cc * sb * sa - sc * ca;  T[2] = cc * sb * ca + sc * sa;
    T[4] = sc * cb;  T[5] = sc * sb * sa + cc * ca;  T[6] = sc * sb * ca - cc * sa;
    T[8] = -sb;      T[9] = cb * sa;                 T[10] = cb * ca;
    T[3] = x[3]; T[7] = x[4]; T[11] = x[5];
}

__device__ double det3_dev(const double *M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// 3x3 SVD by one-sided Jacobi (same routine as oracle/icp.c svd3)
__device__ void svd3_dev(const double *Ain, double *U, double *w, double *V) {
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
                off = fmax(off, fabs(gamma) / sqrt(fmax(alpha * beta, 2.2250738585072014e-308)));
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
    double tiny = nrm[ord[0]] * 1e-300 + 2.2250738585072014e-308;
    for (int k = 0; k < 3; ++k) {
        int j = ord[k];
        w[k] = nrm[j];
        for (int i = 0; i < 3; ++i) {
            V[3 * i + k] = Vv[i][j];
            Um[i][k] = (nrm[j] > tiny) ? A[i][j] / nrm[j] : 0.0;
        }
    }
    double rel = 1e-13 * w[0];
    if (w[0] <= tiny) {
        for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) Um[i][k] = (i == k);
    } else {
        if (w[1] <= rel) {
            double a[3] = {Um[0][0], Um[1][0], Um[2][0]};
            int m = (fabs(a[0]) <= fabs(a[1]) && fabs(a[0]) <= fabs(a[2])) ? 0 : (fabs(a[1]) <= fabs(a[2]) ? 1 : 2);
            double e[3] = {0, 0, 0};
            e[m] = 1.0;
            double dt = a[m];
            double b2[3] = {e[0] - dt * a[0], e[1] - dt * a[1], e[2] - dt * a[2]};
            double nb = sqrt(b2[0] * b2[0] + b2[1] * b2[1] + b2[2] * b2[2]);
            for (int i = 0; i < 3; ++i) Um[i][1] = b2[i] / nb;
        }
        if (w[2] <= rel) {
            Um[0][2] = Um[1][0] * Um[2][1] - Um[2][0] * Um[1][1];
            Um[1][2] = Um[2][0] * Um[0][1] - Um[0][0] * Um[2][1];
            Um[2][2] = Um[0][0] * Um[1][1] - Um[1][0] * Um[0][1];
        }
    }
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) U[3 * i + k] = Um[i][k];
}

// pass p (0 = initial correspondence pass).  Records fitness/rmse of the pass, decides
// whether the loop ends, otherwise derives the next update from the packet.
__global__ __launch_bounds__(256) void icp_solve_kernel(IcpState *__restrict__ st, double *__restrict__ packet,
                                                        const double *__restrict__ partials, int pass, int max_iter,
                                                        int estimator, double n_source, double rel_fitness,
                                                        double rel_rmse, double *__restrict__ trace) {
    if (st->done) return;
    // single-GPU runs fold icp_reduce into this launch (partials != null); with an all-reduce
    // hook the packet was reduced (and summed over ranks) before.  Fixed order: 8 slices of 32
    // partials each, then the slices in order -- run-to-run bit-stable.
    __shared__ double pk[32];
    if (partials) {
        __shared__ double slice[8][32];
        const int k = threadIdx.x & 31, part = threadIdx.x >> 5;
        double v = 0.0;
        if (k < PACKET)
            for (int b = part * (ACC_BLOCKS / 8); b < (part + 1) * (ACC_BLOCKS / 8); ++b) v += partials[(size_t)b * PACKET + k];
        slice[part][k] = v;
        __syncthreads();
        if (threadIdx.x < PACKET) {
            double t = 0.0;
            for (int q = 0; q < 8; ++q) t += slice[q][threadIdx.x];
            packet[threadIdx.x] = t;
            pk[threadIdx.x] = t;
        }
    } else if (threadIdx.x < PACKET) {
        pk[threadIdx.x] = packet[threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    packet = pk;  // the serial part below reads the packet from LDS
    st->sum_cand += (long long)st->n_blocks * (NN_SB * 16);
    st->sum_fb += st->fb_count;
    st->fb_count = 0;
    st->n_cand = 0;
    st->n_blocks = 0;

    const double K = packet[28];
    double fit = 0.0, rmse = 0.0;
    if (K > 0.0) { fit = K / n_source; rmse = sqrt(packet[27] / K); }
    st->prev_fitness = st->fitness;
    st->prev_rmse = st->rmse;
    st->fitness = fit;
    st->rmse = rmse;
    if (trace) {
        double *tr = trace + 18 * pass;
        tr[0] = fit; tr[1] = rmse;
        for (int k = 0; k < 16; ++k) tr[2 + k] = st->T[k];
    }
    st->iters = pass;
    if (pass >= max_iter) { st->done = 1; return; }
    if (pass > 0 && fabs(st->prev_fitness - fit) < rel_fitness && fabs(st->prev_rmse - rmse) < rel_rmse) {
        st->done = 1;
        return;
    }
    double upd[16];
    ident4(upd);
    if (K > 0.0) {
        if (estimator == PEDP_POINT_TO_PLANE) {
            double A[36], nb[6], x[6];
            int k = 0;
            for (int a = 0; a < 6; ++a)
                for (int c = a; c < 6; ++c) { A[6 * a + c] = packet[k]; A[6 * c + a] = packet[k]; ++k; }
            for (int a = 0; a < 6; ++a) nb[a] = -packet[21 + a];
            if (solve6_ldlt(A, nb, x)) vec6_to_T(x, upd);
        } else {
            const double *c = st->centroid;
            double ms[3], mt[3], sig[9];
            for (int a = 0; a < 3; ++a) { ms[a] = packet[a] / K; mt[a] = packet[3 + a] / K; }
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) sig[3 * a + b] = packet[6 + 3 * a + b] / K - mt[a] * ms[b];
            double U[9], w[3], V[9];
            svd3_dev(sig, U, w, V);
            double sgn = (det3_dev(U) * det3_dev(V) < 0.0) ? -1.0 : 1.0;
            double R[9];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b)
                    R[3 * a + b] = U[3 * a] * V[3 * b] + U[3 * a + 1] * V[3 * b + 1] + sgn * U[3 * a + 2] * V[3 * b + 2];
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) upd[4 * a + b] = R[3 * a + b];
                double msa[3] = {ms[0] + c[0], ms[1] + c[1], ms[2] + c[2]};
                upd[4 * a + 3] = (mt[a] + c[a]) - (R[3 * a] * msa[0] + R[3 * a + 1] * msa[1] + R[3 * a + 2] * msa[2]);
            }
        }
    }
    for (int k = 0; k < 16; ++k) st->upd[k] = upd[k];
    mat4_mul_dev(upd, st->T, st->T);
}

// ---- RANSAC draws of registration_ransac_based_on_feature_matching (src/pose_estimation.py:482-501)
// One thread per iteration: three correspondences corres[rand()] (with replacement, Open3D's
// Registration.cpp), Umeyama without scaling over the three pairs, then the reference's checkers in
// its order: edge length, distance, normal.  The draw is a counter-based function of (seed,
// iteration) -- Open3D's per-thread mt19937 engines seeded by random_device are not recoverable --
// operation for operation oracle/features.c pedp_oracle_ransac_hypothesis.
__device__ __forceinline__ unsigned long long splitmix64_dev(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ void kabsch3_dev(const double *S, const double *G, double *T) {  // oracle/icp.c pedp_oracle_kabsch, K = 3
    ident4(T);
    double ms[3] = {0, 0, 0}, mt[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) { ms[k] += S[3 * i + k]; mt[k] += G[3 * i + k]; }
    for (int k = 0; k < 3; ++k) { ms[k] /= 3.0; mt[k] /= 3.0; }
    double sig[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; ++i)
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) sig[3 * a + b] += (G[3 * i + a] - mt[a]) * (S[3 * i + b] - ms[b]);
    for (int k = 0; k < 9; ++k) sig[k] /= 3.0;
    double U[9], w[3], V[9];
    svd3_dev(sig, U, w, V);
    const double sgn = (det3_dev(U) * det3_dev(V) < 0.0) ? -1.0 : 1.0;
    double R[9];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            R[3 * a + b] = U[3 * a + 0] * V[3 * b + 0] + U[3 * a + 1] * V[3 * b + 1] + sgn * U[3 * a + 2] * V[3 * b + 2];
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) T[4 * a + b] = R[3 * a + b];
        T[4 * a + 3] = mt[a] - (R[3 * a] * ms[0] + R[3 * a + 1] * ms[1] + R[3 * a + 2] * ms[2]);
    }
}

__device__ __forceinline__ double sqd3(const double *a, const double *b) {
    const double x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
    return (x * x + y * y) + z * z;
}

__global__ __launch_bounds__(64) void ransac_hypothesis_kernel(unsigned long long seed, long long itr0, int count,
                                                               const double *__restrict__ src, const double *__restrict__ src_nrm,
                                                               long long Ns, const double *__restrict__ tgt,
                                                               const double *__restrict__ tgt_nrm, const int32_t *__restrict__ corr,
                                                               double edge, double dist, double cos_thr,
                                                               unsigned char *__restrict__ flags, double *__restrict__ Ts) {
    const int k0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (k0 >= count) return;
    unsigned long long s = splitmix64_dev(seed ^ splitmix64_dev((unsigned long long)(itr0 + k0)));
    long long pick[3];
    double S[9], G[9];
    for (int k = 0; k < 3; ++k) {
        s = splitmix64_dev(s);
        pick[k] = (long long)(s % (unsigned long long)Ns);
        for (int a = 0; a < 3; ++a) {
            S[3 * k + a] = src[3 * pick[k] + a];
            G[3 * k + a] = tgt[3 * (long long)corr[pick[k]] + a];
        }
    }
    double T[16];
    kabsch3_dev(S, G, T);
    bool ok = true;
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j) {
            const double ds = sqrt(sqd3(S + 3 * i, S + 3 * j)), dt = sqrt(sqd3(G + 3 * i, G + 3 * j));
            if (ds < dt * edge || dt < ds * edge) ok = false;
        }
    for (int k = 0; k < 3 && ok; ++k) {
        double p[3];
        for (int a = 0; a < 3; ++a)
            p[a] = ((T[4 * a] * S[3 * k] + T[4 * a + 1] * S[3 * k + 1]) + T[4 * a + 2] * S[3 * k + 2]) + T[4 * a + 3];
        if (sqrt(sqd3(p, G + 3 * k)) > dist) ok = false;
    }
    if (ok && src_nrm && tgt_nrm) {
        for (int k = 0; k < 3; ++k) {
            const double *n = src_nrm + 3 * pick[k], *m = tgt_nrm + 3 * (long long)corr[pick[k]];
            double rn[3];
            for (int a = 0; a < 3; ++a) rn[a] = (T[4 * a] * n[0] + T[4 * a + 1] * n[1]) + T[4 * a + 2] * n[2];
            if ((rn[0] * m[0] + rn[1] * m[1]) + rn[2] * m[2] < cos_thr) ok = false;
        }
    }
    flags[k0] = ok ? 1 : 0;
    for (int k = 0; k < 16; ++k) Ts[16 * (size_t)k0 + k] = T[k];
}

// ---- fused pass, third launch
// The same pivoted LDLT as solve6_ldlt (and oracle/icp.c), operation for operation, with the
// matrix in registers: every index is a compile-time constant after unrolling and the pivot
// exchange of step k is a chain of predicated swaps, one per candidate row.
__device__ __forceinline__ void swap_if(bool c, double &x, double &y) {
    const double t = x;
    x = c ? y : x;
    y = c ? t : y;
}
__device__ bool solve6_ldlt_reg(const double *__restrict__ Ain, const double *__restrict__ b, double *__restrict__ x) {
    // ONE thread runs this; its pivot index is made wave-uniform (readfirstlane: the only active lane), so the
    // exchange of step k is a scalar branch to the one pair of rows and columns concerned instead of a
    // predicated swap for every candidate row (5 + 4 + 3 + 2 + 1 times 12 swaps: a third of the solve).
    double A[6][6], y[6];
    int tr[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) A[i][j] = Ain[6 * i + j];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int p = k;
        double big = fabs(A[k][k]);
#pragma unroll
        for (int i = k + 1; i < 6; ++i)
            if (fabs(A[i][i]) > big) { big = fabs(A[i][i]); p = i; }
        p = __builtin_amdgcn_readfirstlane(p);
        tr[k] = p;
#pragma unroll
        for (int q = k + 1; q < 6; ++q) {
            if (p == q) {  // scalar branch
#pragma unroll
                for (int j = 0; j < 6; ++j) { const double t = A[k][j]; A[k][j] = A[q][j]; A[q][j] = t; }
#pragma unroll
                for (int i = 0; i < 6; ++i) { const double t = A[i][k]; A[i][k] = A[i][q]; A[i][q] = t; }
            }
        }
        if (k > 0) {
            double tmp[6];
#pragma unroll
            for (int j = 0; j < k; ++j) tmp[j] = A[j][j] * A[k][j];
            double sacc = 0.0;
#pragma unroll
            for (int j = 0; j < k; ++j) sacc += A[k][j] * tmp[j];
            A[k][k] -= sacc;
#pragma unroll
            for (int i = k + 1; i < 6; ++i) {
                double u = 0.0;
#pragma unroll
                for (int j = 0; j < k; ++j) u += A[i][j] * tmp[j];
                A[i][k] -= u;
            }
        }
        const double akk = A[k][k];
        if (fabs(akk) > 0.0) {
#pragma unroll
            for (int i = k + 1; i < 6; ++i) A[i][k] /= akk;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = b[i];
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int q = k + 1; q < 6; ++q)
            if (tr[k] == q) { const double t = y[k]; y[k] = y[q]; y[q] = t; }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j) y[i] -= A[i][j] * y[j];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        if (fabs(A[i][i]) > 2.2250738585072014e-308) y[i] /= A[i][i];
        else y[i] = 0.0;
    }
#pragma unroll
    for (int i = 5; i >= 0; --i)
#pragma unroll
        for (int j = i + 1; j < 6; ++j) y[i] -= A[j][i] * y[j];
#pragma unroll
    for (int k = 5; k >= 0; --k)
#pragma unroll
        for (int q = k + 1; q < 6; ++q)
            if (tr[k] == q) { const double t = y[k]; y[k] = y[q]; y[q] = t; }
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        x[i] = y[i];
        if (!(y[i] == y[i]) || isinf(y[i])) ok = false;
    }
    return ok;
}

// ================================================================== fused pass
// Radius-limited registrations (unit size 1) run ONE launch per correspondence pass:
//   icp_pass_kernel    one workgroup (8 waves) per live scene chunk.  Every WAVE owns one 16-slot
//                      sub-block of the chunk from the transform to the partial sums and never
//                      waits for another wave on the way: transform, box test, compaction into the
//                      wave's slots, the sub-block's bounding sphere, two-level culling of the target
//                      tiles against THAT sphere, MFMA sweep of the survivors (one MFMA per tile,
//                      the two best tiles per lane in registers), exact float64 selection, the
//                      sub-block's partial sums of J^T J / J^T r by a fixed shuffle tree.  The eight
//                      waves meet once, to add their sums in order.  The workgroup that finishes
//                      last (a ticket) adds the live chunks' partial sums in ascending chunk order,
//                      solves the 6x6 system, updates the pose, the convergence test, the motion
//                      bound and the live list (icp_finish_body).
//   icp_finish_kernel  the same body as a launch of its own: only where an exchange step (scene
//                      sharded over ranks) sits between the sum and the solve.
// Round 2 ran the chunk as eight cooperating waves with ten workgroup barriers, its triples and tile
// lists in LDS, and the finish as a second launch: 36 + 10 us per pass of which a few hundred
// nanoseconds were arithmetic.
//
// A chunk is 128 consecutive entries of the scene's spatial order.  Only LIVE chunks are visited:
// those that had a point within r + margin of the target's bounding box when the live set was
// last rebuilt.  A rebuild pass walks the whole scene and recomputes every point from the source
// through the history of updates -- the same float64 operations, in the same order, as applying
// them pass by pass, so a point's coordinates do not depend on when its chunk became live.
// Between rebuilds a point outside the live chunks is farther than r + margin from the box; an
// update (R, t) moves a point x by at most |R - I| |x - c| + |t + (R - I) c| (c = box centre), and
// with e = (distance to the box) + rho (rho = half diagonal), E = r + margin + rho:
//   e_new >= e (1 - theta) - tau   =>   e_n >= E - (Theta E + Tau) = E - mu,
// so no such point can come within r while mu < margin; the finish requests a rebuild at
// mu >= 0.95 margin.  Chunk ids, slot order, tile order and the order of the partial sums depend
// only on the data, never on execution order: results are run-to-run bit-stable.
//
// Hand-over of the partial sums inside the launch (MI355X: per-XCD L2s are not coherent, a CU's L1
// is never refreshed): every partial is stored write-through (sc1), every storing wave drains its
// stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, ONE lane takes an agent-scope
// ticket; the workgroup whose ticket is the last reads every partial with sc1 loads (they bypass
// its L1).  No fence: a release fence per workgroup writes the XCD's L2 back and took the pass from
// 36 to 92 us in round 2.  The live mask words are only ever touched by agent-scope atomics.
constexpr int CH = NN_SB * 16;   // 128 scene points per chunk = slots per scene block
constexpr int BK_W = 8;          // waves of a pass workgroup = sub-blocks of a chunk
constexpr int BK_WCAP = BK_W * 64;  // mask words (64 tiles each) the fused pass handles: 524,288 target points
constexpr int PSTRIDE = 32;      // doubles per chunk in the partials: packet, [29] wave-tiles swept, [30] exact searches
constexpr int LIVE_CAP = 8192;   // live chunks the finish kernel lists in LDS (1M scene points)
constexpr int WTL = 1024;        // tiles a wave lists before it sweeps them
constexpr int L2_WORDS = 8;      // mask words whose tile spheres a wave requests at once
constexpr int SW_G = 8;          // tiles per group of the sweep (A fragments fetched one group ahead)

__device__ __forceinline__ void xform(const double *__restrict__ M, double &x, double &y, double &z) {
    const double nx = dadd(dadd(dadd(dmul(M[0], x), dmul(M[1], y)), dmul(M[2], z)), M[3]);
    const double ny = dadd(dadd(dadd(dmul(M[4], x), dmul(M[5], y)), dmul(M[6], z)), M[7]);
    const double nz = dadd(dadd(dadd(dmul(M[8], x), dmul(M[9], y)), dmul(M[10], z)), M[11]);
    x = nx; y = ny; z = nz;
}

// Pointers that reach a kernel through the argument block parked in LDS have no address space the compiler
// could infer: every access through them came out as a FLAT operation, which counts on the LDS counter as
// well as on the memory counter -- each LDS read of a list entry then waited for the loads still in flight
// (`s_waitcnt vmcnt(0) lgkmcnt(0)` in front of the sweep's MFMAs), so nothing was prefetched at all.  The
// pass kernel states the address space where it dereferences them.
#define PEDP_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ PEDP_GLOBAL T *as_global(T *p) {
    return (PEDP_GLOBAL T *)(uintptr_t)p;
}
__device__ __forceinline__ float4 gload4(const float4 *p) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 v = *(const PEDP_GLOBAL v4 *)(uintptr_t)p;
    return make_float4(v[0], v[1], v[2], v[3]);
}
template <typename P>
__device__ __forceinline__ void xform_g(P M, double &x, double &y, double &z) {  // xform through a global pointer
    const double nx = dadd(dadd(dadd(dmul(M[0], x), dmul(M[1], y)), dmul(M[2], z)), M[3]);
    const double ny = dadd(dadd(dadd(dmul(M[4], x), dmul(M[5], y)), dmul(M[6], z)), M[7]);
    const double nz = dadd(dadd(dadd(dmul(M[8], x), dmul(M[9], y)), dmul(M[10], z)), M[11]);
    x = nx; y = ny; z = nz;
}

struct PassArgs {
    // scene
    const double *src;          // N x 3 source points
    const int32_t *perm;        // spatial order
    int64_t N;
    int n_chunks;
    double *hist;               // [pass + 1][16]: init, then the update of every pass so far
    double *Pk;                 // 2 x N_pad x 3: transformed points in spatial order (live chunks); pass p reads copy p & 1, writes the other
    double *Tprev;              // 2 x N_pad x 3: last pass's nearest neighbour of the point at that position (x = NaN: none); likewise
    size_t pp_stride;           // doubles between the two copies
    unsigned long long *live;   // live mask (n_lw words), behind it the mask before the last rebuild (n_lw words)
    const double *chunk_sph;    // [chunk][8][4]: bounding spheres of the chunk's eight 16-point runs in the source frame
    int32_t *live_list;         // live chunks ascending (valid outside rebuild passes)
    // target
    const float *tgtf;          // sorted target operand, 64 floats per 16-row tile
    int n_tiles, n_words;
    const float4 *tile_sph, *word_sph;
    const double *tgt_s;        // sorted target rows, float64 x 6: x y z nx ny nz
    const int32_t *tperm;       // sorted row -> target index
    int64_t Nt;
    const double *tgt, *nrm;
    // parameters (the radius-dependent ones live in IcpState)
    float Tn, T2;
    double lo[3], hi[3];
    int estimator;
    int32_t *idx_out;
    double *partials;           // n_chunks x PSTRIDE: by chunk id in a rebuild pass, by live rank otherwise
    // Several start poses of one (scene, target) pair share a launch: pose b = blockIdx.y owns the
    // state and the per-pose buffers (Pk, Tprev, live, live_list, hist, idx_out, partials, packet)
    // b * pose_stride bytes behind pose 0's.
    size_t pose_stride;
    // the finish inside the launch (fuse != 0)
    int fuse, n_lw;
    unsigned *ticket;           // workgroups of the running launch that are through (the last one closes the pass); a line of its own:
                                // 512 atomics on the state's line held up every wave's reads of the state
    double *packet, *trace;
    double bc[3];               // centre of the target's box (motion bound)
    // Which workgroup visits which live chunk (single registration only).  With more live chunks than CUs the
    // dispatcher puts workgroups n_cu + k and k on one CU, and two workgroups on a CU run a third slower than one
    // alone: the pass ended with the pairs.  The workgroup that is through FIRST (ticket 0; it has ten microseconds
    // to spare) ranks the chunks by the durations the pass before measured and hands the lightest 2 (n_live - n_cu)
    // of them to the positions that share a CU, the lightest with the heaviest of those.  Only who works on a chunk
    // changes -- partial sums stay indexed by live rank, so no result bit does.
    int4 *visit;                // [2][visit_cap]: (live rank, chunk, pass + 1 it is meant for, n_live) for workgroup b of pass p at [p & 1][b]
    int2 *dur;                  // [2][visit_cap]: (cycles, pass + 1 that measured them) by live rank, at [p & 1][rank]
    int visit_cap, n_cu;
};

template <typename T>
__device__ __host__ __forceinline__ T *pose_ptr(T *p, size_t bytes) {
    return (T *)((char *)p + bytes);
}
template <typename T>
__device__ __host__ __forceinline__ const T *pose_ptr(const T *p, size_t bytes) {
    return (const T *)((const char *)p + bytes);
}

// ------------------------------------------------------------------ finish
// phase 0: sum the live chunks' partials, solve, update (one GPU); phase 1: sum only (the packet
// then goes through the all-reduce); phase 2: solve from the summed packet.
// Order of the sum: 32 contiguous ranges of the live chunks, ascending inside a range, then the
// ranges in order -- a function of the live set alone, whatever the number of threads.
struct FinishArgs {
    unsigned long long *live;
    int32_t *live_list;
    int n_lw;
    const double *partials;
    double *packet;
    int phase, estimator;
    double *trace, *hist;
    double bcx, bcy, bcz;
    // in-launch finish: sixteen counters (32 words apart) on which the launch's workgroups without a chunk sign off;
    // the state is rewritten only once all n_idle of them have
    unsigned *idle = nullptr;
    int n_idle = 0, n_busy = 0;
    // in-launch finish: what the closing workgroup read of the state when the launch began (pass, rebuild flag, live
    // count do not change inside a pass) -- no second, dependent read in front of the partial sums' loads
    int known = 0, k_pass = 0, k_rebuild = 0, k_n_live = 0;
};
template <int NT, int LCAP>
struct FinishLds {
    double slice[32][32], pk[32];
    int scan[NT], lst[LCAP];
    int do_rebuild, n_live_s;
};
// COHERENT: the partial sums and the live mask were written earlier in THIS launch by other
// workgroups (sc1 stores / atomics): read them past this CU's L1 -- global_load ... sc1, never a
// flat_ load (the pointers come out of the LDS-parked argument block, so the address space is
// stated here).
typedef __attribute__((address_space(1))) unsigned long long g_u64;
typedef __attribute__((address_space(1))) int g_i32;
typedef __attribute__((address_space(1))) unsigned g_u32;
__device__ __forceinline__ double load_sc1(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load((g_u64 *)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ int load_sc1(const int32_t *p) {
    return __hip_atomic_load((g_i32 *)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_sc1(double *p, double v) {
    __hip_atomic_store((g_u64 *)(uintptr_t)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool COHERENT>
__device__ __forceinline__ double load_partial(const double *p) {
    return COHERENT ? load_sc1(p) : *p;
}
template <bool COHERENT>
__device__ __forceinline__ unsigned long long load_live(unsigned long long *p) {
    if (COHERENT) return __hip_atomic_fetch_or((g_u64 *)(uintptr_t)p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

template <int NT, int LCAP, bool COHERENT>
__device__ __forceinline__ void icp_finish_body(IcpState *st, const FinishArgs &f, FinishLds<NT, LCAP> &L, const int tid) {
    const int pass = f.known ? f.k_pass : st->pass, max_iter = st->max_iter;
    const double n_source = st->n_source, rel_fitness = st->rel_fitness, rel_rmse = st->rel_rmse, reachE = st->reachE,
                 margin = st->margin;
    constexpr int PARTS = 32, TPARTS = NT / 32, PPT = PARTS / TPARTS;  // ranges; ranges in flight; ranges per thread
    static_assert(NT % 32 == 0 && PARTS % TPARTS == 0, "thread count");
    if (tid == 0) L.do_rebuild = 0;
    if (tid == 0) PEDP_STAMP(2, 0, 0);
    if (tid == 0 && pass == 5) PEDP_STAMP(2, 3, 0);
    // what the solving thread needs of the state is requested now, ahead of the sums
    double T0[16], fit0 = 0.0, rmse0 = 0.0, mu_th0 = 0.0, mu_ta0 = 0.0;
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) T0[k] = st->T[k];
        fit0 = st->fitness; rmse0 = st->rmse; mu_th0 = st->mu_theta; mu_ta0 = st->mu_tau;
    }
#if PEDP_ICP_STAMPS
    if (tid == 0) { g_icp_stamps[2][1][0] = (long long)__builtin_amdgcn_s_memtime(); g_icp_stamps[2][1][1] = (long long)__builtin_amdgcn_s_memrealtime(); }
#endif
    if (f.phase != 2) {
        // Partial sums are indexed by chunk id in a rebuild pass and by live rank otherwise; either
        // way they are summed in ascending chunk order.
        const bool listing = f.known ? f.k_rebuild != 0 : st->rebuild != 0;
        int n_live = f.known ? f.k_n_live : st->n_live;
        bool listed = true;
        if (listing) {
            // the new live list, ascending: thread t owns a contiguous range of mask words
            const int per = (f.n_lw + NT - 1) / NT;
            const int w_lo = tid * per < f.n_lw ? tid * per : f.n_lw, w_hi = w_lo + per < f.n_lw ? w_lo + per : f.n_lw;
            int mine = 0;
            unsigned long long first = 0ull;  // (per == 1 for scenes up to 64 NT chunks: the word is read once)
            for (int wi = w_lo; wi < w_hi; ++wi) {
                const unsigned long long word = load_live<COHERENT>(&f.live[wi]);
                if (wi == w_lo) first = word;
                mine += __builtin_popcountll(word);
            }
            L.scan[tid] = mine;
            __syncthreads();
            for (int off = 1; off < NT; off <<= 1) {
                const int t = tid >= off ? L.scan[tid - off] : 0;
                __syncthreads();
                L.scan[tid] += t;
                __syncthreads();
            }
            n_live = L.scan[NT - 1];
            listed = n_live <= LCAP;
            int at = L.scan[tid] - mine;
            for (int wi = w_lo; wi < w_hi; ++wi) {
                unsigned long long word = wi == w_lo ? first : load_live<COHERENT>(&f.live[wi]);
                while (word != 0ull) {
                    const int chunk = wi * 64 + __builtin_ctzll(word);
                    word &= word - 1ull;
                    if (listed) L.lst[at] = chunk;
                    as_global(f.live_list)[at] = chunk;
                    ++at;
                }
            }
            __syncthreads();
        }
        // PARTS contiguous ranges of the live chunks, ascending inside a range, then the ranges in
        // order.  Sixteen loads in flight per thread and range; branch-free (an entry beyond the range
        // reads the range's last chunk again and is not added), the three ways to a chunk's position in
        // the partials -- live rank; list in LDS; list in memory, written a moment ago by this workgroup:
        // read past L1 -- are told apart outside the loops.
        const int k = tid & 31;
        const int lper = (n_live + PARTS - 1) / PARTS;
        auto sum_ranges = [&](auto position) {
            if constexpr (PPT == 2) {
                // 512 threads, 32 ranges x 32 entries: a thread takes TWO ENTRIES OF ONE range (not one entry of two
                // ranges, one range after the other): every load of the pass's sum is requested in one go -- one round
                // trip past the L2 instead of two -- with the registers the two-range form used (ten loads per entry and
                // round: a range is 9 chunks at the bench frame's 281 live chunks; each entry's additions in the same order)
                constexpr int BW = 10;
                const int part = tid >> 4, k0 = (tid & 15) * 2, k1 = k0 + 1;
                const bool has1 = k1 < PACKET + 2;
                const int l_lo = part * lper < n_live ? part * lper : n_live, l_hi = l_lo + lper < n_live ? l_lo + lper : n_live;
                double v0 = 0.0, v1 = 0.0;
                for (int q = l_lo; q < l_hi; q += BW) {
                    int at[BW];
                    double x0[BW], x1[BW];
#pragma unroll
                    for (int u = 0; u < BW; ++u) at[u] = position(q + u < l_hi ? q + u : l_hi - 1);
#pragma unroll
                    for (int u = 0; u < BW; ++u) {
                        x0[u] = load_partial<COHERENT>(&f.partials[(size_t)at[u] * PSTRIDE + k0]);
                        x1[u] = has1 ? load_partial<COHERENT>(&f.partials[(size_t)at[u] * PSTRIDE + k1]) : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < BW; ++u) {
                        v0 = q + u < l_hi ? v0 + x0[u] : v0;
                        v1 = q + u < l_hi ? v1 + x1[u] : v1;
                    }
                }
                L.slice[part][k0] = v0;
                L.slice[part][k1] = v1;
                return;
            }
#pragma unroll
            for (int pp = 0; pp < PPT; ++pp) {
                const int part = (tid >> 5) + pp * TPARTS;
                const int l_lo = part * lper < n_live ? part * lper : n_live, l_hi = l_lo + lper < n_live ? l_lo + lper : n_live;
                double v = 0.0;
                if (k < PACKET + 2) {
                    for (int q = l_lo; q < l_hi; q += 16) {
                        int at[16];
                        double x[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) at[u] = position(q + u < l_hi ? q + u : l_hi - 1);
#pragma unroll
                        for (int u = 0; u < 16; ++u) x[u] = load_partial<COHERENT>(&f.partials[(size_t)at[u] * PSTRIDE + k]);
#pragma unroll
                        for (int u = 0; u < 16; ++u) v = q + u < l_hi ? v + x[u] : v;
                    }
                }
                L.slice[part][k] = v;
            }
        };
        if (!listing) sum_ranges([&](int q) { return q; });
        else if (listed) sum_ranges([&](int q) { return L.lst[q]; });
        else sum_ranges([&](int q) { return load_sc1(&f.live_list[q]); });
        if (tid == 0 && pass == 5) PEDP_STAMP(2, 3, 1);
        if (tid == 0) L.n_live_s = n_live;
        __syncthreads();
        if (tid == 0 && pass == 5) PEDP_STAMP(2, 3, 2);
        if (tid < 32) {
            double t = 0.0;
            for (int q = 0; q < PARTS; ++q) t += L.slice[q][tid];
            L.pk[tid] = t;
            if (tid < PACKET) as_global(f.packet)[tid] = t;
        }
        __syncthreads();
        if (tid == 0 && f.phase == 1) {  // (phase 0 writes these further down, with the rest of the state)
            st->sum_tiles += (long long)L.pk[PACKET];
            st->sum_fb += (long long)L.pk[PACKET + 1];
            st->n_live = L.n_live_s;
        }
        if (tid == 0) {
#if PEDP_ICP_STAMPS
            g_icp_stamps[2][1][2] = (long long)__builtin_amdgcn_s_memtime(); g_icp_stamps[2][1][3] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
        }
        if (f.phase == 1) return;
    } else {
        if (tid < PACKET) L.pk[tid] = as_global(f.packet)[tid];
        __syncthreads();
    }
    if (COHERENT && f.n_idle > 0 && tid < 64) {  // normally true at the first look
        for (unsigned spins = 0;; ++spins) {
            unsigned c = tid < 16 ? __hip_atomic_load((g_u32 *)(uintptr_t)(f.idle + 32 * tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
            if (__shfl(c, 0, 64) - st->idle_base >= (unsigned)f.n_idle) break;
            if (spins > (1u << 22)) {  // bounded: a lost workgroup must not hang the device -- but the pass is NOT closed over
                if (tid == 0) L.do_rebuild = -1;   // workgroups that may still read the old state: the registration fails loudly
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    if (COHERENT && f.n_idle > 0) {
        __syncthreads();
        if (L.do_rebuild == -1) {  // (workgroup-uniform)
            if (tid == 0) st->done = -1;   // icp_collect / the batch driver turn this into PEDP_ERR_HIP
            return;
        }
    }
    if (tid == 0) {
        PEDP_STAMP(2, 0, 1);
        if (pass == 5) PEDP_STAMP(2, 3, 3);
        if (COHERENT) { st->ticket_base += (unsigned)f.n_busy; st->idle_base += (unsigned)f.n_idle; }
        if (f.phase == 0) {
            st->sum_tiles += (long long)L.pk[PACKET];
            st->sum_fb += (long long)L.pk[PACKET + 1];
            st->n_live = L.n_live_s;
        }
        L.do_rebuild = 0;
        const double *pk = L.pk;
        const double K = pk[28];
        double fit = 0.0, rmse = 0.0;
        if (K > 0.0) { fit = K / n_source; rmse = sqrt(pk[27] / K); }
        st->prev_fitness = fit0;
        st->prev_rmse = rmse0;
        st->fitness = fit;
        st->rmse = rmse;
        if (f.trace) {
            PEDP_GLOBAL double *tr = as_global(f.trace) + 18 * pass;
            tr[0] = fit; tr[1] = rmse;
            for (int k = 0; k < 16; ++k) tr[2 + k] = T0[k];
        }
        st->iters = pass;
        bool stop = pass >= max_iter;
        if (pass > 0 && fabs(fit0 - fit) < rel_fitness && fabs(rmse0 - rmse) < rel_rmse) stop = true;
        if (stop) {
            st->done = 1;
        } else {
            double upd[16];
            ident4(upd);
            PEDP_STAMP(2, 2, 0);
            if (pass == 5) PEDP_STAMP(2, 3, 4);
            if (K > 0.0) {
                if (f.estimator == PEDP_POINT_TO_PLANE) {
                    double A[36], nb[6], x[6];
                    int k = 0;
#pragma unroll
                    for (int u = 0; u < 6; ++u)
#pragma unroll
                        for (int v = u; v < 6; ++v) { A[6 * u + v] = pk[k]; A[6 * v + u] = pk[k]; ++k; }
#pragma unroll
                    for (int u = 0; u < 6; ++u) nb[u] = -pk[21 + u];
                    const bool ok = solve6_ldlt_reg(A, nb, x);
                    PEDP_STAMP(2, 2, 1);
                    if (ok) vec6_to_T(x, upd);
                    PEDP_STAMP(2, 2, 2);
                } else {
                    const double *c = st->centroid;
                    double ms[3], mt[3], sig[9];
                    for (int u = 0; u < 3; ++u) { ms[u] = pk[u] / K; mt[u] = pk[3 + u] / K; }
                    for (int u = 0; u < 3; ++u)
                        for (int v = 0; v < 3; ++v) sig[3 * u + v] = pk[6 + 3 * u + v] / K - mt[u] * ms[v];
                    double U[9], w[3], V[9];
                    svd3_dev(sig, U, w, V);
                    const double sgn = (det3_dev(U) * det3_dev(V) < 0.0) ? -1.0 : 1.0;
                    double R[9];
                    for (int u = 0; u < 3; ++u)
                        for (int v = 0; v < 3; ++v)
                            R[3 * u + v] = U[3 * u] * V[3 * v] + U[3 * u + 1] * V[3 * v + 1] + sgn * U[3 * u + 2] * V[3 * v + 2];
                    for (int u = 0; u < 3; ++u) {
                        for (int v = 0; v < 3; ++v) upd[4 * u + v] = R[3 * u + v];
                        const double msa[3] = {ms[0] + c[0], ms[1] + c[1], ms[2] + c[2]};
                        upd[4 * u + 3] = (mt[u] + c[u]) - (R[3 * u] * msa[0] + R[3 * u + 1] * msa[1] + R[3 * u + 2] * msa[2]);
                    }
                }
            }
            PEDP_STAMP(2, 0, 2);
            if (pass == 5) PEDP_STAMP(2, 3, 5);
            for (int k = 0; k < 16; ++k) { st->upd[k] = upd[k]; as_global(f.hist)[16 * (pass + 1) + k] = upd[k]; }
            {
                double Tn[16];
                mat4_mul_dev(upd, T0, Tn);
                for (int k = 0; k < 16; ++k) st->T[k] = Tn[k];
            }
            // how far this update can move a point near the target: |R - I|_F (>= the spectral norm) and
            // |t + (R - I) c| about the box centre c
            double th2 = 0.0, tv[3];
            for (int u = 0; u < 3; ++u) {
                tv[u] = upd[4 * u + 3];
                const double cc[3] = {f.bcx, f.bcy, f.bcz};
                for (int v = 0; v < 3; ++v) {
                    const double dlt = upd[4 * u + v] - (u == v ? 1.0 : 0.0);
                    th2 += dlt * dlt;
                    tv[u] += dlt * cc[v];
                }
            }
            double mu_th = mu_th0 + sqrt(th2), mu_ta = mu_ta0 + sqrt(tv[0] * tv[0] + tv[1] * tv[1] + tv[2] * tv[2]);
            const double mu = mu_th * reachE + mu_ta;
            if (!(mu < 0.95 * margin)) {  // also when mu is NaN
                L.do_rebuild = 1;
                mu_th = 0.0;
                mu_ta = 0.0;
                st->n_rebuilds += 1;
            }
            st->mu_theta = mu_th;
            st->mu_tau = mu_ta;
            st->rebuild = L.do_rebuild;
            st->pass = pass + 1;
            PEDP_STAMP(2, 2, 3);
            if (pass == 5) PEDP_STAMP(2, 3, 6);
        }
        PEDP_STAMP(2, 0, 3);
    }
    __syncthreads();
    if (L.do_rebuild)  // the next pass lists the live chunks anew; it resets what the old ones leave behind
        for (int wi = tid; wi < f.n_lw; wi += NT) { as_global(f.live)[f.n_lw + wi] = load_live<COHERENT>(&f.live[wi]); as_global(f.live)[wi] = 0ull; }
}

constexpr int FIN_THREADS = 1024;
__global__ __launch_bounds__(FIN_THREADS) void icp_finish_kernel(IcpState *st, unsigned long long *live, int32_t *live_list,
                                                         int n_lw, const double *partials, double *packet, int phase, int estimator,
                                                         double *__restrict__ trace, double *__restrict__ hist,
                                                         double bcx, double bcy, double bcz, size_t pose_stride) {
    {   // pose b = blockIdx.x of a batch: its state and buffers are b * pose_stride bytes behind pose 0's
        const size_t off = (size_t)blockIdx.x * pose_stride;
        st = pose_ptr(st, off); live = pose_ptr(live, off); live_list = pose_ptr(live_list, off);
        partials = pose_ptr(partials, off); packet = pose_ptr(packet, off); hist = pose_ptr(hist, off);
    }
    if (st->done) return;
    __shared__ FinishLds<FIN_THREADS, LIVE_CAP> L;
    FinishArgs f;
    f.live = live; f.live_list = live_list; f.n_lw = n_lw; f.partials = partials; f.packet = packet; f.phase = phase;
    f.estimator = estimator; f.trace = trace; f.hist = hist; f.bcx = bcx; f.bcy = bcy; f.bcz = bcz;
    icp_finish_body<FIN_THREADS, LIVE_CAP, false>(st, f, L, threadIdx.x);
}

// ------------------------------------------------------------------ pass
// exact float64 scan of the rows of the tiles in `near` (one bit per lane's tile), 64 rows per trip
__device__ __forceinline__ void scan_near_tiles(unsigned long long near, int unit_of_lane, const PassArgs &a, double qx,
                                                double qy, double qz, int lane, double &bd, int &bj) {
    while (near != 0ull) {  // wave-uniform: 64 lanes = 64 rows = 4 tiles per trip
        int unit = -1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (near != 0ull) {
                const int bit = __builtin_ctzll(near);
                near &= near - 1ull;
                const int u = __shfl(unit_of_lane, bit, 64);
                if (g == (lane >> 4)) unit = u;
            }
        }
        const int64_t row = (int64_t)unit * 16 + (lane & 15);
        if (unit >= 0 && row < a.Nt)
            lexmin(bd, bj, dist2(qx, qy, qz, as_global(a.tgt_s)[6 * row], as_global(a.tgt_s)[6 * row + 1], as_global(a.tgt_s)[6 * row + 2]), as_global(a.tperm)[row]);
    }
}

// Butterfly partner inside a row of 16 lanes by DPP -- a modifier on a move, no round trip through the LDS
// crossbar like ds_bpermute.  LEVEL 0: lane ^ 1, 1: lane ^ 2 (quad permutes); 2, 3: lane 7 - i of the half row /
// 15 - i of the row, i.e. SOME lane of the partner's group: in a symmetric reduction every lane of that group
// holds what the partner holds once the lower levels are done, so the result is the xor butterfly's, bit for bit.
template <int LEVEL>
__device__ __forceinline__ int row_partner(int v) {
    constexpr int ctrl = LEVEL == 0 ? 0xB1 : (LEVEL == 1 ? 0x4E : (LEVEL == 2 ? 0x141 : 0x140));
    return __builtin_amdgcn_update_dpp(v, v, ctrl, 0xF, 0xF, false);
}
template <int LEVEL>
__device__ __forceinline__ float row_partner(float v) { return __int_as_float(row_partner<LEVEL>(__float_as_int(v))); }
template <int LEVEL>
__device__ __forceinline__ double row_partner(double v) {
    return __hiloint2double(row_partner<LEVEL>(__double2hiint(v)), row_partner<LEVEL>(__double2loint(v)));
}
template <int LEVEL>
__device__ __forceinline__ double row_sum_step(double v) { return v + row_partner<LEVEL>(v); }

// The MFMA loop of one wave over the n tiles of its LDS list against ITS sub-block (B operand b):
// per lane -- slot lane & 15, target rows 4 (lane >> 4) .. + 3 of every tile -- the two best tiles
// (value, tile) and the third-best value.  One MFMA per tile; the A fragments of the next SW_G tiles
// are requested before this group's MFMAs are issued (as GLOBAL loads: while they were flat, every LDS
// read of a list entry waited for them).  SW_PAD pad tiles (rows that never win) follow the
// list's last entry.  (Three groups in flight with the winners kept as list positions measured slower:
// 0.738 against 0.714 ms per registration, round 4.)
constexpr int SW_PAD = 2 * SW_G;
__device__ __forceinline__ void sweep_sub_block(const unsigned *__restrict__ list, int n, const PEDP_GLOBAL float *__restrict__ tgtf,
                                                int frag, float b, float &b1, int &t1, float &b2, int &t2, float &b3) {
    if (n <= 0) return;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    float a[SW_G];
    unsigned un[SW_G];
#pragma unroll
    for (int g = 0; g < SW_G; ++g) {
        un[g] = list[g];
        a[g] = tgtf[(size_t)un[g] * 64 + frag];
    }
    for (int k = 0; k < n; k += SW_G) {
        float an[SW_G];
        unsigned unn[SW_G];
#pragma unroll
        for (int g = 0; g < SW_G; ++g) {
            unn[g] = list[k + SW_G + g];  // pad tiles follow the last real one
            an[g] = tgtf[(size_t)unn[g] * 64 + frag];
        }
        f32x4 acc[SW_G];
#pragma unroll
        for (int g = 0; g < SW_G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g], b, zero, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < SW_G; ++g) {
            const float v = fminf(fminf(fminf(acc[g][0], acc[g][1]), acc[g][2]), acc[g][3]);
            const int tile = (int)un[g];
            const bool lt1 = v < b1, lt2 = v < b2;
            t2 = lt1 ? t1 : (lt2 ? tile : t2);
            t1 = lt1 ? tile : t1;
            b3 = __builtin_amdgcn_fmed3f(b2, b3, v);  // b2 <= b3: the third smallest of the four
            b2 = __builtin_amdgcn_fmed3f(b1, b2, v);  // b1 <= b2
            b1 = fminf(b1, v);
        }
#pragma unroll
        for (int g = 0; g < SW_G; ++g) { a[g] = an[g]; un[g] = unn[g]; }
    }
}

// BATCH: the launch carries several poses (grid.y); a separate instantiation, so that a kernel
// trace tells the single registration's launches from a batch's
template <int W, bool BATCH>
__global__ __launch_bounds__(W * 64, 4) void icp_pass_kernel(IcpState *st0, const PassArgs a0) {
    static_assert(W == 8 && CH == 128, "a chunk is two halves of four 16-slot sub-blocks");
    const size_t pose_off = BATCH ? (size_t)blockIdx.y * a0.pose_stride : 0;
    IcpState *st = pose_ptr(st0, pose_off);
    // The argument block (with this pose's pointers) is parked in LDS and read from there where it is
    // used: held in scalar registers for the whole kernel its 40-odd fields overflow the SGPR file,
    // and the spills -- executed at entry by EVERY launched workgroup -- left tens of MB of dirty
    // scratch for the kernel boundary to write back.
    __shared__ PassArgs sa;
    __shared__ FinishLds<W * 64, 2048> fin;
    // a wave's slots (its sub-block): written and read by that wave only
    __shared__ double wp[W][3][16], accsh[W][PSTRIDE];
    __shared__ float wcs[W][3][16], weps[W][16], wS[W][16], wrho[W][16];
    __shared__ int wpi[W][16], wkk[W][16], misc[8];
    __shared__ unsigned wtl[W][WTL + SW_PAD];
    __shared__ float4 wnode[W][16], wsph0[64];
    __shared__ double rbs[16];  // rebuild passes: the pose so far (3 x 4), its norm bound, the reach
    __shared__ float wnode_r[W][16];
    if (threadIdx.x == 0) {
        PassArgs t = a0;
        t.Pk = pose_ptr(a0.Pk, pose_off); t.Tprev = pose_ptr(a0.Tprev, pose_off); t.live = pose_ptr(a0.live, pose_off);
        t.live_list = pose_ptr(a0.live_list, pose_off); t.hist = pose_ptr(a0.hist, pose_off);
        t.idx_out = pose_ptr(a0.idx_out, pose_off); t.partials = pose_ptr(a0.partials, pose_off);
        t.packet = pose_ptr(a0.packet, pose_off); t.ticket = pose_ptr(a0.ticket, pose_off);
        sa = t;
    }
    const PassArgs &a = sa;
#if PEDP_ICP_STAMPS
    const long long rt_entry = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    // the first unit's live-list entry is requested together with the state (the list has one entry
    // per chunk, so the index is always inside it; the value is used only when it is valid)
    int chunk_next = pose_ptr(a0.live_list, pose_off)[blockIdx.x < (unsigned)a0.n_chunks ? blockIdx.x : 0];
    // both parities of this workgroup's entry of the visit plan, requested before the pass number is known
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i vis0 = {0, 0, 0, 0}, vis1 = {0, 0, 0, 0};
    if (!BATCH && a0.visit && blockIdx.x < (unsigned)a0.visit_cap) {
        vis0 = ((const v4i *)a0.visit)[blockIdx.x];
        vis1 = ((const v4i *)a0.visit)[(size_t)a0.visit_cap + blockIdx.x];
    }
    // word spheres do not depend on the chunk: the first 64 are requested before anything else and parked in LDS
    if (threadIdx.x < 64) wsph0[threadIdx.x] = a0.word_sph[(int)threadIdx.x < a0.n_words ? threadIdx.x : 0];
    if (st->done) return;
    const bool rebuild = st->rebuild != 0;
    const int n_live = st->n_live, pass = st->pass;
    const unsigned ticket_base = st->ticket_base;
    const v4i vis = (pass & 1) ? vis1 : vis0;
    const int tag_now = st->nonce + pass + 1;  // (a registration runs well under 65,535 passes; beyond that no plan is made)
    const bool planned = !BATCH && !rebuild && vis[2] == tag_now && vis[3] == n_live;  // (all workgroups agree: the plan is written whole)
    // Workgroups with chunks take a ticket when they are through; the one that draws the last closes the
    // pass.  The others leave at once -- but sign off first (a counter of sixteen, each on a line of its own:
    // hundreds of atomics on one word in the first microsecond held up everybody's loads), and the closing
    // workgroup rewrites the state only after all of them have: a batch's grid is not resident at once, a
    // workgroup that starts late must not find the next pass's state.
    int n_wg = rebuild ? a0.n_chunks : n_live;
    n_wg = n_wg < (int)gridDim.x ? n_wg : (int)gridDim.x;
    n_wg = n_wg < 1 ? 1 : n_wg;  // (no live chunk at all: workgroup 0 still closes the pass)
    if ((int)blockIdx.x >= n_wg) {
        if (a0.fuse && threadIdx.x == 0)
            __hip_atomic_fetch_add((g_u32 *)(uintptr_t)(pose_ptr(a0.ticket, pose_off) + 32 * (1 + (blockIdx.x & 15))), 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
#if PEDP_ICP_STAMPS
    if (threadIdx.x == 0 && pass < 32 && blockIdx.x < 512 && blockIdx.y == 0) g_icp_rt[pass][blockIdx.x][0] = rt_entry;
#endif
    PEDP_RT(pass, 1);
    const float inf = __uint_as_float(0x7F800000u);
    const double dinf = __longlong_as_double(0x7FF0000000000000ll);
    const double dnan = __longlong_as_double(0x7FF8000000000000ll);
    const double ccx = st->centroid[0], ccy = st->centroid[1], ccz = st->centroid[2];
    const float r_search = st->r_search;
    // A rebuild pass asks the chunks' bounding spheres first (64 of this workgroup's chunks per round, one
    // per lane, every wave for itself): a sphere moved by the pose so far that stays farther than r + margin
    // from the target's box holds no live point -- the chunk is not touched (only, if it was live before,
    // its points' correspondences are withdrawn).  The others are decided point by point as before.
    unsigned long long todo = 0ull;  // wave-uniform: chunks of the current round still to visit
    int todo_base = -64, it = 0;
    if (rebuild && threadIdx.x == 0) {
        double Rs[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) { Rs[k] = st->T[k]; rbs[k] = Rs[k]; }
        // |R x| <= rscale |x|: the square root of the largest row sum of |R^T R| bounds the spectral norm
        double m = 0.0;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            double row = 0.0;
#pragma unroll
            for (int v = 0; v < 3; ++v) row += fabs(Rs[u] * Rs[v] + Rs[4 + u] * Rs[4 + v] + Rs[8 + u] * Rs[8 + v]);
            m = row > m ? row : m;
        }
        rbs[12] = sqrt(m) * (1.0 + 1e-9);
        rbs[13] = sqrt(st->r2live) * (1.0 + 1e-9);
    }
    __syncthreads();  // the argument block, the word spheres (and the rebuild constants) are in LDS
    for (;;) {
        // The thread index is made opaque per chunk: everything derived from it (LDS addresses, lane
        // masks, role predicates) is then computed where it is used instead of being hoisted out of
        // this loop and kept alive -- spilled -- through every phase.
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int half = wv >> 2, q4 = wv & 3;   // the wave's half of the chunk (64 points), its quarter of the half's slots
        const int j = lane & 15, g = lane >> 4;  // MFMA layout: slot of the sub-block, row group / component
        const int frag = j * 4 + g;              // float offset inside a 16-point target tile
        const unsigned long long lt = (1ull << lane) - 1ull;
        // a rebuild pass visits chunks (unit = chunk id), other passes the live list (unit = rank);
        // `unit` also indexes the chunk's partial sums (see icp_finish_body)
        const PEDP_GLOBAL double *Pk_in = as_global(a.Pk) + (size_t)(pass & 1) * a.pp_stride, *Tp_in = as_global(a.Tprev) + (size_t)(pass & 1) * a.pp_stride;
        PEDP_GLOBAL double *Pk_out = as_global(a.Pk) + (size_t)((pass + 1) & 1) * a.pp_stride, *Tp_out = as_global(a.Tprev) + (size_t)((pass + 1) & 1) * a.pp_stride;
        int chunk, unit;
        if (rebuild) {
            bool more = true;
            while (todo == 0ull) {  // wave-uniform
                todo_base += 64;
                if ((long long)blockIdx.x + (long long)todo_base * (long long)gridDim.x >= (long long)a.n_chunks) { more = false; break; }
                const long long u = (long long)blockIdx.x + (long long)(todo_base + lane) * (long long)gridDim.x;
                bool visit = false, withdraw = false;
                if (u < (long long)a.n_chunks) {
                    double Rs[12];
#pragma unroll
                    for (int k = 0; k < 12; ++k) Rs[k] = rbs[k];
                    const double rscale = rbs[12], reach = rbs[13];
                    const bool was_live = (as_global(a.live)[a.n_lw + (u >> 6)] >> (u & 63)) & 1ull;
#pragma unroll 2
                    for (int sb = 0; sb < 8; ++sb) {
                        const PEDP_GLOBAL double *sp8 = as_global(a.chunk_sph) + (size_t)(8 * u + sb) * 4;
                        const double cx = sp8[0], cy = sp8[1], cz = sp8[2], cr = sp8[3];
                        const double tx = Rs[0] * cx + Rs[1] * cy + Rs[2] * cz + Rs[3], ty = Rs[4] * cx + Rs[5] * cy + Rs[6] * cz + Rs[7],
                                     tz = Rs[8] * cx + Rs[9] * cy + Rs[10] * cz + Rs[11];
                        const double ex = fmax(fmax(a.lo[0] - tx, tx - a.hi[0]), 0.0), ey = fmax(fmax(a.lo[1] - ty, ty - a.hi[1]), 0.0),
                                     ez = fmax(fmax(a.lo[2] - tz, tz - a.hi[2]), 0.0);
                        const double lim = cr * rscale + reach + 1e-9 * (fabs(tx) + fabs(ty) + fabs(tz) + 1.0);
                        visit |= !(cr < 0.0) && !(ex * ex + ey * ey + ez * ez > lim * lim);  // (also when anything is NaN)
                    }
                    withdraw = !visit && (was_live || pass == 0);   // pass 0: every chunk's correspondences start at "none"
                }
                todo = __builtin_amdgcn_ballot_w64(visit);
                unsigned long long wd = __builtin_amdgcn_ballot_w64(withdraw);
                while (wd != 0ull) {  // rare: a chunk that was live and no longer is
                    const int i = __builtin_ctzll(wd);
                    wd &= wd - 1ull;
                    const int64_t k = ((int64_t)blockIdx.x + (int64_t)(todo_base + i) * gridDim.x) * CH + tid;
                    if (tid < CH && k < a.N) as_global(a.idx_out)[as_global(a.perm)[k]] = -1;
                }
            }
            if (!more) break;
            const int i = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            unit = (int)(blockIdx.x + (unsigned)(todo_base + i) * gridDim.x);
            chunk = unit;
        } else {
            unit = (int)(blockIdx.x + (unsigned)it * gridDim.x);
            ++it;
            if (unit >= n_live) break;
            if (planned && it == 1) { unit = vis[0]; chunk = vis[1]; }
            else chunk = unit == (int)blockIdx.x ? chunk_next : as_global(a.live_list)[unit];
        }
        long long t_chunk0 = 0;
        if (!BATCH && !rebuild) t_chunk0 = (long long)__builtin_amdgcn_s_memtime();
        if (tid == 0) PEDP_STAMP(1, blockIdx.x, 0);
        PEDP_WV(0, __builtin_amdgcn_s_memtime());
        PEDP_WV(9, pass);
        PEDP_WV(10, __builtin_amdgcn_s_memrealtime());
        PEDP_WV(12, ((long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492));  // XCC_ID, HW_ID
        // ---- 1. every wave transforms the 64 points of its half (the four waves of a half do the same
        // arithmetic and get the same ballots; quarter 0 stores), box test, compaction of the candidates:
        // the half's candidates in ascending position take the half's slots 0.., the wave keeps those
        // whose rank falls into its quarter
        int nsl;  // real slots of this wave's sub-block
        {
            bool cand = false, near = false;
            int pi = -1;
            double x = 0.0, y = 0.0, z = 0.0, dprev = dnan;
            const int64_t k = (int64_t)chunk * CH + half * 64 + lane;
            const bool valid = k < a.N;
            if (valid) {
                pi = as_global(a.perm)[k];
                if (rebuild) {
                    x = as_global(a.src)[3 * (int64_t)pi]; y = as_global(a.src)[3 * (int64_t)pi + 1]; z = as_global(a.src)[3 * (int64_t)pi + 2];
                    xform(st->T_init, x, y, z);
                    for (int q = 1; q <= pass; ++q) xform_g(as_global(a.hist) + 16 * q, x, y, z);
                    // a chunk that was live in the pass before has that pass's neighbours (every point of a live
                    // chunk gets one, or NaN): the search radii need not start from r again
                    if ((as_global(a.live)[a.n_lw + (chunk >> 6)] >> (chunk & 63)) & 1ull)
                        dprev = sqrt(dist2(x, y, z, Tp_in[3 * k], Tp_in[3 * k + 1], Tp_in[3 * k + 2]));
                } else {
                    x = Pk_in[3 * k]; y = Pk_in[3 * k + 1]; z = Pk_in[3 * k + 2];
                    const double ux = Tp_in[3 * k], uy = Tp_in[3 * k + 1], uz = Tp_in[3 * k + 2];
                    xform(st->upd, x, y, z);
                    // Temporal coherence: last pass's neighbour is still a target point, so the new nearest
                    // neighbour is no farther than it is now.  NaN (no neighbour last pass) fails the
                    // comparison below and leaves the full radius.
                    dprev = sqrt(dist2(x, y, z, ux, uy, uz));
                }
                const double ex = fmax(fmax(a.lo[0] - x, x - a.hi[0]), 0.0), ey = fmax(fmax(a.lo[1] - y, y - a.hi[1]), 0.0),
                             ez = fmax(fmax(a.lo[2] - z, z - a.hi[2]), 0.0);
                const double d2box = ex * ex + ey * ey + ez * ez;
                cand = d2box <= st->r2cut;   // r2cut = r^2 (1 + 1e-12): rounding-safe
                near = d2box <= st->r2live;
            }
            const unsigned long long mc = __builtin_amdgcn_ballot_w64(cand), mn = __builtin_amdgcn_ballot_w64(near);
            const int wc = __builtin_popcountll(mc);
            if (q4 == 0) {  // (the other copy: a wave of this half that comes late still reads this pass's inputs)
                if (lane == 0) misc[half] = mn != 0ull;
                if (valid) {  // (a rebuild pass stores every visited chunk's coordinates; only the live ones are read again)
                    if (!cand) { as_global(a.idx_out)[pi] = -1; Tp_out[3 * k] = dnan; }
                    Pk_out[3 * k] = x; Pk_out[3 * k + 1] = y; Pk_out[3 * k + 2] = z;
                }
            }
            const int sl = __builtin_popcountll(mc & lt) - 16 * q4;
            if (cand && sl >= 0 && sl < 16) {
                const float sx = (float)(x - ccx), sy = (float)(y - ccy), sz = (float)(z - ccz);
                // error bound of the fp32 surrogate against the float64 distance
                const float s1 = fabsf(sx) + fabsf(sy) + fabsf(sz);
                const float Mi = 2.0f * s1 * a.Tn + a.T2;
                weps[wv][sl] = 1.1920929e-7f * (5.0f * Mi + 2.0f * fminf(st->r1, s1 + a.Tn) * (a.Tn + s1)) * 1.0001f;
                wS[wv][sl] = sx * sx + sy * sy + sz * sz;
                wpi[wv][sl] = pi;
                wkk[wv][sl] = half * 64 + lane;
                wp[wv][0][sl] = x; wp[wv][1][sl] = y; wp[wv][2][sl] = z;
                wcs[wv][0][sl] = sx; wcs[wv][1][sl] = sy; wcs[wv][2][sl] = sz;
                // search radius of the slot: the distance to last pass's neighbour, rounded up, at most r
                const float rp = (float)dprev * 1.00001f + 1e-5f * s1 + 1e-6f;
                wrho[wv][sl] = rp < r_search ? rp : r_search;
            }
            nsl = wc - 16 * q4;
            nsl = nsl < 0 ? 0 : (nsl > 16 ? 16 : nsl);
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed
        }
        if (rebuild) {  // is the chunk live?  (both halves' flags)
            __syncthreads();
            const bool is_live = (misc[0] | misc[1]) != 0;
            if (!is_live) {  // workgroup-uniform: the chunk stays outside the live set
                __syncthreads();  // (the flags are rewritten by the next chunk)
                continue;
            }
            if (tid == 0) __hip_atomic_fetch_or((g_u64 *)(uintptr_t)&a.live[chunk >> 6], 1ull << (chunk & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) PEDP_STAMP(1, blockIdx.x, 1);
        PEDP_WV(1, __builtin_amdgcn_s_memtime());
        const bool real = j < nsl;
        double fd = dinf;      // the slot's result: squared distance, target index (-1: none), neighbour and normal
        int fj = -1;
        double wt[3] = {0.0, 0.0, 0.0}, wn[3] = {0.0, 0.0, 0.0};
        bool have_tn = false;
        int ntl_w = 0, nfb_w = 0;
        if (nsl > 0) {  // wave-uniform
            // ---- 2. what the target's spheres are tested against: a COVER of the sub-block's slots by bounding
            // spheres (centred fp32 coordinates), each with the largest search radius of its slots.  The slots'
            // binary tree -- pairs, quads, octets, all sixteen: the levels of a butterfly reduction -- is cut
            // where a node's sphere is no wider than r.  A compact sub-block is one node; 16 consecutive points
            // of the spatial order that straddle a jump of the curve (a WIDE sub-block) come out as the few
            // compact groups they really are, down to single points, instead of one huge ball whose tiles
            // would all have to be swept.  Nodes live in the wave's LDS; the first also in registers.
            const float px = real ? wcs[wv][0][j] : 0.f, py = real ? wcs[wv][1][j] : 0.f, pz = real ? wcs[wv][2][j] : 0.f;
            const float rho_j = real ? wrho[wv][j] : 0.f;
            int nn;  // nodes of the cover
            {
                const float big = 3e38f, wr = st->wide_radius;
                float lx = real ? px : big, hx = real ? px : -big, ly = real ? py : big, hy = real ? py : -big,
                      lz = real ? pz : big, hz = real ? pz : -big, rmx = rho_j;
                // level 0: the point itself, widened by the rounding of its centred coordinates
                float4 node = make_float4(px, py, pz, 1e-5f * (fabsf(px) + fabsf(py) + fabsf(pz)) + 1e-6f);
                float node_r = rho_j;
                bool open = real;   // no level of this lane's chain is in the cover yet
                bool mine = false;  // this lane holds a node of the cover
                int cut = 0;        // a chain inside this lane's current node has been closed
                auto level = [&](auto LV) {
                    constexpr int lv = decltype(LV)::value;  // butterfly level 0..3: nodes of 2 << lv slots
                    constexpr int off = 1 << lv;
                    lx = fminf(lx, row_partner<lv>(lx)); hx = fmaxf(hx, row_partner<lv>(hx));
                    ly = fminf(ly, row_partner<lv>(ly)); hy = fmaxf(hy, row_partner<lv>(hy));
                    lz = fminf(lz, row_partner<lv>(lz)); hz = fmaxf(hz, row_partner<lv>(hz));
                    rmx = fmaxf(rmx, row_partner<lv>(rmx));
                    const float mx = 0.5f * (lx + hx), my = 0.5f * (ly + hy), mz = 0.5f * (lz + hz);
                    const float ex = hx - mx, ey = hy - my, ez = hz - mz;
                    const float rad = sqrtf(ex * ex + ey * ey + ez * ez) * 1.0001f + 1e-6f * (fabsf(mx) + fabsf(my) + fabsf(mz)) + 1e-30f;
                    // a node is cut where its sphere is wider than r -- or where a part of it has been cut already
                    // (so that rounding can never leave a slot outside the cover); lanes of one node agree
                    cut |= row_partner<lv>(cut);
                    const bool wide_here = rad > wr || cut != 0;
                    // the level below is in the cover where this level is cut: its nodes close their chains
                    if (open && wide_here) { mine = (j & (off - 1)) == 0; open = false; cut = 1; }
                    if (open) { node = make_float4(mx, my, mz, rad); node_r = rmx; }
                };
                level(std::integral_constant<int, 0>{});
                level(std::integral_constant<int, 1>{});
                level(std::integral_constant<int, 2>{});
                level(std::integral_constant<int, 3>{});
                if (open) mine = j == 0;  // the whole sub-block is one node
                const unsigned long long nm = __builtin_amdgcn_ballot_w64(mine && g == 0);
                nn = __builtin_popcountll(nm);
                if (mine && g == 0) {
                    const int at = __builtin_popcountll(nm & lt);
                    wnode[wv][at] = node;
                    wnode_r[wv][at] = node_r;
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);
            }
            PEDP_WV(7, __builtin_amdgcn_s_memtime());
            const float4 node0 = wnode[wv][0];
            const float node0_r = wnode_r[wv][0];
            // Can a target sphere ts (a tile's, or a whole mask word's) hold the nearest neighbour of a slot
            // of this sub-block?  Every slot has a search radius rho <= r, a node the largest of its slots'.
            auto near_sb = [&](const float4 &ts) -> bool {
                const float dx = ts.x - node0.x, dy = ts.y - node0.y, dz = ts.z - node0.z;
                const float lim = node0_r + node0.w + ts.w;
                bool any = !((dx * dx + dy * dy + dz * dz) > lim * lim * 1.00001f + 1e-6f);
#pragma nounroll
                for (int i = 1; i < nn; ++i) {  // wave-uniform; broadcast reads
                    const float4 nd = wnode[wv][i];
                    const float ex = ts.x - nd.x, ey = ts.y - nd.y, ez = ts.z - nd.z;
                    const float li = wnode_r[wv][i] + nd.w + ts.w;
                    any |= !((ex * ex + ey * ey + ez * ez) > li * li * 1.00001f + 1e-6f);
                }
                return any && ts.w >= 0.f;
            };
            // MFMA B operand of the sub-block: (-2x', -2y', -2z', 1) per slot, dummies (0, 0, 0, 1)
            const float bfrag = g == 3 ? 1.0f : (real ? -2.0f * wcs[wv][g < 3 ? g : 0][j] : 0.f);
            float b1 = inf, b2 = inf, b3 = inf;
            int t1 = a.n_tiles, t2 = a.n_tiles;
#if PEDP_ICP_STAMPS
            int dbg_words = 0, dbg_batches = 0;
#endif
            // ---- 3. culling and sweep.  Level 1: lane l tests the sphere of mask word l (64 tiles = 1,024
            // sorted rows).  Level 2, eight surviving words per round of loads: lane l tests tile 64 word + l,
            // the ballot is the word's tile mask; the survivors of ALL words go to the wave's LDS list, which is
            // then swept in one go (the list is swept early only if the next round might not fit).
            int n = 0;
            for (int R = 0; R * 64 < a.n_words; ++R) {
                const int wi = R * 64 + lane;
                const float4 wsR = R == 0 ? wsph0[lane] : gload4(a.word_sph + (wi < a.n_words ? wi : 0));
                unsigned long long km = __builtin_amdgcn_ballot_w64(wi < a.n_words && near_sb(wsR));
#if PEDP_ICP_STAMPS
                dbg_words += __builtin_popcountll(km);
#endif
                while (km != 0ull) {  // wave-uniform
#if PEDP_ICP_STAMPS
                    ++dbg_batches;
#endif
                    if (n + L2_WORDS * 64 > WTL) {  // rare: a dense neighbourhood
                        if (lane < SW_PAD) wtl[wv][n + lane] = (unsigned)a.n_tiles;  // pad tiles: rows that never win
                        __builtin_amdgcn_s_waitcnt(0xC07F);
                        sweep_sub_block(wtl[wv], n, as_global(a.tgtf), frag, bfrag, b1, t1, b2, t2, b3);
                        ntl_w += n;
                        n = 0;
                    }
                    int word[L2_WORDS];
                    float4 ts[L2_WORDS];
#pragma unroll
                    for (int u = 0; u < L2_WORDS; ++u) {
                        word[u] = -1;
                        if (km != 0ull) {
                            word[u] = R * 64 + __builtin_ctzll(km);
                            km &= km - 1ull;
                        }
                        const int tile = word[u] * 64 + lane;
                        ts[u] = gload4(a.tile_sph + ((word[u] >= 0 && tile < a.n_tiles) ? tile : 0));
                    }
                    // the batch's spheres against the cover, node by node: a node is read from LDS once per
                    // batch (the next one requested before this one's tests), not once per sphere -- a wide
                    // sub-block's four or five nodes used to cost a dependent LDS round trip per (sphere, node)
                    unsigned nearbits = 0u;
                    {
                        float4 nd = node0;
                        float nd_r = node0_r;
                        for (int i = 0; i < nn; ++i) {  // wave-uniform
                            float4 ndn = nd;
                            float ndn_r = nd_r;
                            if (i + 1 < nn) { ndn = wnode[wv][i + 1]; ndn_r = wnode_r[wv][i + 1]; }
#pragma unroll
                            for (int u = 0; u < L2_WORDS; ++u) {
                                const float ex = ts[u].x - nd.x, ey = ts[u].y - nd.y, ez = ts[u].z - nd.z;
                                const float li = nd_r + nd.w + ts[u].w;
                                if (!((ex * ex + ey * ey + ez * ez) > li * li * 1.00001f + 1e-6f)) nearbits |= 1u << u;
                            }
                            nd = ndn;
                            nd_r = ndn_r;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < L2_WORDS; ++u) {
                        if (word[u] < 0) continue;
                        const int tile = word[u] * 64 + lane;
                        const bool keep = tile < a.n_tiles && (nearbits >> u & 1u) != 0u && ts[u].w >= 0.f;
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
                        if (keep) wtl[wv][n + __builtin_popcountll(m & lt)] = (unsigned)tile;
                        n += __builtin_popcountll(m);
                    }
                }
            }
            PEDP_WV(8, __builtin_amdgcn_s_memtime());
            if (lane < SW_PAD) wtl[wv][n + lane] = (unsigned)a.n_tiles;  // pad tiles: rows that never win
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
            sweep_sub_block(wtl[wv], n, as_global(a.tgtf), frag, bfrag, b1, t1, b2, t2, b3);
            ntl_w += n;
            if (tid == 0) PEDP_STAMP(1, blockIdx.x, 2);
            PEDP_WV(2, __builtin_amdgcn_s_memtime());
            PEDP_WV(5, ((long long)dbg_words << 32) | ((long long)dbg_batches << 16) | ((long long)nn << 8) | nsl);
            PEDP_WV(6, ntl_w);
            // ---- 4. exact selection.  fp32 g is a filter: with the slot's error bound e the true nearest
            // neighbour lies in a tile whose value is within 2 e of the slot's minimum.  A lane re-scores the
            // four rows it saw of its best tile (and of its second best, if that is inside the window too) in
            // float64 with the oracle's formula, lexicographic (d^2, index); a third tile of one lane inside
            // the window sends the slot to the exact search.
            auto load_rows = [&](int tile, double (&rw)[4][6], int (&ri)[4]) {
                const int64_t row0 = (int64_t)tile * 16 + 4 * g;  // this lane's rows of the tile
                // four rows of 48 B lie one behind the other, 16-B aligned: twelve 16-B loads (the sorted rows are
                // allocated and zero-filled up to the pad tiles, so rows beyond Nt are readable; they are not scored)
                typedef double v2d __attribute__((ext_vector_type(2)));
                const PEDP_GLOBAL v2d *rows = (const PEDP_GLOBAL v2d *)(uintptr_t)(a.tgt_s + 6 * row0);
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    const v2d t = rows[q];
                    rw[q / 3][2 * (q % 3)] = t[0];
                    rw[q / 3][2 * (q % 3) + 1] = t[1];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) ri[r] = as_global(a.tperm)[row0 + r < a.Nt ? row0 + r : 0];
            };
            const float e = real ? weps[wv][j] : 0.f, Si = real ? wS[wv][j] : 3e38f;
            const double qx = wp[wv][0][j], qy = wp[wv][1][j], qz = wp[wv][2][j];
            float mg = fminf(b1, __shfl_xor(b1, 16, 64));
            mg = fminf(mg, __shfl_xor(mg, 32, 64));
            const bool maybe = real && mg + Si <= st->r2f + 4.0f * e + 4.8e-7f * Si;  // else certainly farther than r
            const float win = mg + 2.0f * e;
            double bd = dinf;
            int bj = 0x7FFFFFFF;
            auto eval_rows = [&](int tile, const double (&rw)[4][6], const int (&ri)[4]) {
                const int64_t row0 = (int64_t)tile * 16 + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (row0 + r < a.Nt) {
                        const double d = dist2(qx, qy, qz, rw[r][0], rw[r][1], rw[r][2]);
                        if (d < bd || (d == bd && ri[r] < bj)) {
                            bd = d; bj = ri[r];
                            wt[0] = rw[r][0]; wt[1] = rw[r][1]; wt[2] = rw[r][2];
                            wn[0] = rw[r][3]; wn[1] = rw[r][4]; wn[2] = rw[r][5];
                        }
                    }
                }
            };
            if (maybe && b1 <= win) {
                double rw1[4][6];
                int ri1[4];
                load_rows(t1, rw1, ri1);
                eval_rows(t1, rw1, ri1);
            }
            if (__builtin_amdgcn_ballot_w64(maybe && b2 <= win) != 0ull) {  // (about one lane in a hundred)
                if (maybe && b2 <= win) {
                    double rw2[4][6];
                    int ri2[4];
                    load_rows(t2, rw2, ri2);
                    eval_rows(t2, rw2, ri2);
                }
            }
            // the slot's winner over its four lanes; the lane that holds it hands neighbour and normal over
            fd = bd;
            int fjj = bj;
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const double od = __shfl_xor(fd, off, 64);
                const int oj = __shfl_xor(fjj, off, 64);
                lexmin(fd, fjj, od, oj);
            }
            const bool found = maybe && fjj != 0x7FFFFFFF;
            {
                int gw = (found && bj == fjj) ? g : 0;  // target indices are unique: one lane of the four at most
                gw |= __shfl_xor(gw, 16, 64);
                gw |= __shfl_xor(gw, 32, 64);
                const int srcl = j + 16 * gw;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    wt[c] = __shfl(wt[c], srcl, 64);
                    wn[c] = __shfl(wn[c], srcl, 64);
                }
            }
            fj = found ? fjj : -1;
            fd = found ? fd : dinf;
            have_tn = found;
            // ---- 5. ambiguous slots (three tiles of one lane inside the window): exact float64 search over
            // the tiles within the slot's own search radius, the whole wave per slot
            unsigned amb = (unsigned)__builtin_amdgcn_ballot_w64(maybe && b3 <= win);
            {
                const unsigned long long am = __builtin_amdgcn_ballot_w64(maybe && b3 <= win);
                amb = (unsigned)((am | (am >> 16) | (am >> 32) | (am >> 48)) & 0xFFFFull);
            }
            if (tid == 0) PEDP_STAMP(1, blockIdx.x, 3);
            while (amb != 0u) {  // wave-uniform, rare
                const int s = __builtin_ctz(amb);
                amb &= amb - 1u;
                ++nfb_w;
                const double sx64 = wp[wv][0][s], sy64 = wp[wv][1][s], sz64 = wp[wv][2][s];
                const float sx = wcs[wv][0][s], sy = wcs[wv][1][s], sz = wcs[wv][2][s], rho_s = wrho[wv][s];
                const float slack = 1e-5f * (fabsf(sx) + fabsf(sy) + fabsf(sz)) + 1e-6f;  // fp32 rounding of the centred point
                auto near_pt = [&](const float4 &ts) -> bool {
                    const float dx = ts.x - sx, dy = ts.y - sy, dz = ts.z - sz;
                    const float lim = rho_s + ts.w + slack;
                    return !((dx * dx + dy * dy + dz * dz) > lim * lim * 1.00001f + 1e-6f) && ts.w >= 0.f;
                };
                double xd = dinf;
                int xj = 0x7FFFFFFF;
                for (int R = 0; R * 64 < a.n_words; ++R) {
                    const int wi = R * 64 + lane;
                    const float4 wsR = R == 0 ? wsph0[lane] : gload4(a.word_sph + (wi < a.n_words ? wi : 0));
                    unsigned long long km = __builtin_amdgcn_ballot_w64(wi < a.n_words && near_pt(wsR));
                    while (km != 0ull) {
                        const int word = R * 64 + __builtin_ctzll(km);
                        km &= km - 1ull;
                        const int tile = word * 64 + lane;
                        const float4 ts = gload4(a.tile_sph + (tile < a.n_tiles ? tile : 0));
                        const bool keep = tile < a.n_tiles && near_pt(ts);
                        scan_near_tiles(__builtin_amdgcn_ballot_w64(keep), tile, a, sx64, sy64, sz64, lane, xd, xj);
                    }
                }
#pragma unroll
                for (int off = 1; off <= 32; off <<= 1) {
                    const double od = __shfl_xor(xd, off, 64);
                    const int oj = __shfl_xor(xj, off, 64);
                    lexmin(xd, xj, od, oj);
                }
                if (j == s) {
                    fd = xj == 0x7FFFFFFF ? dinf : xd;
                    fj = xj == 0x7FFFFFFF ? -1 : xj;
                    have_tn = false;  // neighbour and normal are fetched by index below
                }
            }
        }
        if (tid == 0) PEDP_STAMP(1, blockIdx.x, 4);
        PEDP_WV(3, __builtin_amdgcn_s_memtime());
        // ---- 6. the sub-block's partial sums (layout: see icp_accumulate_kernel).  Four lanes per slot,
        // lane group g owns the packet entries k = g (mod 4); entries are summed over the wave's 16
        // slots by a shuffle tree, then over the waves in order: a fixed tree.
        {
            double acc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = 0.0;
            const int i = real ? wpi[wv][j] : -1;
            int jn = fj;
            const double dd = fd;
            if (i >= 0) {
                if (jn >= 0 && !(dd < st->r2)) jn = -1;  // strict, as SearchHybrid's lower_bound
                const int64_t kp = (int64_t)chunk * CH + wkk[wv][j];
                if (g == 0) {
                    as_global(a.idx_out)[i] = jn;
                    if (jn < 0) Tp_out[3 * kp] = dnan;
                }
                if (jn >= 0) {
                    const double sx = wp[wv][0][j], sy = wp[wv][1][j], sz = wp[wv][2][j];
                    double tx = wt[0], ty = wt[1], tz = wt[2], nx = wn[0], ny = wn[1], nz = wn[2];
                    if (!have_tn) {  // rare: the exact search returns an index
                        tx = as_global(a.tgt)[3 * (int64_t)jn]; ty = as_global(a.tgt)[3 * (int64_t)jn + 1]; tz = as_global(a.tgt)[3 * (int64_t)jn + 2];
                        if (a.estimator == PEDP_POINT_TO_PLANE) {
                            nx = as_global(a.nrm)[3 * (int64_t)jn]; ny = as_global(a.nrm)[3 * (int64_t)jn + 1]; nz = as_global(a.nrm)[3 * (int64_t)jn + 2];
                        }
                    }
                    if (g == 0) { Tp_out[3 * kp] = tx; Tp_out[3 * kp + 1] = ty; Tp_out[3 * kp + 2] = tz; }
                    // entry k of the packet goes to lane group g = k % 4, accumulator k / 4
#define PEDP_PUT(K, V)                                   \
    do {                                                 \
        const double v_ = (V);                           \
        if (g == ((K) & 3)) acc[(K) >> 2] = v_;          \
    } while (0)
                    if (a.estimator == PEDP_POINT_TO_PLANE) {
                        const double r = (sx - tx) * nx + (sy - ty) * ny + (sz - tz) * nz;
                        const double J[6] = {sy * nz - sz * ny, sz * nx - sx * nz, sx * ny - sy * nx, nx, ny, nz};
                        int k = 0;
#pragma unroll
                        for (int u = 0; u < 6; ++u)
#pragma unroll
                            for (int v = u; v < 6; ++v) { PEDP_PUT(k, J[u] * J[v]); ++k; }
#pragma unroll
                        for (int u = 0; u < 6; ++u) PEDP_PUT(21 + u, J[u] * r);
                    } else {
                        const double s3[3] = {sx - ccx, sy - ccy, sz - ccz}, t3[3] = {tx - ccx, ty - ccy, tz - ccz};
#pragma unroll
                        for (int u = 0; u < 3; ++u) { PEDP_PUT(u, s3[u]); PEDP_PUT(3 + u, t3[u]); }
#pragma unroll
                        for (int u = 0; u < 3; ++u)
#pragma unroll
                            for (int v = 0; v < 3; ++v) PEDP_PUT(6 + 3 * u + v, t3[u] * s3[v]);
                    }
                    PEDP_PUT(27, dd);
                    PEDP_PUT(28, 1.0);
#undef PEDP_PUT
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                double v = acc[k];
#pragma unroll
                for (int once = 0; once < 1; ++once) v = row_sum_step<3>(row_sum_step<2>(row_sum_step<1>(row_sum_step<0>(v))));
                if (j == 0) accsh[wv][4 * k + g] = v;
            }
            // (entries 29, 30 of the tree are zero: the statistics replace them)
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (lane == 0) { accsh[wv][PACKET] = (double)ntl_w; accsh[wv][PACKET + 1] = (double)nfb_w; }
        }
        PEDP_WV(4, __builtin_amdgcn_s_memtime());
        PEDP_WV(11, __builtin_amdgcn_s_memrealtime());
        __syncthreads();
        if (tid < PSTRIDE) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < W; ++w) v += accsh[w][tid];
            double *dst = &a.partials[(size_t)unit * PSTRIDE + tid];
            if (a.fuse) store_sc1(dst, v);
            else *as_global(dst) = v;
        }
        if (!BATCH && !rebuild && tid == 0 && a.dur && unit < a.visit_cap) {
            // what this chunk cost, for the plan after next; a workgroup that shared its CU ran about a third slower
            long long d = (long long)__builtin_amdgcn_s_memtime() - t_chunk0;
            const int extra = n_live - a.n_cu;
            if (extra > 0 && ((int)blockIdx.x < extra || (int)blockIdx.x >= a.n_cu)) d = d * 3 / 4;
            typedef int v2i __attribute__((ext_vector_type(2)));
            const v2i e = {(int)(d < 0x7FFFFFFF ? d : 0x7FFFFFFF), tag_now};
            ((PEDP_GLOBAL v2i *)(uintptr_t)a.dur)[(size_t)(pass & 1) * a.visit_cap + unit] = e;
        }
        if (tid == 0) PEDP_STAMP(1, blockIdx.x, 5);
#if PEDP_ICP_STAMPS
        if (tid == 1 && blockIdx.x < 4096) g_icp_stamps[1][blockIdx.x][7] = ((long long)ntl_w << 32) | (long long)(nfb_w << 16) | nsl;
#endif
        __syncthreads();  // the waves' sums are reused by the next chunk
    }
    PEDP_RT(pass, 2);
    if (!a.fuse) return;
    // ---- the pass is closed by the workgroup that finishes last
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add((g_u32 *)(uintptr_t)a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        misc[4] = prev - ticket_base == (unsigned)(n_wg - 1);
        misc[5] = prev == ticket_base;
    }
    __syncthreads();
    PEDP_RT(pass, 3);
    if (!BATCH && misc[5] && !misc[4] && a.visit) {
        // ---- the first workgroup through writes the next pass's visit plan: all of it, valid or not
        typedef int v2i __attribute__((ext_vector_type(2)));
        const int tid = threadIdx.x, n_cu = a.n_cu, extra = n_live - n_cu;
        PEDP_GLOBAL v4i *plan = (PEDP_GLOBAL v4i *)(uintptr_t)a.visit + (size_t)((pass + 1) & 1) * a.visit_cap;
        const bool want = !rebuild && pass < 65000 && extra > 0 && n_live <= 2 * n_cu && n_live <= (int)gridDim.x && n_live <= W * 64 && n_live <= a.visit_cap;
        int d = 0x7FFFFFFF, mine_ok = 1;
        if (want && tid < n_live) {
            const v2i e = ((const PEDP_GLOBAL v2i *)(uintptr_t)a.dur)[(size_t)((pass + 1) & 1) * a.visit_cap + tid];  // pass - 1 wrote this copy
            mine_ok = e[1] == tag_now - 1;
            d = e[0];
        }
        const int chunk_of_rank = (want && tid < n_live) ? as_global(a.live_list)[tid] : 0;
        if (__syncthreads_and(mine_ok) && want) {
            fin.scan[tid] = d;
            __syncthreads();
            if (tid < n_live) {
                int r = 0;  // rank of this chunk by (duration, live rank)
                for (int q = 0; q < n_live; ++q) {
                    const int dq = fin.scan[q];
                    r += (dq < d || (dq == d && q < tid)) ? 1 : 0;
                }
                const int pos = r < extra ? r : (r < 2 * extra ? n_cu + (2 * extra - 1 - r) : r - extra);
                const v4i e = {tid, chunk_of_rank, tag_now + 1, n_live};
                plan[pos] = e;
            }
        } else {
            const v4i none = {0, 0, 0, 0};
            const int n = (int)gridDim.x < a.visit_cap ? (int)gridDim.x : a.visit_cap;
            for (int i = tid; i < n; i += W * 64) plan[i] = none;
        }
        return;
    }
    if (!misc[4]) return;

    if (threadIdx.x == 0) PEDP_STAMP(1, blockIdx.x, 6);
    FinishArgs f;
    f.live = a.live; f.live_list = a.live_list; f.n_lw = a.n_lw; f.partials = a.partials; f.packet = a.packet; f.phase = 0;
    f.estimator = a.estimator; f.trace = a.trace; f.hist = a.hist; f.bcx = a.bc[0]; f.bcy = a.bc[1]; f.bcz = a.bc[2];
    f.idle = a.ticket + 32;
    f.n_idle = (int)gridDim.x - n_wg;
    f.n_busy = n_wg;
    if (!BATCH && planned && threadIdx.x == 0) st->n_planned += 1;
    f.known = 1; f.k_pass = pass; f.k_rebuild = rebuild ? 1 : 0; f.k_n_live = n_live;
    icp_finish_body<W * 64, 2048, true>(st, f, fin, threadIdx.x);
    PEDP_RT(pass, 4);
}

// Start states of a batch's group: from the page-locked block straight into the poses' state slots (G blocks
// pose_stride apart), and each pose's tickets, sign-off counters and live masks zeroed -- one launch instead of a
// 2-D copy and a 2-D fill (hipMemcpy2DAsync measured 73 us per call in the frame chain's hip trace, eight calls a frame).
__global__ __launch_bounds__(256) void batch_state_scatter_kernel(const unsigned long long *__restrict__ up, char *st0, size_t pose_stride,
                                                                  int state_words, char *zero0, int zero_words) {
    const unsigned long long *src = up + (size_t)blockIdx.x * state_words;
    unsigned long long *dst = (unsigned long long *)(st0 + (size_t)blockIdx.x * pose_stride);
    for (int i = threadIdx.x; i < state_words; i += blockDim.x) dst[i] = src[i];
    unsigned long long *z = (unsigned long long *)(zero0 + (size_t)blockIdx.x * pose_stride);
    for (int i = threadIdx.x; i < zero_words; i += blockDim.x) z[i] = 0ull;
}
// ... and the final states back into the page-locked block (a zero-copy write; the host reads after the stream has finished)
__global__ __launch_bounds__(256) void batch_state_gather_kernel(const char *__restrict__ st0, size_t pose_stride, int state_words,
                                                                 unsigned long long *__restrict__ down) {
    const unsigned long long *src = (const unsigned long long *)(st0 + (size_t)blockIdx.x * pose_stride);
    unsigned long long *dst = down + (size_t)blockIdx.x * state_words;
    for (int i = threadIdx.x; i < state_words; i += blockDim.x) dst[i] = src[i];
}

// ------------------------------------------------------------------ host side

struct TargetPrep {
    double c[3];
    float Tn, T2;
    double lo[3], hi[3];
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct IcpWorkspace {
    IcpState *st;
    double *P, *d2, *partials, *packet, *trace;
    float4 *B, *blk_sph;
    const float4 *tgt4, *tile_sph;
    const int32_t *src_perm, *tgt_perm;
    float *eps, *S;
    int32_t *idx, *fb, *list;
    unsigned long long *mask;                         // [blocks_cap][n_words] surviving-tile bits
    int32_t *blk_cnt, *blk_segstart;                  // [blocks_cap], [blocks_cap + 1]
    int32_t *seg_blk, *seg_rank0, *seg_n;             // [max_segs]
    float *tr_b1, *tr_b2;
    int32_t *tr_t1;
    int64_t Ns_pad, Nt_pad, blocks_cap;
    int n_words, max_segs;
    int qt;  // MFMA tiles per target unit for this call (1: culled registration, 4: dense sweep)
    // fused pass (qt == 1 inside a registration)
    bool fused;
    double *Pk, *hist, *cpart, *Tprev;
    unsigned *ticket;
    unsigned long long *live;
    int32_t *live_list;
    int4 *visit;   // visit plan and chunk durations (single registration; see PassArgs)
    int2 *dur;
    int visit_cap;
    int n_lw, n_chunks;
    const float4 *word_sph;
    const double *tgt_s;
    size_t pose_stride;  // bytes between the per-pose blocks of a batch (all pointers above are pose 0's)
};

// trace and update history are sized for a capacity class of max_iter, not for max_iter itself: the
// per-pose block layout a captured batch graph bakes in then changes only with the class
inline int iter_capacity(int max_iter) { return (max_iter + 2 + 63) / 64 * 64; }

int carve_workspace(pedp_ctx_t c, int64_t Ns, int64_t Nt, int max_iter, int qt, IcpWorkspace &w, bool fused = false,
                    int poses = 1) {
    w.qt = qt;
    w.fused = fused;
    w.Ns_pad = (int64_t)align_up((size_t)(Ns > 0 ? Ns : 1), NN_PTS_PER_WG);
    w.Nt_pad = (int64_t)align_up((size_t)(Nt > 0 ? Nt : 1), 16 * NN_TU);
    w.blocks_cap = w.Ns_pad / (NN_SB * 16) + 1;
    const int64_t n_tiles = w.Nt_pad / (16 * qt);  // target units
    w.n_words = (int)((n_tiles + 63) / 64);
    // segment table: enough pieces that even the dense case (every tile survives for every
    // block) keeps a piece within the LDS list of one wave
    const int64_t list_units = NN_LIST_TILES / qt;
    int64_t ms = (w.blocks_cap * n_tiles + list_units - 1) / list_units + w.blocks_cap;
    // room for shortest-length (SEG_MIN) pieces up to 8192 of them: the table bounds the sweep's
    // grid, so a small scene (a few blocks) gets a small grid instead of thousands of idle workgroups
    const int64_t seg_min_units = SEG_MIN_TILES / qt;
    int64_t fine = (w.blocks_cap * n_tiles + seg_min_units - 1) / seg_min_units + w.blocks_cap;
    if (fine > 8192) fine = 8192;
    if (ms < fine) ms = fine;
    if (fused) ms = 1;  // the fused pass keeps its triples in LDS: no segment table
    PEDP_REQUIRE(ms < (int64_t)1 << 22, "pedp_icp: problem too large for the segment table (%lld x %lld points)",
                 (long long)Ns, (long long)Nt);
    w.max_segs = (int)ms;
    w.n_chunks = (int)((Ns + CH - 1) / CH);
    w.n_lw = (w.n_chunks + 63) / 64;
    if (w.n_lw < 1) w.n_lw = 1;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    // what only the segmented (dense / large-radius) path uses shrinks to nothing in the fused pass,
    // which keeps a chunk's slots, masks and triples in LDS
    const size_t seg_only = fused ? 0 : 1;
    size_t o_st = take(sizeof(IcpState));
    size_t o_P = take(sizeof(double) * 3 * (size_t)w.Ns_pad * seg_only);
    size_t o_d2 = take(sizeof(double) * (size_t)w.Ns_pad * seg_only);
    size_t o_part = take(sizeof(double) * ACC_BLOCKS * PACKET);
    size_t o_pack = take(sizeof(double) * 32);
    size_t o_trace = take(sizeof(double) * 18 * (size_t)iter_capacity(max_iter));
    size_t o_B = take(sizeof(float4) * (size_t)w.Ns_pad);
    size_t o_eps = take(sizeof(float) * (size_t)w.Ns_pad);
    size_t o_S = take(sizeof(float) * (size_t)w.Ns_pad);
    size_t o_idx = take(sizeof(int32_t) * (size_t)w.Ns_pad);
    size_t o_fb = take(sizeof(int32_t) * (size_t)w.Ns_pad);
    size_t o_list = take(sizeof(int32_t) * (size_t)w.Ns_pad);
    size_t o_bsph = take(sizeof(float4) * NN_SB * (size_t)w.blocks_cap);
    size_t o_mask = take(sizeof(unsigned long long) * (size_t)w.blocks_cap * (size_t)w.n_words);
    size_t o_bcnt = take(sizeof(int32_t) * (size_t)w.blocks_cap);
    size_t o_bseg = take(sizeof(int32_t) * (size_t)(w.blocks_cap + 1));
    size_t o_sblk = take(sizeof(int32_t) * (size_t)w.max_segs);
    size_t o_srk = take(sizeof(int32_t) * (size_t)w.max_segs);
    size_t o_sn = take(sizeof(int32_t) * (size_t)w.max_segs);
    const size_t tr = (size_t)w.max_segs * 4 * (NN_SB * 16);
    size_t o_b1 = take(sizeof(float) * tr);
    size_t o_t1 = take(sizeof(int32_t) * tr);
    size_t o_b2 = take(sizeof(float) * tr);
    size_t o_Pk = 0, o_hist = 0, o_cpart = 0, o_live = 0, o_llist = 0, o_tprev = 0, o_ticket = 0, o_visit = 0;
    w.visit_cap = 0;
    if (fused) {
        o_Pk = take(sizeof(double) * 3 * (size_t)w.Ns_pad * 2);     // read from [pass & 1], written to the other: no wave
        o_tprev = take(sizeof(double) * 3 * (size_t)w.Ns_pad * 2);  // ever reads what a faster wave of the same pass has rewritten
        o_hist = take(sizeof(double) * 16 * (size_t)iter_capacity(max_iter));
        o_cpart = take(sizeof(double) * PSTRIDE * (size_t)w.blocks_cap);
        o_ticket = take(17 * 128);  // the ticket and the sixteen sign-off counters, a line each; the live masks right behind (one memset)
        o_live = take(sizeof(unsigned long long) * 2 * (size_t)w.n_lw);  // live mask + the mask before the last rebuild
        o_llist = take(sizeof(int32_t) * (size_t)w.blocks_cap);
        if (poses <= 1) {  // a batch fills every CU several times over anyway
            w.visit_cap = (int)(w.blocks_cap < 2 * c->num_cus ? w.blocks_cap : 2 * c->num_cus);
            o_visit = take((sizeof(int4) + sizeof(int2)) * 2 * (size_t)w.visit_cap);
        }
    }
    off = align_up(off, 4096);
    w.pose_stride = off;  // a batch lays `poses` such blocks one behind the other
    int st = c->icp_ws.reserve(off * (size_t)(poses > 0 ? poses : 1));
    if (st) return st;
    char *b = (char *)c->icp_ws.ptr;
    w.st = (IcpState *)(b + o_st);
    w.P = (double *)(b + o_P);
    w.d2 = (double *)(b + o_d2);
    w.partials = (double *)(b + o_part);
    w.packet = (double *)(b + o_pack);
    w.trace = (double *)(b + o_trace);
    w.B = (float4 *)(b + o_B);
    w.eps = (float *)(b + o_eps);
    w.S = (float *)(b + o_S);
    w.idx = (int32_t *)(b + o_idx);
    w.fb = (int32_t *)(b + o_fb);
    w.list = (int32_t *)(b + o_list);
    w.blk_sph = (float4 *)(b + o_bsph);
    w.mask = (unsigned long long *)(b + o_mask);
    w.blk_cnt = (int32_t *)(b + o_bcnt);
    w.blk_segstart = (int32_t *)(b + o_bseg);
    w.seg_blk = (int32_t *)(b + o_sblk);
    w.seg_rank0 = (int32_t *)(b + o_srk);
    w.seg_n = (int32_t *)(b + o_sn);
    w.tr_b1 = (float *)(b + o_b1);
    w.tr_t1 = (int32_t *)(b + o_t1);
    w.tr_b2 = (float *)(b + o_b2);
    w.Pk = (double *)(b + o_Pk);
    w.Tprev = (double *)(b + o_tprev);
    w.hist = (double *)(b + o_hist);
    w.cpart = (double *)(b + o_cpart);
    w.live = (unsigned long long *)(b + o_live);
    w.live_list = (int32_t *)(b + o_llist);
    w.ticket = (unsigned *)(b + o_ticket);
    w.visit = w.visit_cap ? (int4 *)(b + o_visit) : nullptr;
    w.dur = w.visit_cap ? (int2 *)(b + o_visit + sizeof(int4) * 2 * (size_t)w.visit_cap) : nullptr;
    // the per-block survivor counters start at zero (the segment kernel re-zeroes them per pass)
    if (!fused) PEDP_HIP_CHECK(hipMemsetAsync(w.blk_cnt, 0, sizeof(int32_t) * (size_t)w.blocks_cap, c->stream));
    return PEDP_OK;
}

// Enqueue one correspondence pass (transform, sweep, fallback).  mode as in
// icp_transform_pack_kernel.
inline bool nn_bf16_sweep() {  // PEDP_NN_F32=1: the dense sweep on the f32-input MFMA (A/B, tests)
    static const bool off = getenv("PEDP_NN_F32") && atoi(getenv("PEDP_NN_F32")) != 0;
    return !off;
}
int enqueue_nn_pass(pedp_ctx_t c, const IcpWorkspace &w, pedp_cloud_t src, pedp_cloud_t tgt, int mode,
                    const TargetPrep &tp, double r, hipEvent_t ev0, hipEvent_t ev1, bool exhaustive = false) {
    const int64_t Ns = src->N, Nt = tgt->N;
    // r1: distance scale of the candidates the bound must hold for (anything farther is
    // not an inlier anyway); huge radii fall back to the cloud scale inside the kernel.
    const float r1 = (float)(r * 1.01);
    // exhaustive: the box test and the sphere culling use an infinite radius (every point is a
    // candidate, every mask bit is set); the selection still applies r, so results do not change
    const double r_cull = exhaustive ? 1e18 : r;
    const double r2cut = r_cull * r_cull * (1.0 + 1e-12);
    const float r_search = (float)(r_cull * (1.0 + 1e-6)) + 1e-6f;
    const int n_tiles = (int)(w.Nt_pad / (16 * w.qt));  // target units
    // the dense sweep (units of four tiles) runs on the bf16 matrix pipe (PEDP_NN_F32=1: the f32-input MFMA of rounds 1-3)
    const bool bf16_sweep = w.qt == 4 && nn_bf16_sweep() && tgt->tgt_bf != nullptr;
    {
        int64_t grid = (Ns + 511) / 512;  // 128 points per wave, 4 waves per workgroup
        hipLaunchKernelGGL(icp_transform_pack_kernel, dim3((unsigned)grid), dim3(256), 0, c->stream, w.st, mode,
                           src->pts, w.P, w.src_perm, Ns, w.B, w.eps, w.S, w.list, w.blk_sph, w.idx, w.d2, tp.Tn, tp.T2,
                           r1, r2cut, tp.lo[0], tp.lo[1], tp.lo[2], tp.hi[0], tp.hi[1], tp.hi[2], bf16_sweep ? 34.0f : 5.0f);
    }
    const float r2f = (float)(r * r) * 1.00001f;
    const unsigned sel_grid = (unsigned)((4 * w.Ns_pad + 255) / 256);
    const unsigned sweep_grid = (unsigned)((w.max_segs + NN_WAVES - 1) / NN_WAVES);
    // ambiguous slots are a small fraction of the scene: one workgroup per 8 slots at most, two per CU at most
    int64_t fbg = w.Ns_pad / 8;
    if (fbg > 2 * c->num_cus) fbg = 2 * c->num_cus;
    if (fbg < 1) fbg = 1;
    const unsigned fb_grid = (unsigned)fbg;
    {
        const int64_t groups = (w.n_words + CULL_WORDS - 1) / CULL_WORDS;
        const int64_t waves = w.blocks_cap * groups;
        hipLaunchKernelGGL(nn_cull_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, c->stream, w.st, w.tile_sph,
                           n_tiles, w.n_words, w.blk_sph, r_search, w.mask, w.blk_cnt);
        hipLaunchKernelGGL(nn_segment_kernel, dim3(1), dim3(1024), 0, c->stream, w.st, w.blk_cnt, w.blk_segstart,
                           w.seg_blk, w.seg_rank0, w.seg_n, w.max_segs, SEG_MIN_TILES / w.qt, NN_LIST_TILES / w.qt);
    }
#define PEDP_NN_STAGE(QTV, GV)                                                                                          \
    do {                                                                                                              \
        if (ev0) PEDP_HIP_CHECK(hipEventRecord(ev0, c->stream));                                                      \
        hipLaunchKernelGGL((nn_sweep_kernel<QTV, GV>), dim3(sweep_grid), dim3(NN_WAVES * 64), 0, c->stream, w.st,            \
                           (const float *)w.tgt4, n_tiles, w.n_words, w.mask, w.seg_blk, w.seg_rank0, w.seg_n,         \
                           (const float *)w.B, w.tr_b1, w.tr_t1, w.tr_b2);                                            \
        if (ev1) { PEDP_HIP_CHECK(hipEventRecord(ev1, c->stream)); c->nn_timed = true; }                               \
        hipLaunchKernelGGL(nn_select_kernel<QTV>, dim3(sel_grid), dim3(256), 0, c->stream, w.st, w.blk_segstart,       \
                           w.tr_b1, w.tr_t1, w.tr_b2, tgt->pts, w.tgt_perm, Nt, w.P, w.eps, w.S, w.list, r2f, w.idx,   \
                           w.d2, w.fb);                                                                               \
        hipLaunchKernelGGL(nn_fallback_kernel<QTV>, dim3(fb_grid), dim3(FB_WAVES * 64), 0, c->stream, w.st, w.fb, w.list,     \
                           w.mask, w.n_words, w.tile_sph, r_search, tgt->pts, w.tgt_perm, Nt, w.P, w.idx, w.d2);       \
    } while (0)
    if (bf16_sweep) {
        if (ev0) PEDP_HIP_CHECK(hipEventRecord(ev0, c->stream));
        hipLaunchKernelGGL((nn_sweep_bf16_kernel<4, 2>), dim3(sweep_grid), dim3(NN_WAVES * 64), 0, c->stream, w.st,
                           (const uint4 *)tgt->tgt_bf, n_tiles, w.n_words, w.mask, w.seg_blk, w.seg_rank0, w.seg_n,
                           (const float4 *)w.B, w.tr_b1, w.tr_t1, w.tr_b2);
        if (ev1) { PEDP_HIP_CHECK(hipEventRecord(ev1, c->stream)); c->nn_timed = true; }
        hipLaunchKernelGGL(nn_select_kernel<4>, dim3(sel_grid), dim3(256), 0, c->stream, w.st, w.blk_segstart,
                           w.tr_b1, w.tr_t1, w.tr_b2, tgt->pts, w.tgt_perm, Nt, w.P, w.eps, w.S, w.list, r2f, w.idx,
                           w.d2, w.fb);
        hipLaunchKernelGGL(nn_fallback_kernel<4>, dim3(fb_grid), dim3(FB_WAVES * 64), 0, c->stream, w.st, w.fb, w.list,
                           w.mask, w.n_words, w.tile_sph, r_search, tgt->pts, w.tgt_perm, Nt, w.P, w.idx, w.d2);
    } else if (w.qt == 4) PEDP_NN_STAGE(4, 2);
    else PEDP_NN_STAGE(1, 4);
#undef PEDP_NN_STAGE
    PEDP_HIP_CHECK(hipGetLastError());
    return PEDP_OK;
}

// Margin of the live set beyond the correspondence radius (see the fused-pass comment).
inline double fused_margin(double r) { return 2.0 * r; }   // (1 r ... 4 r measured on the bench registration: 0.702-0.720 ms, no trend; any margin > 0 is exact)

// Test hook (tests/test_icp_gpu.py): PEDP_ICP_UNFUSED_FINISH=1 closes every pass with a launch of
// icp_finish_kernel instead of the in-launch hand-over -- the results must not differ in any bit.
inline bool unfused_finish() {
    static const bool m = getenv("PEDP_ICP_UNFUSED_FINISH") && atoi(getenv("PEDP_ICP_UNFUSED_FINISH")) != 0;
    return m;
}
inline bool no_visit_plan() {  // PEDP_ICP_NO_VISIT_PLAN=1: every workgroup takes the live chunk of its own index (tests compare the two)
    static const bool m = getenv("PEDP_ICP_NO_VISIT_PLAN") && atoi(getenv("PEDP_ICP_NO_VISIT_PLAN")) != 0;
    return m;
}
// Enqueue the kernel of a fused pass.  fuse: the workgroup that finishes last closes the pass
// (sum, solve, update); otherwise icp_finish_kernel launches follow (exchange step in between).
int enqueue_fused_pass(pedp_ctx_t c, const IcpWorkspace &w, pedp_cloud_t src, pedp_cloud_t tgt, int estimator,
                       const TargetPrep &tp, hipEvent_t ev0, hipEvent_t ev1, bool fuse, double *trace, int poses = 1) {
    PassArgs pa;
    pa.src = src->pts; pa.perm = w.src_perm; pa.N = src->N; pa.n_chunks = w.n_chunks;
    pa.hist = w.hist; pa.Pk = w.Pk; pa.Tprev = w.Tprev; pa.live = w.live; pa.live_list = w.live_list;
    pa.chunk_sph = (const double *)src->chunk_sph;
    pa.pp_stride = 3 * (size_t)w.Ns_pad;
    pa.tgtf = (const float *)w.tgt4; pa.n_tiles = (int)(w.Nt_pad / 16); pa.n_words = w.n_words;
    pa.tile_sph = w.tile_sph; pa.word_sph = w.word_sph; pa.tgt_s = w.tgt_s; pa.tperm = w.tgt_perm; pa.Nt = tgt->N;
    pa.tgt = tgt->pts; pa.nrm = tgt->normals;
    pa.Tn = tp.Tn; pa.T2 = tp.T2;
    for (int k = 0; k < 3; ++k) { pa.lo[k] = tp.lo[k]; pa.hi[k] = tp.hi[k]; pa.bc[k] = 0.5 * (tp.lo[k] + tp.hi[k]); }
    pa.estimator = estimator; pa.idx_out = w.idx; pa.partials = w.cpart;
    pa.pose_stride = poses > 1 ? w.pose_stride : 0;
    pa.fuse = fuse && !unfused_finish() ? 1 : 0; pa.n_lw = w.n_lw; pa.packet = w.packet; pa.trace = trace; pa.ticket = w.ticket;
    const bool plan = poses <= 1 && !no_visit_plan();
    pa.visit = plan ? w.visit : nullptr; pa.dur = plan ? w.dur : nullptr; pa.visit_cap = w.visit_cap; pa.n_cu = c->num_cus;
    // grid-stride loop over the live chunks: any grid is correct; two workgroups per CU are resident
    int64_t g = w.n_chunks;
    if (g > 2 * c->num_cus) g = 2 * c->num_cus;
    if (g < 1) g = 1;
    if (ev0) PEDP_HIP_CHECK(hipEventRecord(ev0, c->stream));
    if (poses > 1)
        hipLaunchKernelGGL((icp_pass_kernel<BK_W, true>), dim3((unsigned)g, (unsigned)poses), dim3(BK_W * 64), 0, c->stream, w.st, pa);
    else
        hipLaunchKernelGGL((icp_pass_kernel<BK_W, false>), dim3((unsigned)g), dim3(BK_W * 64), 0, c->stream, w.st, pa);
    if (ev1) { PEDP_HIP_CHECK(hipEventRecord(ev1, c->stream)); c->nn_timed = true; }
    PEDP_HIP_CHECK(hipGetLastError());
    return PEDP_OK;
}

// Spatial order of a cloud, cached in the handle: a function of the cloud alone (cells over its own
// bounding box), built once, stream-ordered in pooled scratch (no allocation besides the order
// itself, no synchronisation).
int ensure_spatial_perm(pedp_ctx_t c, pedp_cloud_t cl) {
    if (cl->N == 0 || cl->perm) return PEDP_OK;
    const int64_t n_chunks = (cl->N + 127) / 128;
    size_t perm_cap = 0, sph_cap = 0;
    void *perm = c->cloud_pool.take(sizeof(int32_t) * (size_t)cl->N, &perm_cap);
    void *sph = c->cloud_pool.take(sizeof(double) * 32 * (size_t)n_chunks, &sph_cap);
    auto fail = [&](int rc, const char *what) {
        (void)hipStreamSynchronize(c->stream);
        c->cloud_pool.give(perm, perm_cap);
        c->cloud_pool.give(sph, sph_cap);
        if (what) pedp_set_error("pedp_icp: spatial order: %s", what);
        return rc;
    };
    if (!perm || !sph) return fail(PEDP_ERR_ALLOC, "allocation failed");
    unsigned long long *keys = nullptr;
    int rcs = pedp_sort_keys64_begin(c, cl->N, SORT_KEY_BITS, &keys);
    if (rcs) return fail(rcs, nullptr);
    const unsigned grid = (unsigned)((cl->N + 255) / 256);
    hipLaunchKernelGGL(cell_key_kernel, dim3(grid), dim3(256), 0, c->stream, cl->pts, cl->N, (const double *)cl->d_box, keys);
    rcs = pedp_sort_keys64_run(c, cl->N, SORT_KEY_BITS, (int32_t *)perm);
    if (rcs) return fail(rcs, nullptr);
    hipLaunchKernelGGL(chunk_sphere_kernel, dim3((unsigned)n_chunks), dim3(64), 0, c->stream, cl->pts, (const int32_t *)perm, cl->N,
                       (double *)sph);
    if (hipGetLastError() != hipSuccess) return fail(PEDP_ERR_HIP, "launch failed");
    cl->perm = perm; cl->perm_cap = perm_cap;
    cl->chunk_sph = sph; cl->sph_cap = sph_cap;
    return PEDP_OK;
}

// Target-side operand of the sweep, built once per cloud and kept in the handle (the
// reference re-runs ICP ~50x per frame against the same model, pose_estimation.py:577-613).
int ensure_target_pack(pedp_ctx_t c, pedp_cloud_t tgt, TargetPrep &tp) {
    { const int rs = pedp_cloud_host_stats(c, tgt); if (rs) return rs; }
    for (int k = 0; k < 3; ++k) tp.c[k] = tgt->centroid[k];
    tp.Tn = tgt->Tn;
    tp.T2 = tgt->T2;
    for (int k = 0; k < 3; ++k) { tp.lo[k] = tgt->lo[k]; tp.hi[k] = tgt->hi[k]; }
    if (tgt->tgt4) return PEDP_OK;
    int rc = ensure_spatial_perm(c, tgt);
    if (rc) return rc;
    // real tiles rounded to NN_TU, plus readable pad tiles the pipelined sweep may prefetch
    int64_t pad = (int64_t)align_up((size_t)(tgt->N > 0 ? tgt->N : 1), 16 * NN_TU) + 16 * NN_TILE_PAD;
    // all three or none: a half-built pack must not look finished to the next call
    void *t4 = nullptr, *s1 = nullptr, *s4 = nullptr, *sw = nullptr, *ts = nullptr;
    const int64_t n_wsph = (pad / 16 + 63) / 64;  // one sphere per mask word of 16-row tiles (1024 rows)
    hipError_t e = hipMalloc(&t4, sizeof(float4) * (size_t)pad);
    if (e == hipSuccess) e = hipMalloc(&s1, sizeof(float4) * (size_t)(pad / 16));
    if (e == hipSuccess) e = hipMalloc(&s4, sizeof(float4) * (size_t)(pad / 64));
    if (e == hipSuccess) e = hipMalloc(&sw, sizeof(float4) * (size_t)n_wsph);
    if (e == hipSuccess) e = hipMalloc(&ts, sizeof(double) * 6 * (size_t)pad);
    if (e != hipSuccess) {
        if (t4) (void)hipFree(t4);
        if (s1) (void)hipFree(s1);
        if (s4) (void)hipFree(s4);
        if (sw) (void)hipFree(sw);
        if (ts) (void)hipFree(ts);
        pedp_set_error("pedp_icp: target pack allocation failed: %s", hipGetErrorString(e));
        return PEDP_ERR_ALLOC;
    }
    tgt->tgt4 = t4;
    tgt->tile_sph = s1;
    tgt->tile_sph4 = s4;
    tgt->tile_sphw = sw;
    tgt->tgt_s = ts;
    tgt->tgt4_pad = pad;
    int64_t grid = (pad + 255) / 256;
    hipLaunchKernelGGL(pack_target_kernel, dim3((unsigned)grid), dim3(256), 0, c->stream, tgt->pts,
                       (const int32_t *)tgt->perm, tgt->N, pad, tp.c[0], tp.c[1], tp.c[2], (float4 *)tgt->tgt4);
    hipLaunchKernelGGL(tile_sphere_kernel, dim3((unsigned)((pad / 16 + 255) / 256)), dim3(256), 0, c->stream,
                       (const float4 *)tgt->tgt4, tgt->N, pad / 16, 16, (float4 *)tgt->tile_sph);
    hipLaunchKernelGGL(tile_sphere_kernel, dim3((unsigned)((pad / 64 + 255) / 256)), dim3(256), 0, c->stream,
                       (const float4 *)tgt->tgt4, tgt->N, pad / 64, 64, (float4 *)tgt->tile_sph4);
    hipLaunchKernelGGL(tile_sphere_kernel, dim3((unsigned)((n_wsph + 63) / 64)), dim3(64), 0, c->stream,
                       (const float4 *)tgt->tgt4, tgt->N, n_wsph, 1024, (float4 *)tgt->tile_sphw);
    hipLaunchKernelGGL(sort_rows_kernel, dim3((unsigned)grid), dim3(256), 0, c->stream, tgt->pts, tgt->normals,
                       (const int32_t *)tgt->perm, tgt->N, pad, (double *)tgt->tgt_s);
    PEDP_HIP_CHECK(hipGetLastError());
    return PEDP_OK;
}

// The bf16 pieces of the sorted operand, for the dense sweep: built when a call first takes that path
int ensure_target_bf16(pedp_ctx_t c, pedp_cloud_t tgt) {
    if (tgt->tgt_bf || !tgt->tgt4 || !nn_bf16_sweep()) return PEDP_OK;
    void *p = nullptr;
    if (hipMalloc(&p, sizeof(uint4) * 4 * (size_t)tgt->tgt4_pad) != hipSuccess) {
        pedp_set_error("pedp_icp: target pack allocation failed (bf16 operand)");
        return PEDP_ERR_ALLOC;
    }
    tgt->tgt_bf = p;
    const int64_t n = tgt->tgt4_pad * 4;
    hipLaunchKernelGGL(pack_target_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const float4 *)tgt->tgt4,
                       tgt->tgt4_pad, (uint4 *)tgt->tgt_bf);
    PEDP_HIP_CHECK(hipGetLastError());
    return PEDP_OK;
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {

// One registration = enqueue (all passes, on the executor's stream and workspace) + collect
// (one synchronisation, results to the host).  The executor is the clouds' own context for
// pedp_icp and one of its sub-contexts (own stream + workspace) for pedp_icp_batched, where
// several registrations are in flight at once.
struct IcpJob {
    IcpWorkspace w;
    int max_iter = 0, qt = 1;
    int64_t Ns = 0, Nt = 0;
    bool exhaustive = false;
    int timed_pass = -1;
};

int icp_check_args(pedp_ctx_t c, pedp_cloud_t source, pedp_cloud_t target, const pedp_icp_params *prm) {
    PEDP_REQUIRE(source->ctx == c && target->ctx == c, "pedp_icp: cloud belongs to another context");
    PEDP_REQUIRE(prm->estimator == PEDP_POINT_TO_PLANE || prm->estimator == PEDP_POINT_TO_POINT,
                 "pedp_icp: unknown estimator %d", prm->estimator);
    PEDP_REQUIRE(prm->max_iteration >= 0 && prm->max_iteration <= 100000, "pedp_icp: max_iteration out of range");
    PEDP_REQUIRE(!prm->use_comm || c->comm, "pedp_icp: use_comm is set but the context has no communicator (pedp_comm_create)");
    PEDP_REQUIRE(!(prm->use_comm && prm->allreduce), "pedp_icp: use_comm and an all-reduce hook are exclusive");
    if (prm->estimator == PEDP_POINT_TO_PLANE && !target->has_normals) {
        pedp_set_error("pedp_icp: TransformationEstimationPointToPlane requires target normals");
        return PEDP_ERR_NO_NORMALS;
    }
    return PEDP_OK;
}

// unit size: fine units while the radius is small against the model (culling decides the
// cost), 64-row units when the sweep is dense anyway (cheaper epilogue)
int icp_unit_size(pedp_cloud_t target, double r) {
    double diag2 = 0.0;
    for (int k = 0; k < 3; ++k) diag2 += (target->hi[k] - target->lo[k]) * (target->hi[k] - target->lo[k]);
    return (r * r < diag2 / 16.0) ? 1 : 4;
}

// Cached per-cloud preparation on the owner's stream: sorted target operand + unit spheres, spatial
// order of the scene.
int icp_prepare(pedp_ctx_t c, pedp_cloud_t source, pedp_cloud_t target, TargetPrep &tp) {
    int rc = ensure_target_pack(c, target, tp);
    if (rc) return rc;
    rc = ensure_target_bf16(c, target);   // (with the pack, on the owner's stream: sub-contexts start after it is complete)
    if (rc) return rc;
    return ensure_spatial_perm(c, source);
}

// workspace of one registration on executor x (may grow the executor's scratch buffer)
int icp_job_setup(pedp_ctx_t x, pedp_cloud_t source, pedp_cloud_t target, const pedp_icp_params *prm, IcpJob &job) {
    PEDP_HIP_CHECK(hipSetDevice(x->device));
    job.max_iter = prm->max_iteration;
    job.Ns = source->N;
    job.Nt = target->N;
    job.qt = x->icp_exhaustive ? 4 : icp_unit_size(target, prm->max_correspondence_distance);
    job.exhaustive = x->icp_exhaustive;
    job.timed_pass = x->icp_timed_pass;
    IcpWorkspace &w = job.w;
    const double r = prm->max_correspondence_distance;
    // the fused pass lists a target's mask words in LDS: up to BK_WCAP words of 1024 rows
    const bool fused = job.qt == 1 && !job.exhaustive && r > 0.0 && job.Ns > 0 && job.Nt > 0 &&
                       (job.Nt + 1023) / 1024 <= BK_WCAP;
    int rc = carve_workspace(x, job.Ns, job.Nt, job.max_iter, job.qt, w, fused);
    if (rc) return rc;
    w.tgt4 = (const float4 *)target->tgt4;
    w.word_sph = (const float4 *)target->tile_sphw;
    w.tgt_s = (const double *)target->tgt_s;
    w.tile_sph = (const float4 *)(job.qt == 4 ? target->tile_sph4 : target->tile_sph);
    w.tgt_perm = (const int32_t *)target->perm;
    w.src_perm = (const int32_t *)source->perm;
    return PEDP_OK;
}

// start state of a registration in the executor's pinned block (uploaded by the first node of
// icp_enqueue; the same block receives the final state)
void icp_fill_state(IcpState *dst, const TargetPrep &tp, const double init[16], const pedp_icp_params *prm, int64_t Ns) {
    IcpState h{};
    static std::atomic<int> registration_counter{0};
    h.nonce = ((registration_counter.fetch_add(1, std::memory_order_relaxed) & 0x3FFF) + 1) << 16;
    for (int k = 0; k < 16; ++k) { h.T[k] = init[k]; h.upd[k] = init[k]; h.T_init[k] = init[k]; }
    for (int k = 0; k < 3; ++k) h.centroid[k] = tp.c[k];
    h.rebuild = 1;  // fused pass: pass 0 builds the live chunk set from the whole scene
    const double r = prm->max_correspondence_distance, margin = fused_margin(r);
    double rho2 = 0.0;
    for (int k = 0; k < 3; ++k) rho2 += 0.25 * (tp.hi[k] - tp.lo[k]) * (tp.hi[k] - tp.lo[k]);
    h.r2 = r * r;
    h.r2cut = r * r * (1.0 + 1e-12);
    h.r2live = (r + margin) * (r + margin) * (1.0 + 1e-12);
    h.reachE = r + margin + std::sqrt(rho2);
    h.margin = margin;
    h.rel_fitness = prm->relative_fitness;
    h.rel_rmse = prm->relative_rmse;
    const double ng = prm->n_source_global > 0 ? (double)prm->n_source_global : (double)Ns;
    h.n_source = ng > 0 ? ng : 1.0;
    h.r1 = (float)(r * 1.01);
    h.r_search = (float)(r * (1.0 + 1e-6)) + 1e-6f;
    h.wide_radius = (float)r;
    h.r2f = (float)(r * r) * 1.00001f;
    h.max_iter = prm->max_iteration;
    *dst = h;
}

// every pass of one registration on x's stream; nothing here allocates or synchronises unless
// early_stop is set, so the sequence can be captured into a graph
int icp_enqueue(pedp_ctx_t x, pedp_cloud_t source, pedp_cloud_t target, const TargetPrep &tp,
                const pedp_icp_params *prm, bool want_trace, bool early_stop, const IcpJob &job) {
    const int64_t Ns = source->N, Nt = target->N;
    const int max_iter = prm->max_iteration;
    const double r = prm->max_correspondence_distance;
    const double n_global = prm->n_source_global > 0 ? (double)prm->n_source_global : (double)Ns;
    const IcpWorkspace &w = job.w;
    int rc;
    // Open3D: max_correspondence_distance <= 0 or an empty cloud gives an empty result
    const bool degenerate = (r <= 0.0 || Ns == 0 || Nt == 0);
    IcpState *hp = (IcpState *)x->pinned;
    PEDP_HIP_CHECK(hipMemcpyAsync(w.st, hp, sizeof(IcpState), hipMemcpyHostToDevice, x->stream));
    const double r2 = r * r;
    const double ng = n_global > 0 ? n_global : 1.0;
    const bool exchange = prm->allreduce || prm->use_comm;  // the packet is summed over ranks before the solve
    const bool fused = w.fused && !degenerate;
    double bc[3] = {0, 0, 0};
    if (fused) {
        // tickets, sign-off counters and both live masks start at zero: one memset (they lie one behind the other); the
        // start transformation -- slot 0 of the history -- arrives inside the state; pass 0 itself writes "no
        // correspondence" for every chunk it does not visit
        PEDP_HIP_CHECK(hipMemsetAsync(w.ticket, 0, (size_t)((char *)w.live - (char *)w.ticket) + sizeof(unsigned long long) * 2 * (size_t)w.n_lw, x->stream));
        for (int k = 0; k < 3; ++k) bc[k] = 0.5 * (tp.lo[k] + tp.hi[k]);
    }
    if (job.timed_pass != -1) { x->nn_pairs = 0; x->nn_span_launches = 1; }
    for (int pass = 0; pass <= max_iter; ++pass) {
        // timing: one chosen pass (pair nn_ev0/1), or -- timed_pass = -2 -- every fourth pass from
        // pass 1 on, up to eight pairs, whose mean pedp_nn_last_sweep_ms reports
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (pass == job.timed_pass) { ev0 = x->nn_ev0; ev1 = x->nn_ev1; }
        if (job.timed_pass == -3) {  // ONE pair around all the passes' launches: the mean launch-to-launch span
            if (pass == 0) { ev0 = x->nn_ev0; x->nn_span_launches = max_iter + 1; }
            if (pass == max_iter) ev1 = x->nn_ev1;
        }
        if (job.timed_pass == -2 && (pass & 3) == 1 && x->nn_pairs < 8) {
            for (int k = 0; k < 2; ++k)
                if (!x->nn_evs[2 * x->nn_pairs + k]) PEDP_HIP_CHECK(hipEventCreate(&x->nn_evs[2 * x->nn_pairs + k]));
            ev0 = x->nn_evs[2 * x->nn_pairs];
            ev1 = x->nn_evs[2 * x->nn_pairs + 1];
            ++x->nn_pairs;
        }
        if (fused) {
            rc = enqueue_fused_pass(x, w, source, target, prm->estimator, tp, ev0, ev1, !exchange, want_trace ? w.trace : nullptr);
            if (rc) return rc;
            if (!exchange && unfused_finish())
                hipLaunchKernelGGL(icp_finish_kernel, dim3(1), dim3(FIN_THREADS), 0, x->stream, w.st, w.live, w.live_list, w.n_lw, w.cpart, w.packet, 0,
                                   prm->estimator, want_trace ? w.trace : nullptr, w.hist, bc[0], bc[1], bc[2], (size_t)0);
            if (exchange) {  // sum -> all-reduce over the ranks -> solve
                hipLaunchKernelGGL(icp_finish_kernel, dim3(1), dim3(FIN_THREADS), 0, x->stream, w.st, w.live, w.live_list, w.n_lw, w.cpart, w.packet, 1,
                                   prm->estimator, want_trace ? w.trace : nullptr, w.hist, bc[0], bc[1], bc[2], (size_t)0);
                if (prm->use_comm) {
                    rc = pedp_comm_allreduce_sum_f64(x, w.packet, PACKET);
                    if (rc) { (void)hipStreamSynchronize(x->stream); return rc; }
                } else if (prm->allreduce(prm->allreduce_user, w.packet, PACKET, (void *)x->stream) != 0) {
                    pedp_set_error("pedp_icp: all-reduce hook failed in pass %d", pass);
                    (void)hipStreamSynchronize(x->stream);
                    return PEDP_ERR_COLLECTIVE;
                }
                hipLaunchKernelGGL(icp_finish_kernel, dim3(1), dim3(FIN_THREADS), 0, x->stream, w.st, w.live, w.live_list, w.n_lw, w.cpart, w.packet, 2,
                                   prm->estimator, want_trace ? w.trace : nullptr, w.hist, bc[0], bc[1], bc[2], (size_t)0);
            }
            PEDP_HIP_CHECK(hipGetLastError());
        } else {
        if (!degenerate) {
            rc = enqueue_nn_pass(x, w, source, target, pass == 0 ? 0 : 1, tp, r, ev0, ev1, job.exhaustive);
            if (rc) return rc;
            hipLaunchKernelGGL(icp_accumulate_kernel, dim3(ACC_BLOCKS), dim3(ACC_THREADS), 0, x->stream, w.st,
                               prm->estimator, w.P, Ns, target->pts, target->normals, w.idx, w.d2, r2, w.partials);
            if (exchange) hipLaunchKernelGGL(icp_reduce_kernel, dim3(1), dim3(64), 0, x->stream, w.st, w.partials, w.packet);
        } else {
            PEDP_HIP_CHECK(hipMemsetAsync(w.packet, 0, sizeof(double) * 32, x->stream));
            if (pass == 0 && Ns > 0) PEDP_HIP_CHECK(hipMemsetAsync(w.idx, 0xFF, sizeof(int32_t) * (size_t)Ns, x->stream));
        }
        if (prm->use_comm) {  // RCCL all-reduce on this stream, issued by the library
            rc = pedp_comm_allreduce_sum_f64(x, w.packet, PACKET);
            if (rc) { (void)hipStreamSynchronize(x->stream); return rc; }
        } else if (prm->allreduce) {
            if (prm->allreduce(prm->allreduce_user, w.packet, PACKET, (void *)x->stream) != 0) {
                pedp_set_error("pedp_icp: all-reduce hook failed in pass %d", pass);
                (void)hipStreamSynchronize(x->stream);
                return PEDP_ERR_COLLECTIVE;
            }
        }
        // A fused accumulate + solve (last workgroup done runs the solve) was measured slower:
        // the device-scope release every workgroup needs writes the whole L2 back (43 us vs 12 + 16).
        const double *fold = (!degenerate && !exchange) ? w.partials : nullptr;
        hipLaunchKernelGGL(icp_solve_kernel, dim3(1), dim3(256), 0, x->stream, w.st, w.packet, fold, pass, max_iter,
                           prm->estimator, ng, prm->relative_fitness, prm->relative_rmse, want_trace ? w.trace : nullptr);
        PEDP_HIP_CHECK(hipGetLastError());
        }
        // Passes after convergence are no-ops on the device but still cost launches; with the
        // early exit enabled, look at the flag every 8th pass (one 4-byte read-back, identical
        // on every rank of a sharded run) and stop enqueuing once it is set.
        if (early_stop && prm->relative_fitness >= 0.0 && (pass & 7) == 7 && pass < max_iter) {
            int *flag = (int *)((char *)x->pinned + 4096);
            PEDP_HIP_CHECK(hipMemcpyAsync(flag, &w.st->done, sizeof(int), hipMemcpyDeviceToHost, x->stream));
            PEDP_HIP_CHECK(hipStreamSynchronize(x->stream));
            if (*flag) break;
        }
    }
    PEDP_HIP_CHECK(hipMemcpyAsync(hp, w.st, sizeof(IcpState), hipMemcpyDeviceToHost, x->stream));
    return PEDP_OK;
}

int icp_collect(pedp_ctx_t x, const IcpJob &job, double T_out[16], double *fitness, double *inlier_rmse,
                int32_t *n_iter_done, int32_t *corr, double *trace) {
    PEDP_HIP_CHECK(hipSetDevice(x->device));
    const IcpWorkspace &w = job.w;
    if (corr && job.Ns > 0)
        { int dn_ = pedp_download(x, corr, w.idx, sizeof(int32_t) * (size_t)job.Ns); if (dn_) return dn_; }
    if (trace)
        PEDP_HIP_CHECK(hipMemcpyAsync(trace, w.trace, sizeof(double) * 18 * (size_t)(job.max_iter + 1), hipMemcpyDeviceToHost, x->stream));
    PEDP_HIP_CHECK(hipStreamSynchronize(x->stream));
    const IcpState *hp = (const IcpState *)x->pinned;
    if (hp->done < 0) {
        pedp_set_error("pedp_icp: a pass could not be closed (workgroups of its launch did not sign off in time)");
        return PEDP_ERR_HIP;
    }
    for (int k = 0; k < 16; ++k) T_out[k] = hp->T[k];
    // (scene slot, target point) pairs the MFMAs evaluated: fused pass counts 16 x 16 wave-tiles, the segmented path units x 128 slots
    x->icp_last_cand = job.w.fused ? hp->sum_tiles * 256 : hp->sum_tiles * (16 * job.qt) * (NN_SB * 16);
    x->icp_last_fb = hp->sum_fb;
    x->icp_last_passes = hp->iters + 1;
    x->icp_last_planned = hp->n_planned;
    x->icp_last_nt = job.Nt;
    if (fitness) *fitness = hp->fitness;
    if (inlier_rmse) *inlier_rmse = hp->rmse;
    if (n_iter_done) *n_iter_done = hp->iters;
    return PEDP_OK;
}

// Batched registrations on the SEGMENTED path (dense sweeps, large radii) are bound by the launch
// rate of their short kernels, so a sub-context captures the whole pass sequence of a registration
// once into a hipGraph and replays it per start pose: the graph's nodes read the start state from
// the executor's pinned block and everything else from device memory.  Anything the captured
// launches depend on is in the key.
bool graph_key_equal(const pedp_icp_graph_key &a, const pedp_icp_graph_key &b) {
    return a.src_gen == b.src_gen && a.tgt_gen == b.tgt_gen && a.ws == b.ws && a.Ns == b.Ns && a.Nt == b.Nt &&
           a.max_iter == b.max_iter && a.qt == b.qt && a.estimator == b.estimator && a.r == b.r &&
           a.rel_fitness == b.rel_fitness && a.rel_rmse == b.rel_rmse && a.seg == b.seg && a.poses == b.poses;
}

int icp_launch_replayed(pedp_ctx_t x, pedp_cloud_t source, pedp_cloud_t target, const TargetPrep &tp,
                        const pedp_icp_params *prm, const double init[16], IcpJob &job) {
    int rc = icp_job_setup(x, source, target, prm, job);
    if (rc) return rc;
    icp_fill_state((IcpState *)x->pinned, tp, init, prm, source->N);
    pedp_icp_graph_key key;
    key.src_gen = source->gen; key.tgt_gen = target->gen; key.ws = x->icp_ws.ptr;
    key.Ns = job.Ns; key.Nt = job.Nt; key.max_iter = job.max_iter; key.qt = job.qt; key.estimator = prm->estimator;
    key.r = prm->max_correspondence_distance;
    key.rel_fitness = prm->relative_fitness;
    key.rel_rmse = prm->relative_rmse;
    key.seg = job.w.fused ? -1 : 0;
    const bool same = x->icp_graph && graph_key_equal(x->icp_graph_key, key);
    if (!same) {
        if (x->icp_graph) { (void)hipGraphExecDestroy(x->icp_graph); x->icp_graph = nullptr; }
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(x->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            rc = icp_enqueue(x, source, target, tp, prm, false, false, job);
            const hipError_t e = hipStreamEndCapture(x->stream, &graph);
            if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            if (e == hipSuccess && graph && hipGraphInstantiate(&x->icp_graph, graph, nullptr, nullptr, 0) == hipSuccess)
                x->icp_graph_key = key;
            else
                x->icp_graph = nullptr;
            if (graph) (void)hipGraphDestroy(graph);
        }
        (void)hipGetLastError();
        if (!x->icp_graph)  // capture unavailable: plain launches, same work
            return icp_enqueue(x, source, target, tp, prm, false, false, job);
    }
    PEDP_HIP_CHECK(hipGraphLaunch(x->icp_graph, x->stream));
    return PEDP_OK;
}

// Batched registrations on the FUSED path share their launches: pose b of a group is blockIdx.y
// of the chunk kernel and blockIdx.x of the finish kernel, and owns one of `G` equal blocks of the
// workspace.  One captured graph holds `seg` passes (two launches each, no copies) for the whole
// group; the start states are uploaded in front of the first replay, the states are read back
// behind every replay, and the graph is replayed until every pose reports done.  Radius, criteria,
// pass counter and iteration limit live in the device state, so the same graph serves any start
// poses -- and a group that converges after 7 iterations costs 10 passes of launches, not 31.
constexpr int BATCH_GROUP_MAX = 32;

int icp_batch_fused(pedp_ctx_t c, pedp_cloud_t source, pedp_cloud_t target, const TargetPrep &tp,
                    const pedp_icp_params *prms, const double *inits, int B, double *T_out, double *fitness,
                    double *inlier_rmse, int32_t *n_iter_done) {
    bool no_exit = true;
    for (int b = 0; b < B; ++b) no_exit = no_exit && (prms[b].relative_fitness < 0.0 || prms[b].relative_rmse < 0.0);
    int G = 1;  // group size: the power of two that holds the batch, at most BATCH_GROUP_MAX (one cached graph per size)
    int slot = 0;
    while (G < B && G < BATCH_GROUP_MAX) { G <<= 1; ++slot; }
    static_assert(sizeof(IcpState) * BATCH_GROUP_MAX <= 32768, "two pinned halves hold a group's states");
    IcpJob job;
    job.max_iter = prms[0].max_iteration;
    job.Ns = source->N;
    job.Nt = target->N;
    job.qt = 1;
    IcpWorkspace &w = job.w;
    int rc = carve_workspace(c, job.Ns, job.Nt, job.max_iter, 1, w, true, G);
    if (rc) return rc;
    w.tgt4 = (const float4 *)target->tgt4;
    w.word_sph = (const float4 *)target->tile_sphw;
    w.tgt_s = (const double *)target->tgt_s;
    w.tile_sph = (const float4 *)target->tile_sph;
    w.tgt_perm = (const int32_t *)target->perm;
    w.src_perm = (const int32_t *)source->perm;
    const int all = job.max_iter + 1;
    pedp_icp_graph_key key;
    key.src_gen = source->gen; key.tgt_gen = target->gen; key.ws = c->icp_ws.ptr;
    key.Ns = job.Ns; key.Nt = job.Nt; key.qt = 1; key.estimator = prms[0].estimator;
    key.max_iter = iter_capacity(job.max_iter);  // the limit itself is in the device state; the workspace layout (trace, history) depends on its capacity class
    key.r = -1.0;
    // passes per replay: everything when no pose can stop early, else a stretch that covers the usual
    // convergence (Open3D's default criteria stop problems of this kind after 6-9 iterations)
    key.seg = no_exit ? all : (all < 10 ? all : 10);
    key.poses = G;
    slot = 3 * slot + (key.seg == 1 ? 0 : (key.seg == 2 ? 1 : 2));  // a graph per group size and kind of stretch
    // One or two passes per stretch (a z-search probe is a one-iteration registration) are launched as they are: a graph
    // of two kernels saves nothing, and with a new scene handle every frame its capture and instantiation (35-50 us) came
    // back for every frame's first batch
    const bool direct = key.seg <= 2;
    if (!direct && !(c->icp_bgraph[slot] && graph_key_equal(c->icp_bgraph_key[slot], key))) {
        if (c->icp_bgraph[slot]) { (void)hipGraphExecDestroy(c->icp_bgraph[slot]); c->icp_bgraph[slot] = nullptr; }
        hipGraph_t graph = nullptr;
        PEDP_HIP_CHECK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        for (int p = 0; p < key.seg && !rc; ++p) {
            rc = enqueue_fused_pass(c, w, source, target, prms[0].estimator, tp, nullptr, nullptr, true, nullptr, G);
            if (unfused_finish())
                hipLaunchKernelGGL(icp_finish_kernel, dim3((unsigned)G), dim3(FIN_THREADS), 0, c->stream, w.st, w.live, w.live_list,
                                   w.n_lw, w.cpart, w.packet, 0, prms[0].estimator, (double *)nullptr, w.hist,
                                   0.5 * (tp.lo[0] + tp.hi[0]), 0.5 * (tp.lo[1] + tp.hi[1]), 0.5 * (tp.lo[2] + tp.hi[2]),
                                   G > 1 ? w.pose_stride : (size_t)0);
        }
        const hipError_t e = hipStreamEndCapture(c->stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess || !graph || hipGraphInstantiate(&c->icp_bgraph[slot], graph, nullptr, nullptr, 0) != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            c->icp_bgraph[slot] = nullptr;
            pedp_set_error("pedp_icp_batched: graph capture failed: %s", hipGetErrorString(e));
            return PEDP_ERR_HIP;
        }
        (void)hipGraphDestroy(graph);
        c->icp_bgraph_key[slot] = key;
    }
    IcpState *up = (IcpState *)c->pinned, *down = (IcpState *)((char *)c->pinned + 32768);
    c->icp_last_cand = c->icp_last_fb = c->icp_last_passes = 0;  // statistics: totals over the batch
    c->icp_last_nt = target->N;
    for (int b0 = 0; b0 < B; b0 += G) {
        const int n = B - b0 < G ? B - b0 : G;
        for (int k = 0; k < G; ++k) {
            if (k < n) {
                icp_fill_state(&up[k], tp, inits + 16 * (b0 + k), &prms[b0 + k], source->N);
            } else {
                up[k] = IcpState{};
                up[k].done = 1;  // an empty seat of the group: every kernel returns at once
            }
        }
        // start states (each carries its start transformation: slot 0 of the history); tickets and live masks at zero
        static_assert(sizeof(IcpState) % 8 == 0, "states move as 8-byte words");
        const size_t zero_bytes = (size_t)((char *)w.live - (char *)w.ticket) + sizeof(unsigned long long) * 2 * (size_t)w.n_lw;
        hipLaunchKernelGGL(batch_state_scatter_kernel, dim3((unsigned)G), dim3(256), 0, c->stream, (const unsigned long long *)up, (char *)w.st,
                           w.pose_stride, (int)(sizeof(IcpState) / 8), (char *)w.ticket, (int)(zero_bytes / 8));
        bool finished = false;
        for (int guard = 0; guard < (1 << 20) && !finished; ++guard) {
            if (direct) {
                for (int p = 0; p < key.seg; ++p) {
                    int rp = enqueue_fused_pass(c, w, source, target, prms[0].estimator, tp, nullptr, nullptr, true, nullptr, G);
                    if (rp) return rp;
                    if (unfused_finish())
                        hipLaunchKernelGGL(icp_finish_kernel, dim3((unsigned)G), dim3(FIN_THREADS), 0, c->stream, w.st, w.live, w.live_list,
                                           w.n_lw, w.cpart, w.packet, 0, prms[0].estimator, (double *)nullptr, w.hist,
                                           0.5 * (tp.lo[0] + tp.hi[0]), 0.5 * (tp.lo[1] + tp.hi[1]), 0.5 * (tp.lo[2] + tp.hi[2]),
                                           G > 1 ? w.pose_stride : (size_t)0);
                }
            } else
            PEDP_HIP_CHECK(hipGraphLaunch(c->icp_bgraph[slot], c->stream));
            hipLaunchKernelGGL(batch_state_gather_kernel, dim3((unsigned)G), dim3(256), 0, c->stream, (const char *)w.st, w.pose_stride,
                               (int)(sizeof(IcpState) / 8), (unsigned long long *)down);
            PEDP_HIP_CHECK(hipGetLastError());
            PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
            finished = true;
            for (int k = 0; k < n; ++k) finished = finished && down[k].done;
        }
        PEDP_REQUIRE(finished, "pedp_icp_batched: a registration did not finish");
        for (int k = 0; k < n; ++k)
            if (down[k].done < 0) {
                pedp_set_error("pedp_icp_batched: a pass of registration %d could not be closed (workgroups of its launch did not sign off in time)", b0 + k);
                return PEDP_ERR_HIP;
            }
        for (int k = 0; k < n; ++k) {
            const IcpState &h = down[k];
            for (int q = 0; q < 16; ++q) T_out[16 * (b0 + k) + q] = h.T[q];
            if (fitness) fitness[b0 + k] = h.fitness;
            if (inlier_rmse) inlier_rmse[b0 + k] = h.rmse;
            if (n_iter_done) n_iter_done[b0 + k] = h.iters;
            c->icp_last_cand += h.sum_tiles * 256;
            c->icp_last_fb += h.sum_fb;
            c->icp_last_passes += h.iters + 1;
        }
    }
    return PEDP_OK;
}

}  // namespace

void pedp_icp_drop_pending(pedp_ctx_t c) {
    if (c && c->icp_pending) {
        delete (IcpJob *)c->icp_pending;
        c->icp_pending = nullptr;
    }
}

extern "C" {

int pedp_icp(pedp_ctx_t c, pedp_cloud_t source, pedp_cloud_t target, const pedp_icp_params *prm,
             const double init[16], double T_out[16], double *fitness, double *inlier_rmse,
             int32_t *n_iter_done, int32_t *corr, double *trace) {
    PEDP_REQUIRE(c && source && target && prm && init && T_out, "pedp_icp: null argument");
    PEDP_REQUIRE(!c->icp_pending, "pedp_icp: a registration is pending on this context (pedp_icp_end first)");
    int rc = icp_check_args(c, source, target, prm);
    if (rc) return rc;
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    TargetPrep tp;
    rc = icp_prepare(c, source, target, tp);
    if (rc) return rc;
    IcpJob job;
    rc = icp_job_setup(c, source, target, prm, job);
    if (rc) return rc;
    icp_fill_state((IcpState *)c->pinned, tp, init, prm, source->N);
    rc = icp_enqueue(c, source, target, tp, prm, trace != nullptr, true, job);
    if (rc) return rc;
    return icp_collect(c, job, T_out, fitness, inlier_rmse, n_iter_done, corr, trace);
}

int pedp_icp_begin(pedp_ctx_t c, pedp_cloud_t source, pedp_cloud_t target, const pedp_icp_params *prm, const double init[16],
                   int want_trace) {
    PEDP_REQUIRE(c && source && target && prm && init, "pedp_icp_begin: null argument");
    PEDP_REQUIRE(!c->icp_pending, "pedp_icp_begin: a registration is pending on this context (pedp_icp_end first)");
    int rc = icp_check_args(c, source, target, prm);
    if (rc) return rc;
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    TargetPrep tp;
    rc = icp_prepare(c, source, target, tp);
    if (rc) return rc;
    IcpJob *job = new (std::nothrow) IcpJob;
    PEDP_REQUIRE(job, "pedp_icp_begin: out of memory");
    rc = icp_job_setup(c, source, target, prm, *job);
    if (!rc) {
        icp_fill_state((IcpState *)c->pinned, tp, init, prm, source->N);
        rc = icp_enqueue(c, source, target, tp, prm, want_trace != 0, false, *job);
    }
    if (rc) { delete job; return rc; }
    c->icp_pending = job;
    return PEDP_OK;
}

int pedp_icp_end(pedp_ctx_t c, double T_out[16], double *fitness, double *inlier_rmse, int32_t *n_iter_done, int32_t *corr,
                 double *trace) {
    PEDP_REQUIRE(c && T_out, "pedp_icp_end: null argument");
    PEDP_REQUIRE(c->icp_pending, "pedp_icp_end: no registration is pending on this context");
    IcpJob *job = (IcpJob *)c->icp_pending;
    c->icp_pending = nullptr;
    const int rc = icp_collect(c, *job, T_out, fitness, inlier_rmse, n_iter_done, corr, trace);
    delete job;
    return rc;
}

// Hypotheses are independent: up to PEDP_MAX_SUB registrations are in flight at once, each on
// its own stream and workspace (sub-contexts of c), sharing the clouds' cached preparation.
int pedp_icp_batched_ex(pedp_ctx_t c, pedp_cloud_t source, pedp_cloud_t target, const pedp_icp_params *prms,
                        const double *inits, int B, double *T_out, double *fitness, double *inlier_rmse, int32_t *n_iter_done) {
    PEDP_REQUIRE(c && source && target && prms && inits && T_out, "pedp_icp_batched: null argument");
    PEDP_REQUIRE(!c->icp_pending, "pedp_icp_batched: a registration is pending on this context (pedp_icp_end first)");
    PEDP_REQUIRE(B >= 0, "pedp_icp_batched: negative batch");
    if (B == 0) return PEDP_OK;
    bool uniform = true;
    double r_max = prms[0].max_correspondence_distance;
    for (int b = 0; b < B; ++b) {
        const pedp_icp_params &q = prms[b];
        PEDP_REQUIRE(!q.allreduce && !q.use_comm, "pedp_icp_batched: hypotheses shard across ranks, not within one registration");
        PEDP_REQUIRE(q.estimator == prms[0].estimator && q.max_iteration == prms[0].max_iteration,
                     "pedp_icp_batched: estimator and max_iteration must be the same for the whole batch");
        uniform = uniform && q.max_correspondence_distance == prms[0].max_correspondence_distance &&
                  q.relative_fitness == prms[0].relative_fitness && q.relative_rmse == prms[0].relative_rmse;
        if (q.max_correspondence_distance > r_max) r_max = q.max_correspondence_distance;
    }
    int rc = icp_check_args(c, source, target, &prms[0]);
    if (rc) return rc;
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    TargetPrep tp;
    rc = icp_prepare(c, source, target, tp);
    if (rc) return rc;
    // The fused path takes a whole group of poses per launch, whatever their radii and criteria (they
    // live in the device state).  The segmented path replays one graph per pose on sub-contexts and
    // needs one radius and one set of criteria for that; a mixed batch that is not fused-eligible
    // runs one by one on the owner's stream -- same results.
    bool all_fused = !c->icp_exhaustive;
    for (int b = 0; b < B && all_fused; ++b) {
        const double r = prms[b].max_correspondence_distance;
        all_fused = icp_unit_size(target, r) == 1 && r > 0.0 && source->N > 0 && target->N > 0 &&
                    (target->N + 1023) / 1024 <= BK_WCAP;
    }
    if (all_fused) return icp_batch_fused(c, source, target, tp, prms, inits, B, T_out, fitness, inlier_rmse, n_iter_done);
    if (!uniform) {
        {
            for (int b = 0; b < B; ++b) {
                IcpJob job;
                rc = icp_job_setup(c, source, target, &prms[b], job);
                if (rc) return rc;
                icp_fill_state((IcpState *)c->pinned, tp, inits + 16 * b, &prms[b], source->N);
                rc = icp_enqueue(c, source, target, tp, &prms[b], false, true, job);
                if (rc) return rc;
                rc = icp_collect(c, job, T_out + 16 * b, fitness ? fitness + b : nullptr, inlier_rmse ? inlier_rmse + b : nullptr,
                                 n_iter_done ? n_iter_done + b : nullptr, nullptr, nullptr);
                if (rc) return rc;
            }
            return PEDP_OK;
        }
    }
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));  // preparation is visible to the sub-streams
    const int K = B < PEDP_MAX_SUB ? B : PEDP_MAX_SUB;
    for (int k = 0; k < K; ++k) {
        if (!c->sub[k]) {
            rc = pedp_ctx_create(c->device, nullptr, &c->sub[k]);
            if (rc) return rc;
        }
        c->sub[k]->icp_exhaustive = c->icp_exhaustive;
    }
    IcpJob jobs[PEDP_MAX_SUB];
    c->icp_last_cand = c->icp_last_fb = c->icp_last_passes = 0;  // statistics: totals over the batch
    c->icp_last_nt = target->N;
    // on an error return no sub-stream may still be running on the clouds' buffers (the caller
    // is free to destroy them): drain them all first
    auto drain = [&](int rc_) {
        for (int k = 0; k < K; ++k)
            if (c->sub[k]) (void)hipStreamSynchronize(c->sub[k]->stream);
        return rc_;
    };
    for (int b0 = 0; b0 < B; b0 += K) {
        const int n = (B - b0 < K) ? B - b0 : K;
        for (int k = 0; k < n; ++k) {
            rc = icp_launch_replayed(c->sub[k], source, target, tp, &prms[b0 + k], inits + 16 * (b0 + k), jobs[k]);
            if (rc) return drain(rc);
        }
        for (int k = 0; k < n; ++k) {
            rc = icp_collect(c->sub[k], jobs[k], T_out + 16 * (b0 + k), fitness ? fitness + b0 + k : nullptr,
                             inlier_rmse ? inlier_rmse + b0 + k : nullptr, n_iter_done ? n_iter_done + b0 + k : nullptr,
                             nullptr, nullptr);
            if (rc) return drain(rc);
            c->icp_last_cand += c->sub[k]->icp_last_cand;
            c->icp_last_fb += c->sub[k]->icp_last_fb;
            c->icp_last_passes += c->sub[k]->icp_last_passes;
        }
    }
    return PEDP_OK;
}

int pedp_icp_batched(pedp_ctx_t c, pedp_cloud_t source, pedp_cloud_t target, const pedp_icp_params *prm,
                     const double *inits, int B, double *T_out, double *fitness, double *inlier_rmse) {
    PEDP_REQUIRE(prm, "pedp_icp_batched: null argument");
    PEDP_REQUIRE(B >= 0 && B <= (1 << 20), "pedp_icp_batched: batch size out of range");
    pedp_icp_params *all = (pedp_icp_params *)malloc(sizeof(pedp_icp_params) * (size_t)(B > 0 ? B : 1));
    if (!all) { pedp_set_error("pedp_icp_batched: out of host memory"); return PEDP_ERR_ALLOC; }
    for (int b = 0; b < B; ++b) {
        all[b] = *prm;
        all[b].relative_fitness = -1.0;  // no early exit across the batch
        all[b].relative_rmse = -1.0;
    }
    const int rc = pedp_icp_batched_ex(c, source, target, all, inits, B, T_out, fitness, inlier_rmse, nullptr);
    free(all);
    return rc;
}

int pedp_nn(pedp_ctx_t c, pedp_cloud_t source, pedp_cloud_t target, const double T[16], int32_t *idx, double *d2) {
    PEDP_REQUIRE(c && source && target && T && idx && d2, "pedp_nn: null argument");
    PEDP_REQUIRE(source->ctx == c && target->ctx == c, "pedp_nn: cloud belongs to another context");
    PEDP_REQUIRE(target->N > 0, "pedp_nn: empty target");
    if (source->N == 0) return PEDP_OK;
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    IcpWorkspace w;
    int rc = carve_workspace(c, source->N, target->N, 0, 4, w);  // no radius: dense sweep, 64-row units
    if (rc) return rc;
    TargetPrep tp;
    rc = ensure_target_pack(c, target, tp);
    if (!rc) rc = ensure_target_bf16(c, target);
    if (rc) return rc;
    w.tgt4 = (const float4 *)target->tgt4;
    w.tile_sph = (const float4 *)target->tile_sph4;
    w.tgt_perm = (const int32_t *)target->perm;
    rc = ensure_spatial_perm(c, source);
    if (rc) return rc;
    w.src_perm = (const int32_t *)source->perm;
    IcpState *hp = (IcpState *)c->pinned;
    IcpState h{};
    for (int k = 0; k < 16; ++k) { h.T[k] = T[k]; h.upd[k] = T[k]; }
    for (int k = 0; k < 3; ++k) h.centroid[k] = tp.c[k];
    *hp = h;
    PEDP_HIP_CHECK(hipMemcpyAsync(w.st, hp, sizeof(IcpState), hipMemcpyHostToDevice, c->stream));
    // every point is a candidate: radius = "infinite" (cloud scale bound)
    c->nn_pairs = 0;
    rc = enqueue_nn_pass(c, w, source, target, 0, tp, 1e18, c->nn_ev0, c->nn_ev1);
    if (rc) return rc;
    { int dn_ = pedp_download(c, idx, w.idx, sizeof(int32_t) * (size_t)source->N); if (dn_) return dn_; }
    { int dn_ = pedp_download(c, d2, w.d2, sizeof(double) * (size_t)source->N); if (dn_) return dn_; }
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PEDP_OK;
}

int pedp_ransac_hypotheses(pedp_ctx_t c, pedp_cloud_t src, pedp_cloud_t tgt, const int32_t *corr, uint64_t seed, int64_t itr0,
                           int count, double edge_similarity, double max_distance, double normal_angle, uint8_t *accepted,
                           double *T) {
    PEDP_REQUIRE(c && src && tgt, "pedp_ransac_hypotheses: null context / cloud");
    PEDP_REQUIRE(src->ctx == c && tgt->ctx == c, "pedp_ransac_hypotheses: clouds belong to another context");
    PEDP_REQUIRE(count >= 0 && count <= (1 << 22) && itr0 >= 0, "pedp_ransac_hypotheses: count must be in 0..2^22");
    if (count == 0) return PEDP_OK;
    PEDP_REQUIRE(corr && accepted && T, "pedp_ransac_hypotheses: null arrays");
    PEDP_REQUIRE(src->N > 0 && tgt->N > 0, "pedp_ransac_hypotheses: empty cloud");
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    const size_t sz_corr = align_up(sizeof(int32_t) * (size_t)src->N, 256), sz_flag = align_up((size_t)count, 256);
    int st = c->ops.reserve(sz_corr + sz_flag + sizeof(double) * 16 * (size_t)count + 256);
    if (st) return st;
    int32_t *d_corr = (int32_t *)c->ops.ptr;
    unsigned char *d_flag = (unsigned char *)c->ops.ptr + sz_corr;
    double *d_T = (double *)((char *)c->ops.ptr + sz_corr + sz_flag);
    // correspondences must point inside the target (checked on the host: they index device memory)
    for (int64_t i = 0; i < src->N; ++i)
        PEDP_REQUIRE(corr[i] >= 0 && corr[i] < tgt->N, "pedp_ransac_hypotheses: correspondence %lld -> %d outside the target",
                     (long long)i, (int)corr[i]);
    { int up_ = pedp_upload(c, d_corr, corr, sizeof(int32_t) * (size_t)src->N); if (up_) return up_; }
    hipLaunchKernelGGL(ransac_hypothesis_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, c->stream,
                       (unsigned long long)seed, (long long)itr0, count, (const double *)src->pts, (const double *)src->normals,
                       (long long)src->N, (const double *)tgt->pts, (const double *)tgt->normals, (const int32_t *)d_corr,
                       edge_similarity, max_distance, cos(normal_angle), d_flag, d_T);
    PEDP_HIP_CHECK(hipGetLastError());
    { int dn_ = pedp_download(c, accepted, d_flag, (size_t)count); if (dn_) return dn_; }
    { int dn_ = pedp_download(c, T, d_T, sizeof(double) * 16 * (size_t)count); if (dn_) return dn_; }
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PEDP_OK;
}

int pedp_icp_configure(pedp_ctx_t c, int exhaustive, int timed_pass) {
    PEDP_REQUIRE(c, "pedp_icp_configure: null context");
    c->icp_exhaustive = exhaustive != 0;
    c->icp_timed_pass = timed_pass;
    for (int k = 0; k < PEDP_MAX_SUB; ++k)  // captured graphs bake the mode in
        if (c->sub[k] && c->sub[k]->icp_graph) { (void)hipGraphExecDestroy(c->sub[k]->icp_graph); c->sub[k]->icp_graph = nullptr; }
    for (hipGraphExec_t &g : c->icp_bgraph)
        if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
    return PEDP_OK;
}

int pedp_icp_last_stats(pedp_ctx_t c, int64_t *passes, int64_t *pairs_swept, int64_t *fallback_points) {
    PEDP_REQUIRE(c, "pedp_icp_last_stats: null context");
    if (passes) *passes = c->icp_last_passes;
    if (pairs_swept) *pairs_swept = c->icp_last_cand;
    if (fallback_points) *fallback_points = c->icp_last_fb;
    return PEDP_OK;
}

int pedp_debug_nn_bf16(pedp_ctx_t c, const float *src4, int64_t n_src, const float *tgt4, int64_t n_tgt, float *g) {
    PEDP_REQUIRE(c && src4 && tgt4 && g && n_src > 0 && n_tgt > 0 && n_src % 16 == 0 && n_tgt % 16 == 0 && n_src * n_tgt <= ((int64_t)1 << 26),
                 "pedp_debug_nn_bf16: counts must be positive multiples of 16, at most 2^26 pairs");
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    float4 *d_s = nullptr, *d_t = nullptr;
    uint4 *d_b = nullptr;
    float *d_g = nullptr;
    hipError_t e = hipMalloc((void **)&d_s, sizeof(float4) * (size_t)n_src);
    if (e == hipSuccess) e = hipMalloc((void **)&d_t, sizeof(float4) * (size_t)n_tgt);
    if (e == hipSuccess) e = hipMalloc((void **)&d_b, sizeof(uint4) * 4 * (size_t)n_tgt);
    if (e == hipSuccess) e = hipMalloc((void **)&d_g, sizeof(float) * (size_t)(n_src * n_tgt));
    if (e == hipSuccess) e = hipMemcpyAsync(d_s, src4, sizeof(float4) * (size_t)n_src, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_t, tgt4, sizeof(float4) * (size_t)n_tgt, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pack_target_bf16_kernel, dim3((unsigned)((n_tgt * 4 + 255) / 256)), dim3(256), 0, c->stream, (const float4 *)d_t, n_tgt, d_b);
        hipLaunchKernelGGL(nn_bf16_debug_kernel, dim3((unsigned)((n_src / 16) * (n_tgt / 16))), dim3(64), 0, c->stream, (const uint4 *)d_b,
                           (const float4 *)d_s, (int)(n_src / 16), (int)n_tgt, d_g);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(g, d_g, sizeof(float) * (size_t)(n_src * n_tgt), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_s); (void)hipFree(d_t); (void)hipFree(d_b); (void)hipFree(d_g);
    PEDP_HIP_CHECK(e);
    return PEDP_OK;
}

int pedp_icp_last_planned_passes(pedp_ctx_t c, int64_t *planned) {
    PEDP_REQUIRE(c && planned, "pedp_icp_last_planned_passes: null argument");
    *planned = c->icp_last_planned;
    return PEDP_OK;
}

int pedp_nn_last_sweep_ms(pedp_ctx_t c, float *ms) {
    PEDP_REQUIRE(c && ms, "pedp_nn_last_sweep_ms: null argument");
    PEDP_REQUIRE(c->nn_timed, "pedp_nn_last_sweep_ms: no timed sweep has run on this context");
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    if (c->nn_pairs > 0) {  // sampled passes: the mean
        float sum = 0.f;
        for (int k = 0; k < c->nn_pairs; ++k) {
            float one = 0.f;
            PEDP_HIP_CHECK(hipEventSynchronize(c->nn_evs[2 * k + 1]));
            PEDP_HIP_CHECK(hipEventElapsedTime(&one, c->nn_evs[2 * k], c->nn_evs[2 * k + 1]));
            sum += one;
        }
        *ms = sum / (float)c->nn_pairs;
        return PEDP_OK;
    }
    PEDP_HIP_CHECK(hipEventSynchronize(c->nn_ev1));
    PEDP_HIP_CHECK(hipEventElapsedTime(ms, c->nn_ev0, c->nn_ev1));
    if (c->nn_span_launches > 1) *ms /= (float)c->nn_span_launches;
    return PEDP_OK;
}

}  // extern "C"
