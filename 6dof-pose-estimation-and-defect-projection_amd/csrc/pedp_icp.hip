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
#include <cmath>
#include <cstdlib>
#include <new>

#ifndef PEDP_NN_EXPERIMENT
#define PEDP_NN_EXPERIMENT 0
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
constexpr int SORT_BITS = 5;          // spatial sort: 32^3 Hilbert-ordered cells over the cloud's bounding box
constexpr int SORT_CELLS = 1 << (3 * SORT_BITS);
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
// Counting sort by the Hilbert-curve index of the point's cell in a 32^3 grid over the bounding box:
// consecutive entries of `perm` are neighbours in space, so 128-point scene blocks and 16-point
// target tiles are compact and their bounding spheres are small.  Order inside a cell is
// arbitrary (atomics); nothing downstream depends on it.
// 3-D Hilbert index of cell (x, y, z), SORT_BITS bits per axis (Skilling, "Programming the
// Hilbert curve", 2004: axes -> transpose, then bit interleave).  Unlike Morton order, points
// that are consecutive along the curve are always neighbours in space, so no 128-point scene
// block or 16-point target tile straddles a long jump (such blocks would defeat the culling).
__device__ __forceinline__ unsigned hilbert3(unsigned x, unsigned y, unsigned z) {
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
    unsigned h = 0;
#pragma unroll
    for (int b = SORT_BITS - 1; b >= 0; --b)
        h = (h << 3) | (((X[0] >> b) & 1u) << 2) | (((X[1] >> b) & 1u) << 1) | ((X[2] >> b) & 1u);
    return h;
}
// Cell of point i; points outside the region [lo, lo + extent) all share the extra bucket
// SORT_CELLS (they are no ICP candidates; one bucket keeps them off the border cells).
__device__ __forceinline__ unsigned point_cell(const double *__restrict__ pts, int64_t i, double lox, double loy,
                                               double loz, double sx, double sy, double sz) {
    const double fx = (pts[3 * i] - lox) * sx, fy = (pts[3 * i + 1] - loy) * sy, fz = (pts[3 * i + 2] - loz) * sz;
    const double top = (double)(1 << SORT_BITS);
    if (!(fx >= 0.0 && fx < top && fy >= 0.0 && fy < top && fz >= 0.0 && fz < top)) return (unsigned)SORT_CELLS;
    return hilbert3((unsigned)(int)fx, (unsigned)(int)fy, (unsigned)(int)fz);
}
// one atomic per wave for the (possibly huge) outside bucket, per-lane atomics for real cells
__device__ __forceinline__ unsigned cell_add(unsigned *__restrict__ ctr, unsigned cell, bool active) {
    const bool outside = active && cell == (unsigned)SORT_CELLS;
    const unsigned long long om = __builtin_amdgcn_ballot_w64(outside);
    unsigned res = 0;
    if (om != 0ull) {
        const int lane = threadIdx.x & 63, leader = __builtin_ctzll(om);
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(&ctr[SORT_CELLS], (unsigned)__builtin_popcountll(om));
        base = __shfl(base, leader, 64);
        if (outside) res = base + (unsigned)__builtin_popcountll(om & ((1ull << lane) - 1ull));
    }
    if (active && !outside) res = atomicAdd(&ctr[cell], 1u);
    return res;
}
__global__ void cell_count_kernel(const double *__restrict__ pts, int64_t N, double lox, double loy, double loz,
                                  double sx, double sy, double sz, unsigned *__restrict__ hist) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < N;
    (void)cell_add(hist, active ? point_cell(pts, i, lox, loy, loz, sx, sy, sz) : 0u, active);
}
__global__ __launch_bounds__(1024) void cell_scan_kernel(unsigned *__restrict__ hist) {
    __shared__ unsigned part[1024];
    constexpr int PER = SORT_CELLS / 1024;
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
    if (threadIdx.x == 1023) hist[SORT_CELLS] = run;  // the outside bucket follows all real cells
}
__global__ void cell_scatter_kernel(const double *__restrict__ pts, int64_t N, double lox, double loy, double loz,
                                    double sx, double sy, double sz, unsigned *__restrict__ cursor,
                                    int32_t *__restrict__ perm) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < N;
    const unsigned at = cell_add(cursor, active ? point_cell(pts, i, lox, loy, loz, sx, sy, sz) : 0u, active);
    if (active) perm[at] = (int32_t)i;
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
                                           float *__restrict__ S, int32_t *__restrict__ list, float Tn, float T2, float r1) {
    B[slot] = make_float4(-2.0f * p.sx, -2.0f * p.sy, -2.0f * p.sz, 1.0f);
    // Error bound of the fp32 surrogate relative to the float64 distance, for points whose
    // nearest neighbour is closer than r1 (see DESIGN.md "NN filter bound"):
    //   eps = 2^-23 * (5 * (2*|s'|_1*Tn + T2) + 2*min(r1, |s'|_1 + Tn)*(Tn + |s'|_1))
    float s1 = fabsf(p.sx) + fabsf(p.sy) + fabsf(p.sz);
    float Mi = 2.0f * s1 * Tn + T2;
    eps[slot] = 1.1920929e-7f * (5.0f * Mi + 2.0f * fminf(r1, s1 + Tn) * (Tn + s1)) * 1.0001f;
    S[slot] = p.sx * p.sx + p.sy * p.sy + p.sz * p.sz;
    list[slot] = p.i;
}
__global__ __launch_bounds__(256) void icp_transform_pack_kernel(
    IcpState *__restrict__ st, int mode, const double *__restrict__ src, double *__restrict__ P,
    const int32_t *__restrict__ perm, int64_t N, float4 *__restrict__ B, float *__restrict__ eps, float *__restrict__ S,
    int32_t *__restrict__ list, float4 *__restrict__ blk_sph, int32_t *__restrict__ idx_out,
    double *__restrict__ d2_out, float Tn, float T2, float r1, double r2cut, double lox, double loy, double loz,
    double hix, double hiy, double hiz) {
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
    if (p0.cand) pack_store(p0, base + __builtin_popcountll(m0 & lt), B, eps, S, list, Tn, T2, r1);
    if (p1.cand) pack_store(p1, base + c0 + __builtin_popcountll(m1 & lt), B, eps, S, list, Tn, T2, r1);
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

// ---- 3. sweep: one wave per segment ----
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

    // ---- expand ranks [r0, r0 + n_s) of the block's mask into the LDS tile list
    unsigned *mine = surv[wv];
    {
        int running = 0;
        const unsigned long long *mw = mask + (size_t)blk * n_words;
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
    }
    // pad units behind the list (rows that can never win): the last group is filled up with
    // them and the prefetch of the trip after it reads them
    if (lane < 2 * G) mine[n_s + lane] = (unsigned)n_tiles;  // first pad unit (|t|^2 = 1e30)
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes have landed

    float b[NN_SB];
#pragma unroll
    for (int sb = 0; sb < NN_SB; ++sb) b[sb] = srcf[(base + sb * 16) * 4 + frag];
    float b1[NN_SB], b2[NN_SB], vq[NN_SB];
    int t1[NN_SB];
#pragma unroll
    for (int sb = 0; sb < NN_SB; ++sb) { b1[sb] = __uint_as_float(0x7F800000u); b2[sb] = b1[sb]; vq[sb] = b1[sb]; t1[sb] = n_tiles; }

    if (n_s > 0) {
        // Per unit (QT MFMA tiles = 16 QT target rows): the values a lane sees are folded with
        // two v_min3 per MFMA, and only once per unit the running (best value, unit, second-best
        // value) is updated.  The matrix pipe works on the next tile while the VALU folds this one
        // (software pipeline), and the A operands of a whole group of G units (G QT tiles) are
        // fetched one group ahead: G QT x NN_SB MFMAs cover the load latency.
        constexpr int U = G * QT;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
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
    const int q = lane >> 4, j = lane & 15;
#pragma unroll
    for (int sb = 0; sb < NN_SB; ++sb) {
        const size_t o = ((size_t)seg * 4 + q) * (NN_SB * 16) + (size_t)(sb * 16 + j);
        tr_b1[o] = b1[sb];
        tr_t1[o] = t1[sb];
        tr_b2[o] = b2[sb];
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
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

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
    double ca = cos(x[0]), sa = sin(x[0]);
    double cb = cos(x[1]), sb = sin(x[1]);
    double cc = cos(x[2]), sc = sin(x[2]);
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
};

int carve_workspace(pedp_ctx_t c, int64_t Ns, int64_t Nt, int max_iter, int qt, IcpWorkspace &w) {
    w.qt = qt;
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
    PEDP_REQUIRE(ms < (int64_t)1 << 22, "pedp_icp: problem too large for the segment table (%lld x %lld points)",
                 (long long)Ns, (long long)Nt);
    w.max_segs = (int)ms;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    size_t o_st = take(sizeof(IcpState));
    size_t o_P = take(sizeof(double) * 3 * (size_t)w.Ns_pad);
    size_t o_d2 = take(sizeof(double) * (size_t)w.Ns_pad);
    size_t o_part = take(sizeof(double) * ACC_BLOCKS * PACKET);
    size_t o_pack = take(sizeof(double) * 32);
    size_t o_trace = take(sizeof(double) * 18 * (size_t)(max_iter + 1));
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
    int st = c->icp_ws.reserve(off);
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
    // the per-block survivor counters start at zero (the segment kernel re-zeroes them per pass)
    PEDP_HIP_CHECK(hipMemsetAsync(w.blk_cnt, 0, sizeof(int32_t) * (size_t)w.blocks_cap, c->stream));
    return PEDP_OK;
}

// Enqueue one correspondence pass (transform, sweep, fallback).  mode as in
// icp_transform_pack_kernel.
int enqueue_nn_pass(pedp_ctx_t c, const IcpWorkspace &w, pedp_cloud_t src, pedp_cloud_t tgt, int mode,
                    const TargetPrep &tp, double r, bool timed, bool exhaustive = false) {
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
    {
        int64_t grid = (Ns + 511) / 512;  // 128 points per wave, 4 waves per workgroup
        hipLaunchKernelGGL(icp_transform_pack_kernel, dim3((unsigned)grid), dim3(256), 0, c->stream, w.st, mode,
                           src->pts, w.P, w.src_perm, Ns, w.B, w.eps, w.S, w.list, w.blk_sph, w.idx, w.d2, tp.Tn, tp.T2,
                           r1, r2cut, tp.lo[0], tp.lo[1], tp.lo[2], tp.hi[0], tp.hi[1], tp.hi[2]);
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
        if (timed) PEDP_HIP_CHECK(hipEventRecord(c->nn_ev0, c->stream));                                              \
        hipLaunchKernelGGL((nn_sweep_kernel<QTV, GV>), dim3(sweep_grid), dim3(NN_WAVES * 64), 0, c->stream, w.st,            \
                           (const float *)w.tgt4, n_tiles, w.n_words, w.mask, w.seg_blk, w.seg_rank0, w.seg_n,         \
                           (const float *)w.B, w.tr_b1, w.tr_t1, w.tr_b2);                                            \
        if (timed) { PEDP_HIP_CHECK(hipEventRecord(c->nn_ev1, c->stream)); c->nn_timed = true; }                       \
        hipLaunchKernelGGL(nn_select_kernel<QTV>, dim3(sel_grid), dim3(256), 0, c->stream, w.st, w.blk_segstart,       \
                           w.tr_b1, w.tr_t1, w.tr_b2, tgt->pts, w.tgt_perm, Nt, w.P, w.eps, w.S, w.list, r2f, w.idx,   \
                           w.d2, w.fb);                                                                               \
        hipLaunchKernelGGL(nn_fallback_kernel<QTV>, dim3(fb_grid), dim3(FB_WAVES * 64), 0, c->stream, w.st, w.fb, w.list,     \
                           w.mask, w.n_words, w.tile_sph, r_search, tgt->pts, w.tgt_perm, Nt, w.P, w.idx, w.d2);       \
    } while (0)
    if (w.qt == 4) PEDP_NN_STAGE(4, 2);
    else PEDP_NN_STAGE(1, 4);
#undef PEDP_NN_STAGE
    PEDP_HIP_CHECK(hipGetLastError());
    return PEDP_OK;
}

// Spatial order of a cloud (device counting sort), cached in the handle.  The 32^3 cells are
// laid over `roi` (lo xyz, hi xyz), not over the whole cloud: for a scene cloud that is the
// region the target can occupy (its bounding box + radius, moved into the scene frame with
// the inverse of the first registration's init), so the resolution goes where candidate
// points are; points outside are clamped to the border cells (they are no candidates).  The
// order is rebuilt if a later call's region has moved by more than half its size.
int ensure_spatial_perm(pedp_ctx_t c, pedp_cloud_t cl, const double *roi) {
    if (cl->N == 0) return PEDP_OK;
    double lo[3], hi[3];
    for (int k = 0; k < 3; ++k) { lo[k] = roi ? roi[k] : cl->lo[k]; hi[k] = roi ? roi[3 + k] : cl->hi[k]; }
    if (cl->perm) {
        bool moved = false;
        for (int k = 0; k < 3; ++k) {
            const double ext = cl->perm_hi[k] - cl->perm_lo[k];
            if (std::fabs(0.5 * (lo[k] + hi[k]) - 0.5 * (cl->perm_lo[k] + cl->perm_hi[k])) > 0.5 * ext + 1e-12) moved = true;
        }
        if (!moved) return PEDP_OK;
    } else {
        PEDP_HIP_CHECK(hipMalloc(&cl->perm, sizeof(int32_t) * (size_t)cl->N));
        for (int k = 0; k < 3; ++k) { cl->perm_lo[k] = 0.0; cl->perm_hi[k] = -1.0; }  // no region yet: every request rebuilds
    }
    struct HistGuard {  // freed on every exit path
        unsigned *p = nullptr;
        ~HistGuard() { if (p) (void)hipFree(p); }
    } guard;
    PEDP_HIP_CHECK(hipMalloc((void **)&guard.p, sizeof(unsigned) * (SORT_CELLS + 1)));
    unsigned *hist = guard.p;
    PEDP_HIP_CHECK(hipMemsetAsync(hist, 0, sizeof(unsigned) * (SORT_CELLS + 1), c->stream));
    double sc[3];
    for (int k = 0; k < 3; ++k) {
        double ext = hi[k] - lo[k];
        sc[k] = ext > 0.0 ? (double)(1 << SORT_BITS) / ext * (1.0 - 1e-9) : 0.0;
    }
    const unsigned grid = (unsigned)((cl->N + 255) / 256);
    hipLaunchKernelGGL(cell_count_kernel, dim3(grid), dim3(256), 0, c->stream, cl->pts, cl->N, lo[0], lo[1], lo[2],
                       sc[0], sc[1], sc[2], hist);
    hipLaunchKernelGGL(cell_scan_kernel, dim3(1), dim3(1024), 0, c->stream, hist);
    hipLaunchKernelGGL(cell_scatter_kernel, dim3(grid), dim3(256), 0, c->stream, cl->pts, cl->N, lo[0], lo[1], lo[2],
                       sc[0], sc[1], sc[2], hist, (int32_t *)cl->perm);
    PEDP_HIP_CHECK(hipGetLastError());
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int k = 0; k < 3; ++k) { cl->perm_lo[k] = lo[k]; cl->perm_hi[k] = hi[k]; }  // valid only now
    return PEDP_OK;
}

// Region of the scene frame the target can reach: corners of (target box + r) through inv(init).
void scene_roi(pedp_cloud_t tgt, double r, const double init[16], double roi[6]) {
    // init maps scene -> target: x_t = R x_s + t  =>  x_s = R^T (x_t - t)
    const double rr = std::isfinite(r) && r < 1e15 ? r : 0.0;
    for (int k = 0; k < 3; ++k) { roi[k] = 1e300; roi[3 + k] = -1e300; }
    for (int corner = 0; corner < 8; ++corner) {
        double p[3];
        for (int k = 0; k < 3; ++k) p[k] = ((corner >> k) & 1 ? tgt->hi[k] + rr : tgt->lo[k] - rr) - init[4 * k + 3];
        for (int k = 0; k < 3; ++k) {
            const double v = init[0 * 4 + k] * p[0] + init[1 * 4 + k] * p[1] + init[2 * 4 + k] * p[2];
            if (v < roi[k]) roi[k] = v;
            if (v > roi[3 + k]) roi[3 + k] = v;
        }
    }
}

// Target-side operand of the sweep, built once per cloud and kept in the handle (the
// reference re-runs ICP ~50x per frame against the same model, pose_estimation.py:577-613).
int ensure_target_pack(pedp_ctx_t c, pedp_cloud_t tgt, TargetPrep &tp) {
    for (int k = 0; k < 3; ++k) tp.c[k] = tgt->centroid[k];
    tp.Tn = tgt->Tn;
    tp.T2 = tgt->T2;
    for (int k = 0; k < 3; ++k) { tp.lo[k] = tgt->lo[k]; tp.hi[k] = tgt->hi[k]; }
    if (tgt->tgt4) return PEDP_OK;
    int rc = ensure_spatial_perm(c, tgt, nullptr);
    if (rc) return rc;
    // real tiles rounded to NN_TU, plus readable pad tiles the pipelined sweep may prefetch
    int64_t pad = (int64_t)align_up((size_t)(tgt->N > 0 ? tgt->N : 1), 16 * NN_TU) + 16 * NN_TILE_PAD;
    // all three or none: a half-built pack must not look finished to the next call
    void *t4 = nullptr, *s1 = nullptr, *s4 = nullptr;
    hipError_t e = hipMalloc(&t4, sizeof(float4) * (size_t)pad);
    if (e == hipSuccess) e = hipMalloc(&s1, sizeof(float4) * (size_t)(pad / 16));
    if (e == hipSuccess) e = hipMalloc(&s4, sizeof(float4) * (size_t)(pad / 64));
    if (e != hipSuccess) {
        if (t4) (void)hipFree(t4);
        if (s1) (void)hipFree(s1);
        if (s4) (void)hipFree(s4);
        pedp_set_error("pedp_icp: target pack allocation failed: %s", hipGetErrorString(e));
        return PEDP_ERR_ALLOC;
    }
    tgt->tgt4 = t4;
    tgt->tile_sph = s1;
    tgt->tile_sph4 = s4;
    tgt->tgt4_pad = pad;
    int64_t grid = (pad + 255) / 256;
    hipLaunchKernelGGL(pack_target_kernel, dim3((unsigned)grid), dim3(256), 0, c->stream, tgt->pts,
                       (const int32_t *)tgt->perm, tgt->N, pad, tp.c[0], tp.c[1], tp.c[2], (float4 *)tgt->tgt4);
    hipLaunchKernelGGL(tile_sphere_kernel, dim3((unsigned)((pad / 16 + 255) / 256)), dim3(256), 0, c->stream,
                       (const float4 *)tgt->tgt4, tgt->N, pad / 16, 16, (float4 *)tgt->tile_sph);
    hipLaunchKernelGGL(tile_sphere_kernel, dim3((unsigned)((pad / 64 + 255) / 256)), dim3(256), 0, c->stream,
                       (const float4 *)tgt->tgt4, tgt->N, pad / 64, 64, (float4 *)tgt->tile_sph4);
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

// Cached per-cloud preparation on the owner's stream: sorted target operand + unit spheres,
// spatial order of the scene over the region the target can reach from `init`.
int icp_prepare(pedp_ctx_t c, pedp_cloud_t source, pedp_cloud_t target, double r, const double *inits, int B, TargetPrep &tp) {
    int rc = ensure_target_pack(c, target, tp);
    if (rc) return rc;
    double roi[6];
    scene_roi(target, r, inits, roi);
    for (int b = 1; b < B; ++b) {  // a batch orders the scene over the union of its start poses
        double rb[6];
        scene_roi(target, r, inits + 16 * b, rb);
        for (int k = 0; k < 3; ++k) {
            roi[k] = rb[k] < roi[k] ? rb[k] : roi[k];
            roi[3 + k] = rb[3 + k] > roi[3 + k] ? rb[3 + k] : roi[3 + k];
        }
    }
    return ensure_spatial_perm(c, source, roi);
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
    int rc = carve_workspace(x, job.Ns, job.Nt, job.max_iter, job.qt, w);
    if (rc) return rc;
    w.tgt4 = (const float4 *)target->tgt4;
    w.tile_sph = (const float4 *)(job.qt == 4 ? target->tile_sph4 : target->tile_sph);
    w.tgt_perm = (const int32_t *)target->perm;
    w.src_perm = (const int32_t *)source->perm;
    return PEDP_OK;
}

// start state of a registration in the executor's pinned block (uploaded by the first node of
// icp_enqueue; the same block receives the final state)
void icp_fill_state(pedp_ctx_t x, const TargetPrep &tp, const double init[16]) {
    IcpState h{};
    for (int k = 0; k < 16; ++k) { h.T[k] = init[k]; h.upd[k] = init[k]; }
    for (int k = 0; k < 3; ++k) h.centroid[k] = tp.c[k];
    *(IcpState *)x->pinned = h;
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
    for (int pass = 0; pass <= max_iter; ++pass) {
        if (!degenerate) {
            rc = enqueue_nn_pass(x, w, source, target, pass == 0 ? 0 : 1, tp, r, pass == job.timed_pass, job.exhaustive);
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
        PEDP_HIP_CHECK(hipMemcpyAsync(corr, w.idx, sizeof(int32_t) * (size_t)job.Ns, hipMemcpyDeviceToHost, x->stream));
    if (trace)
        PEDP_HIP_CHECK(hipMemcpyAsync(trace, w.trace, sizeof(double) * 18 * (size_t)(job.max_iter + 1), hipMemcpyDeviceToHost, x->stream));
    PEDP_HIP_CHECK(hipStreamSynchronize(x->stream));
    const IcpState *hp = (const IcpState *)x->pinned;
    for (int k = 0; k < 16; ++k) T_out[k] = hp->T[k];
    x->icp_last_cand = hp->sum_tiles * (16 * job.qt) * (NN_SB * 16);  // (scene slot, target point) pairs the MFMAs evaluated
    x->icp_last_fb = hp->sum_fb;
    x->icp_last_passes = hp->iters + 1;
    x->icp_last_nt = job.Nt;
    if (fitness) *fitness = hp->fitness;
    if (inlier_rmse) *inlier_rmse = hp->rmse;
    if (n_iter_done) *n_iter_done = hp->iters;
    return PEDP_OK;
}

// Batched registrations are bound by the host's launch rate (a pass is 8 small kernels), so a
// sub-context captures the whole pass sequence once into a hipGraph and replays it per start
// pose: the graph's nodes read the start state from the executor's pinned block and everything
// else from device memory.  Anything the captured launches depend on is in the key.
int icp_launch_replayed(pedp_ctx_t x, pedp_cloud_t source, pedp_cloud_t target, const TargetPrep &tp,
                        const pedp_icp_params *prm, const double init[16], IcpJob &job) {
    int rc = icp_job_setup(x, source, target, prm, job);
    if (rc) return rc;
    icp_fill_state(x, tp, init);
    pedp_icp_graph_key key;
    key.src_gen = source->gen; key.tgt_gen = target->gen; key.ws = x->icp_ws.ptr;
    key.Ns = job.Ns; key.Nt = job.Nt; key.max_iter = job.max_iter; key.qt = job.qt; key.estimator = prm->estimator;
    key.r = prm->max_correspondence_distance;
    const pedp_icp_graph_key &have = x->icp_graph_key;
    const bool same = x->icp_graph && have.src_gen == key.src_gen && have.tgt_gen == key.tgt_gen && have.ws == key.ws && have.Ns == key.Ns &&
                      have.Nt == key.Nt && have.max_iter == key.max_iter && have.qt == key.qt &&
                      have.estimator == key.estimator && have.r == key.r;
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

}  // namespace

extern "C" {

int pedp_icp(pedp_ctx_t c, pedp_cloud_t source, pedp_cloud_t target, const pedp_icp_params *prm,
             const double init[16], double T_out[16], double *fitness, double *inlier_rmse,
             int32_t *n_iter_done, int32_t *corr, double *trace) {
    PEDP_REQUIRE(c && source && target && prm && init && T_out, "pedp_icp: null argument");
    int rc = icp_check_args(c, source, target, prm);
    if (rc) return rc;
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    TargetPrep tp;
    rc = icp_prepare(c, source, target, prm->max_correspondence_distance, init, 1, tp);
    if (rc) return rc;
    IcpJob job;
    rc = icp_job_setup(c, source, target, prm, job);
    if (rc) return rc;
    icp_fill_state(c, tp, init);
    rc = icp_enqueue(c, source, target, tp, prm, trace != nullptr, true, job);
    if (rc) return rc;
    return icp_collect(c, job, T_out, fitness, inlier_rmse, n_iter_done, corr, trace);
}

// Hypotheses are independent: up to PEDP_MAX_SUB registrations are in flight at once, each on
// its own stream and workspace (sub-contexts of c), sharing the clouds' cached preparation.
int pedp_icp_batched(pedp_ctx_t c, pedp_cloud_t source, pedp_cloud_t target, const pedp_icp_params *prm,
                     const double *inits, int B, double *T_out, double *fitness, double *inlier_rmse) {
    PEDP_REQUIRE(c && source && target && prm && inits && T_out, "pedp_icp_batched: null argument");
    PEDP_REQUIRE(B >= 0, "pedp_icp_batched: negative batch");
    PEDP_REQUIRE(!prm->allreduce && !prm->use_comm, "pedp_icp_batched: hypotheses shard across ranks, not within one registration");
    if (B == 0) return PEDP_OK;
    int rc = icp_check_args(c, source, target, prm);
    if (rc) return rc;
    pedp_icp_params p = *prm;
    p.relative_fitness = -1.0;  // no early exit across the batch
    p.relative_rmse = -1.0;
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    TargetPrep tp;
    rc = icp_prepare(c, source, target, p.max_correspondence_distance, inits, B, tp);
    if (rc) return rc;
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));  // preparation is visible to the sub-streams
    const int K = B < PEDP_MAX_SUB ? B : PEDP_MAX_SUB;
    for (int k = 0; k < K; ++k) {
        if (!c->sub[k]) {
            rc = pedp_ctx_create(c->device, nullptr, &c->sub[k]);
            if (rc) return rc;
        }
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
            rc = icp_launch_replayed(c->sub[k], source, target, tp, &p, inits + 16 * (b0 + k), jobs[k]);
            if (rc) return drain(rc);
        }
        for (int k = 0; k < n; ++k) {
            rc = icp_collect(c->sub[k], jobs[k], T_out + 16 * (b0 + k), fitness ? fitness + b0 + k : nullptr,
                             inlier_rmse ? inlier_rmse + b0 + k : nullptr, nullptr, nullptr, nullptr);
            if (rc) return drain(rc);
            c->icp_last_cand += c->sub[k]->icp_last_cand;
            c->icp_last_fb += c->sub[k]->icp_last_fb;
            c->icp_last_passes += c->sub[k]->icp_last_passes;
        }
    }
    return PEDP_OK;
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
    if (rc) return rc;
    w.tgt4 = (const float4 *)target->tgt4;
    w.tile_sph = (const float4 *)target->tile_sph4;
    w.tgt_perm = (const int32_t *)target->perm;
    rc = ensure_spatial_perm(c, source, nullptr);  // no radius: order over the whole cloud
    if (rc) return rc;
    w.src_perm = (const int32_t *)source->perm;
    IcpState *hp = (IcpState *)c->pinned;
    IcpState h{};
    for (int k = 0; k < 16; ++k) { h.T[k] = T[k]; h.upd[k] = T[k]; }
    for (int k = 0; k < 3; ++k) h.centroid[k] = tp.c[k];
    *hp = h;
    PEDP_HIP_CHECK(hipMemcpyAsync(w.st, hp, sizeof(IcpState), hipMemcpyHostToDevice, c->stream));
    // every point is a candidate: radius = "infinite" (cloud scale bound)
    rc = enqueue_nn_pass(c, w, source, target, 0, tp, 1e18, true);
    if (rc) return rc;
    PEDP_HIP_CHECK(hipMemcpyAsync(idx, w.idx, sizeof(int32_t) * (size_t)source->N, hipMemcpyDeviceToHost, c->stream));
    PEDP_HIP_CHECK(hipMemcpyAsync(d2, w.d2, sizeof(double) * (size_t)source->N, hipMemcpyDeviceToHost, c->stream));
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PEDP_OK;
}

int pedp_icp_configure(pedp_ctx_t c, int exhaustive, int timed_pass) {
    PEDP_REQUIRE(c, "pedp_icp_configure: null context");
    c->icp_exhaustive = exhaustive != 0;
    c->icp_timed_pass = timed_pass;
    for (int k = 0; k < PEDP_MAX_SUB; ++k)  // captured graphs bake the mode in
        if (c->sub[k] && c->sub[k]->icp_graph) { (void)hipGraphExecDestroy(c->sub[k]->icp_graph); c->sub[k]->icp_graph = nullptr; }
    return PEDP_OK;
}

int pedp_icp_last_stats(pedp_ctx_t c, int64_t *passes, int64_t *pairs_swept, int64_t *fallback_points) {
    PEDP_REQUIRE(c, "pedp_icp_last_stats: null context");
    if (passes) *passes = c->icp_last_passes;
    if (pairs_swept) *pairs_swept = c->icp_last_cand;
    if (fallback_points) *fallback_points = c->icp_last_fb;
    return PEDP_OK;
}

int pedp_nn_last_sweep_ms(pedp_ctx_t c, float *ms) {
    PEDP_REQUIRE(c && ms, "pedp_nn_last_sweep_ms: null argument");
    PEDP_REQUIRE(c->nn_timed, "pedp_nn_last_sweep_ms: no timed sweep has run on this context");
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    PEDP_HIP_CHECK(hipEventSynchronize(c->nn_ev1));
    PEDP_HIP_CHECK(hipEventElapsedTime(ms, c->nn_ev0, c->nn_ev1));
    return PEDP_OK;
}

}  // extern "C"
