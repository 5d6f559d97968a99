// Depth pre-filters: the reference's warp-lang kernels and its depth back-projection
// as HIP for gfx950.
//   erode_depth_kernel             Utils.py:356-383   (launcher :386-396, callers estimater.py:171, :255)
//   bilateral_filter_depth_kernel  Utils.py:304-345   (launcher :347-357, callers estimater.py:172, :256)
//   depth2xyzmap                   Utils.py:401-420   (callers run.py:89, estimater.py:175, :212)
//   depth2xyzmap_batch             Utils.py:423-442   (caller estimater.py:259)
// All float32 like the warp kernels (`float` = f32 there); depth2xyzmap forms x, y in float64 and
// stores float32 (numpy promotes (u - cx) * z / fx to float64, the map is float32).
//
// Stencils: one workgroup = 64 x 16 output pixels, staged with their halo in LDS; a thread owns a
// 1 x 4 column strip and walks the window column by column (u outer, v inner: the reference's
// loop order, which fixes the float32 summation order of the bilateral filter), keeping each
// window column in registers for its four outputs.  Compulsory traffic is 8 B per pixel (4 in,
// 4 out): the erode kernel is HBM-bound, the bilateral kernel is bound by its 25 exp per pixel.
#include "pedp_internal.h"
#include <algorithm>

namespace {

// Camera frames (a few hundred workgroups in all) take 64 x 16 tiles, a thread owning 4 rows; the tiles are handed out
// so that every XCD works through one contiguous run of them (workgroup i runs on XCD i % 8).  Large images take the
// band walkers further down.
constexpr int WIDE_PIXELS = 1 << 20;  // images from here on take the band walkers

// tile origin of this workgroup: tiles in row-major order, dealt to the XCDs in eight contiguous runs
__device__ __forceinline__ bool tile_origin(int tiles_x, int tiles_total, int tile_w, int tile_h, int &x0, int &y0) {
    const int per = (tiles_total + 7) >> 3, logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (logical >= tiles_total) return false;
    const int ty = logical / tiles_x;
    x0 = (logical - ty * tiles_x) * tile_w;
    y0 = ty * tile_h;
    return true;
}

struct StencilArgs {
    int H, W, radius;
    float zfar;
    float a, b;  // erode: depth_diff_thres, ratio_thres; bilateral: sigmaD, sigmaR
};

__device__ __forceinline__ bool depth_valid(float d, float zfar) { return d >= 0.001f && d < zfar; }

// erode: out = 0 if more than ratio_thres of the window is invalid or differs from the centre by
// more than depth_diff_thres, else the centre value unchanged (also when the centre itself is
// invalid: the reference's first assignment is overwritten by the second, Utils.py:363-383).
__device__ __forceinline__ float erode_finish(float d_ori, float bad, float total, float ratio) {
    return (bad / total > ratio) ? 0.0f : d_ori;
}

// The window test per neighbour is "invalid, or farther than depth_diff_thres from the centre".
// Staged values carry the validity: an invalid reading is staged as +inf and a pixel outside the
// image as NaN, so for a finite centre d the test is the single comparison |v - d| > thres
// (+inf: always true; NaN: never, like the reference's comparisons on a NaN reading), and the
// window size is the closed-form count of in-image pixels.  Centres that are NaN or infinite --
// where |v - d| is NaN -- count by class instead: a NaN centre makes exactly the invalid readings
// bad, an infinite centre everything that is not NaN.
// bad += the number of the five differences beyond thr.  The compares are issued together into five mask registers and
// counted afterwards: a compare's mask may not be read by the next two instructions on gfx950, and with one mask
// register per compare the compiler's schedule was compare, two idle slots, count (a third of the issue slots idle).
// Ten instructions for five tests; two tests share one add-with-carry.
__device__ __forceinline__ void count5_beyond(int &bad, float d0, float d1, float d2, float d3, float d4, float thr) {
    unsigned long long m0, m1, m2, m3;
    int x0, x2;
    asm("v_cmp_gt_f32_e64 %[m0], |%[d0]|, %[t]\n\t"
        "v_cmp_gt_f32_e64 %[m1], |%[d1]|, %[t]\n\t"
        "v_cmp_gt_f32_e64 %[m2], |%[d2]|, %[t]\n\t"
        "v_cmp_gt_f32_e64 %[m3], |%[d3]|, %[t]\n\t"
        "v_cmp_gt_f32_e64 vcc, |%[d4]|, %[t]\n\t"
        "v_cndmask_b32_e64 %[x0], 0, 1, %[m0]\n\t"
        "v_addc_co_u32_e64 %[bad], %[m1], %[bad], %[x0], %[m1]\n\t"
        "v_cndmask_b32_e64 %[x2], 0, 1, %[m2]\n\t"
        "v_addc_co_u32_e64 %[bad], %[m3], %[bad], %[x2], %[m3]\n\t"
        "v_addc_co_u32_e32 %[bad], vcc, 0, %[bad], vcc"
        : [bad] "+v"(bad), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [x0] "=&v"(x0), [x2] "=&v"(x2)
        : [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [d4] "v"(d4), [t] "s"(thr)
        : "vcc");
}

template <int R, int TILE_W, int STRIP>
__global__ __launch_bounds__(256) void erode_kernel(const float *__restrict__ depth, float *__restrict__ out, StencilArgs p, int tiles_x,
                                                    int tiles_total) {
    constexpr int TILE_H = (256 / TILE_W) * STRIP, TW = TILE_W + 2 * R, TH = TILE_H + 2 * R;
    __shared__ float tile[TH][TW];
    int x0, y0;
    if (!tile_origin(tiles_x, tiles_total, TILE_W, TILE_H, x0, y0)) return;  // workgroup-uniform
    const float inf = __builtin_inff(), nan = __builtin_nanf("");
    for (int i = threadIdx.x; i < TW * TH; i += 256) {
        const int ty = i / TW, tx = i - ty * TW, gy = y0 + ty - R, gx = x0 + tx - R;
        float v = nan;
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
            v = depth[(size_t)gy * p.W + gx];
            v = (v < 0.001f || v >= p.zfar) ? inf : v;
        }
        tile[ty][tx] = v;
    }
    __syncthreads();
    const int lx = threadIdx.x % TILE_W, ly = threadIdx.x / TILE_W;
    const int x = x0 + lx, ys = y0 + ly * STRIP;
    if (x >= p.W) return;
    float d_ori[STRIP];
    int bad[STRIP];
    int cls[STRIP];  // 0 finite centre, 1 NaN, 2 infinite
#pragma unroll
    for (int j = 0; j < STRIP; ++j) {
        d_ori[j] = (ys + j < p.H) ? depth[(size_t)(ys + j) * p.W + x] : 0.0f;
        bad[j] = 0;
        cls[j] = (d_ori[j] != d_ori[j]) ? 1 : (fabsf(d_ori[j]) == inf ? 2 : 0);
    }
    const int cols = min(x + R, p.W - 1) - max(x - R, 0) + 1;
#pragma unroll
    for (int du = -R; du <= R; ++du) {
        float col[STRIP + 2 * R];
#pragma unroll
        for (int k = 0; k < STRIP + 2 * R; ++k) col[k] = tile[ly * STRIP + k][lx + du + R];  // NaN outside the image
#pragma unroll
        for (int j = 0; j < STRIP; ++j) {
            if constexpr (R == 2)
                count5_beyond(bad[j], col[j] - d_ori[j], col[j + 1] - d_ori[j], col[j + 2] - d_ori[j], col[j + 3] - d_ori[j],
                              col[j + 4] - d_ori[j], p.a);
            else {
#pragma unroll
                for (int dv = -R; dv <= R; ++dv) bad[j] += (fabsf(col[j + dv + R] - d_ori[j]) > p.a) ? 1 : 0;
            }
        }
    }
    // NaN / infinite centres (rare; the branch is taken by a wave only if one of its lanes has one)
    int any_cls = 0;
#pragma unroll
    for (int j = 0; j < STRIP; ++j) any_cls |= cls[j];
    if (any_cls) {
#pragma unroll
        for (int j = 0; j < STRIP; ++j) {
            if (cls[j] == 0) continue;
            int b = 0;
            for (int du = -R; du <= R; ++du)
                for (int dv = -R; dv <= R; ++dv) {
                    const float v = tile[ly * STRIP + j + dv + R][lx + du + R];
                    b += (cls[j] == 1) ? (v == inf) : (v == v);
                }
            bad[j] = b;
        }
    }
#pragma unroll
    for (int j = 0; j < STRIP; ++j) {
        const int y = ys + j;
        if (y >= p.H) continue;
        const int rows = min(y + R, p.H - 1) - max(y - R, 0) + 1;
        out[(size_t)y * p.W + x] = erode_finish(d_ori[j], (float)bad[j], (float)(rows * cols), p.b);
    }
}

// Large images: a workgroup WALKS down a band of 64 columns, 16 rows a step.  The rows of the band live in a ring of
// four 16-row slots in LDS (the step's block, the one above, the one below, and the slot the next block is being
// written to), so every image row is fetched once per band -- no halo rows -- and the next block's loads are in flight
// while the current block is counted: one barrier per step, nothing waits for memory but the prologue.  The bands are
// dealt to the XCDs in eight contiguous runs (workgroup i runs on XCD i % 8), neighbouring bands walk in step, and the
// two halo columns either side come out of the XCD's L2.  Per step and thread: five loads for the ring, four centre
// readings for the next step (the ring holds the staged values -- invalid readings as +inf -- and an invalid centre
// needs the reading itself), the 4 x 25 tests of its 1 x 4 strip, four stores.
constexpr int WALK_W = 64, WALK_BH = 16, WALK_SLOTS = 4, WALK_STRIP = 4;
struct WalkArgs { int bands, segs, seg_rows; };

__device__ __forceinline__ bool walk_origin(const WalkArgs &w, int H, int &x0, int &ys0, int &ys1) {
    const int n = w.bands * w.segs, per = (n + 7) >> 3, logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (logical >= n) return false;
    const int band = logical / w.segs, seg = logical - band * w.segs;  // band-major: an XCD's run is whole bands
    x0 = band * WALK_W;
    ys0 = seg * w.seg_rows;
    ys1 = min(ys0 + w.seg_rows, H);
    return ys0 < H;
}

template <int R>
__global__ __launch_bounds__(256) void erode_walk_kernel(const float *__restrict__ depth, float *__restrict__ out, StencilArgs p,
                                                         WalkArgs w) {
    constexpr int TW = WALK_W + 2 * R, NL = (WALK_BH * TW + 255) / 256, STRIP = WALK_STRIP, RING = WALK_SLOTS * WALK_BH;
    static_assert(R <= WALK_BH && (RING & (RING - 1)) == 0, "halo within one block, ring a power of two");
    __shared__ float ring[RING][TW];
    int x0, ys0, ys1;
    if (!walk_origin(w, p.H, x0, ys0, ys1)) return;  // workgroup-uniform
    const float inf = __builtin_inff(), nan = __builtin_nanf("");
    const int n_blocks = (ys1 - ys0 + WALK_BH - 1) / WALK_BH;
    // element e of a block: row e / TW, column e % TW of the band with its halo columns.  What does not change from block to
    // block is worked out once per thread: the element's offset into the image for block 0 (32 bits: an image is below
    // 2^31 pixels), whether its column is in the image, its place in a ring slot.
    unsigned f_off[NL];   // (ys0 + row) * W + gx for block 0 (meaningless where f_col is false)
    int f_row[NL], f_lds[NL];
    bool f_col[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int e = (int)threadIdx.x + 256 * i, row = e / TW, col = e - row * TW, gx = x0 - R + col;
        f_row[i] = row;
        f_lds[i] = row * TW + col;
        f_col[i] = e < WALK_BH * TW && gx >= 0 && gx < p.W;
        f_off[i] = (unsigned)((long long)(ys0 + row) * p.W + gx);
    }
    const int g_lo = max(ys0 - R, 0), g_hi = min(ys1 + R, p.H);  // the image rows this segment has any use for
    float *const ring_flat = &ring[0][0];
    auto fetch = [&](int b, float (&v)[NL]) {
        const unsigned step = (unsigned)(WALK_BH * b) * (unsigned)p.W;  // (wraps for b = -1 like the sum below)
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int g = ys0 + WALK_BH * b + f_row[i];
            const bool need = f_col[i] && g >= g_lo && g < g_hi;
            v[i] = need ? depth[f_off[i] + step] : nan;  // outside the image (or of no use to this segment): NaN
        }
    };
    auto park = [&](int b, const float (&v)[NL]) {  // block b (-1 ..) lives in slot (b + 1) % 4
        const int slot = ((b + 1) & (WALK_SLOTS - 1)) * (WALK_BH * TW);
#pragma unroll
        for (int i = 0; i < NL; ++i)
            if ((int)threadIdx.x + 256 * i < WALK_BH * TW) ring_flat[slot + f_lds[i]] = (v[i] < 0.001f || v[i] >= p.zfar) ? inf : v[i];
    };
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6, x = x0 + lx;
    const bool mine = x < p.W;
    auto centres = [&](int b, float (&d)[STRIP]) {
#pragma unroll
        for (int j = 0; j < STRIP; ++j) {
            const int y = ys0 + WALK_BH * b + ly * STRIP + j;
            d[j] = (mine && y < ys1) ? depth[(size_t)y * p.W + x] : 0.0f;
        }
    };
    float d_ori[STRIP];
    {
        float va[NL], vb[NL], vc[NL];
        fetch(-1, va); fetch(0, vb); fetch(1, vc);
        centres(0, d_ori);
        park(-1, va); park(0, vb); park(1, vc);
    }
    __syncthreads();
    const int cols = min(x + R, p.W - 1) - max(x - R, 0) + 1;
    // bad / total > ratio is monotone in the count: for the full window the division is done once per thread -- the
    // smallest count that passes -- and every interior pixel compares integers (a float division is a dozen lane-ops)
    constexpr int FULL = (2 * R + 1) * (2 * R + 1);
    int full_from = FULL + 1;
    for (int n = FULL; n >= 0; --n)
        if ((float)n / (float)FULL > p.b) full_from = n;
    for (int b = 0; b < n_blocks; ++b) {
        float nv[NL], d_next[STRIP];
        const bool more = b + 2 <= n_blocks;  // block n_blocks holds the halo rows below the segment
        if (more) fetch(b + 2, nv);
        if (b + 1 < n_blocks) centres(b + 1, d_next);
        int bad[STRIP], any_cls = 0;
        const int base = (b + 1) * WALK_BH + ly * STRIP - R;  // ring row of the strip's first window row
#pragma unroll
        for (int j = 0; j < STRIP; ++j) {
            bad[j] = 0;
            any_cls |= (d_ori[j] != d_ori[j] || fabsf(d_ori[j]) == inf) ? 1 : 0;
        }
#pragma unroll
        for (int du = -R; du <= R; ++du) {
            float col[STRIP + 2 * R];
#pragma unroll
            for (int k = 0; k < STRIP + 2 * R; ++k) col[k] = ring[(base + k) & (RING - 1)][lx + du + R];
#pragma unroll
            for (int j = 0; j < STRIP; ++j) {
                if constexpr (R == 2)
                    count5_beyond(bad[j], col[j] - d_ori[j], col[j + 1] - d_ori[j], col[j + 2] - d_ori[j], col[j + 3] - d_ori[j],
                                  col[j + 4] - d_ori[j], p.a);
                else {
#pragma unroll
                    for (int dv = -R; dv <= R; ++dv) bad[j] += (fabsf(col[j + dv + R] - d_ori[j]) > p.a) ? 1 : 0;
                }
            }
        }
        if (any_cls) {  // NaN / infinite centres count by class (erode_kernel)
#pragma unroll
            for (int j = 0; j < STRIP; ++j) {
                const int cls = (d_ori[j] != d_ori[j]) ? 1 : (fabsf(d_ori[j]) == inf ? 2 : 0);
                if (cls == 0) continue;
                int n = 0;
                for (int du = -R; du <= R; ++du)
                    for (int dv = -R; dv <= R; ++dv) {
                        const float v = ring[(base + j + dv + R) & (RING - 1)][lx + du + R];
                        n += (cls == 1) ? (v == inf) : (v == v);
                    }
                bad[j] = n;
            }
        }
#pragma unroll
        for (int j = 0; j < STRIP; ++j) {
            const int y = ys0 + WALK_BH * b + ly * STRIP + j;
            if (!mine || y >= ys1) continue;
            const int rows = min(y + R, p.H - 1) - max(y - R, 0) + 1;
            out[(size_t)y * p.W + x] = rows * cols == FULL ? (bad[j] >= full_from ? 0.0f : d_ori[j])
                                                           : erode_finish(d_ori[j], (float)bad[j], (float)(rows * cols), p.b);
        }
        if (more) park(b + 2, nv);  // slot (b + 3) % 4: no one reads it in this step
#pragma unroll
        for (int j = 0; j < STRIP; ++j) d_ori[j] = d_next[j];
        __syncthreads();
    }
}

// any radius: one thread per pixel straight from global memory (the caches carry the reuse)
__global__ __launch_bounds__(256) void erode_generic_kernel(const float *__restrict__ depth, float *__restrict__ out,
                                                            StencilArgs p) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.W || y >= p.H) return;
    const float d_ori = depth[(size_t)y * p.W + x];
    float bad = 0.f, total = 0.f;
    for (int u = x - p.radius; u <= x + p.radius; ++u) {
        if (u < 0 || u >= p.W) continue;
        for (int v = y - p.radius; v <= y + p.radius; ++v) {
            if (v < 0 || v >= p.H) continue;
            const float cur = depth[(size_t)v * p.W + u];
            total += 1.0f;
            if (cur < 0.001f || cur >= p.zfar || fabsf(cur - d_ori) > p.a) bad += 1.0f;
        }
    }
    out[(size_t)y * p.W + x] = erode_finish(d_ori, bad, total, p.b);
}

// bilateral weight of one neighbour (Utils.py:341): every operation float32, no contraction
__device__ __forceinline__ float bilateral_weight(int du, int dv, float centre, float cur, float two_sd2, float two_sr2) {
    const float a = __fdiv_rn(-(float)(du * du + dv * dv), two_sd2);
    const float dc = __fsub_rn(centre, cur);
    const float b = __fdiv_rn(__fmul_rn(dc, dc), two_sr2);
    return expf(__fsub_rn(a, b));
}

template <int R, int TILE_W, int STRIP>
__global__ __launch_bounds__(256) void bilateral_kernel(const float *__restrict__ depth, float *__restrict__ out,
                                                        StencilArgs p, int tiles_x, int tiles_total) {
    constexpr int TILE_H = (256 / TILE_W) * STRIP, TW = TILE_W + 2 * R, TH = TILE_H + 2 * R;
    __shared__ float tile[TH][TW];
    int x0, y0;
    if (!tile_origin(tiles_x, tiles_total, TILE_W, TILE_H, x0, y0)) return;  // workgroup-uniform
    for (int i = threadIdx.x; i < TW * TH; i += 256) {
        const int ty = i / TW, tx = i - ty * TW, gy = y0 + ty - R, gx = x0 + tx - R;
        tile[ty][tx] = (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? depth[(size_t)gy * p.W + gx] : 0.0f;
    }
    __syncthreads();
    const int lx = threadIdx.x % TILE_W, ly = threadIdx.x / TILE_W;
    const int x = x0 + lx, ys = y0 + ly * STRIP;
    if (x >= p.W) return;
    const float two_sd2 = __fmul_rn(__fmul_rn(2.0f, p.a), p.a), two_sr2 = __fmul_rn(__fmul_rn(2.0f, p.b), p.b);
    constexpr int KMAX = 2 * R * R;
    float wa[KMAX + 1], ww[KMAX + 1], wthr[KMAX + 1];
#pragma unroll
    for (int k = 0; k <= KMAX; ++k) {
        wa[k] = __fdiv_rn(-(float)k, two_sd2);
        ww[k] = expf(wa[k]);
        wthr[k] = two_sr2 * fabsf(wa[k]) * 1.4901161e-8f;  // 2^-26 |a| <= ulp(a) / 4
    }
    float mean[STRIP];
    int nv[STRIP];
#pragma unroll
    for (int j = 0; j < STRIP; ++j) { mean[j] = 0.f; nv[j] = 0; }
    // a pixel outside the image is staged as 0, an invalid reading: both passes skip it like the reference's bounds
    // tests do, without testing bounds per tap
#pragma unroll
    for (int du = -R; du <= R; ++du) {
        float col[STRIP + 2 * R];
#pragma unroll
        for (int k = 0; k < STRIP + 2 * R; ++k) col[k] = tile[ly * STRIP + k][lx + du + R];
#pragma unroll
        for (int j = 0; j < STRIP; ++j) {
#pragma unroll
            for (int dv = -R; dv <= R; ++dv) {
                const float cur = col[j + dv + R];
                if (depth_valid(cur, p.zfar)) { ++nv[j]; mean[j] = __fadd_rn(mean[j], cur); }
            }
        }
    }
    float sw[STRIP], sum[STRIP], centre[STRIP];
#pragma unroll
    for (int j = 0; j < STRIP; ++j) {
        mean[j] = nv[j] ? __fdiv_rn(mean[j], (float)nv[j]) : 0.f;
        sw[j] = 0.f; sum[j] = 0.f;
        centre[j] = tile[ly * STRIP + j + R][lx + R];
    }
#pragma unroll
    for (int du = -R; du <= R; ++du) {
        float col[STRIP + 2 * R];
#pragma unroll
        for (int k = 0; k < STRIP + 2 * R; ++k) col[k] = tile[ly * STRIP + k][lx + du + R];
#pragma unroll
        for (int j = 0; j < STRIP; ++j) {
#pragma unroll
            for (int dv = -R; dv <= R; ++dv) {
                const float cur = col[j + dv + R];
                if (depth_valid(cur, p.zfar) && fabsf(__fsub_rn(cur, mean[j])) < 0.01f) {
                    const int kk = du * du + dv * dv;
                    const float dc = __fsub_rn(centre[j], cur);
                    const float w = __fmul_rn(dc, dc) < wthr[kk] ? ww[kk]
                                                               : bilateral_weight(du, dv, centre[j], cur, two_sd2, two_sr2);
                    sw[j] = __fadd_rn(sw[j], w);
                    sum[j] = __fadd_rn(sum[j], __fmul_rn(w, cur));
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < STRIP; ++j)
        if (ys + j < p.H) out[(size_t)(ys + j) * p.W + x] = (sw[j] > 0.f && nv[j] > 0) ? __fdiv_rn(sum[j], sw[j]) : 0.0f;
}

__global__ __launch_bounds__(256) void bilateral_generic_kernel(const float *__restrict__ depth, float *__restrict__ out,
                                                                StencilArgs p) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.W || y >= p.H) return;
    const float two_sd2 = __fmul_rn(__fmul_rn(2.0f, p.a), p.a), two_sr2 = __fmul_rn(__fmul_rn(2.0f, p.b), p.b);
    float mean = 0.f;
    int nv = 0;
    for (int u = x - p.radius; u <= x + p.radius; ++u) {
        if (u < 0 || u >= p.W) continue;
        for (int v = y - p.radius; v <= y + p.radius; ++v) {
            if (v < 0 || v >= p.H) continue;
            const float cur = depth[(size_t)v * p.W + u];
            if (depth_valid(cur, p.zfar)) { ++nv; mean = __fadd_rn(mean, cur); }
        }
    }
    float res = 0.0f;
    if (nv) {
        mean = __fdiv_rn(mean, (float)nv);
        const float centre = depth[(size_t)y * p.W + x];
        float sw = 0.f, sum = 0.f;
        for (int u = x - p.radius; u <= x + p.radius; ++u) {
            if (u < 0 || u >= p.W) continue;
            for (int v = y - p.radius; v <= y + p.radius; ++v) {
                if (v < 0 || v >= p.H) continue;
                const float cur = depth[(size_t)v * p.W + u];
                if (depth_valid(cur, p.zfar) && fabsf(__fsub_rn(cur, mean)) < 0.01f) {
                    const float w = bilateral_weight(u - x, y - v, centre, cur, two_sd2, two_sr2);
                    sw = __fadd_rn(sw, w);
                    sum = __fadd_rn(sum, __fmul_rn(w, cur));
                }
            }
        }
        if (sw > 0.f) res = __fdiv_rn(sum, sw);
    }
    out[(size_t)y * p.W + x] = res;
}

// depth2xyzmap: x = (u - cx) * z / fx in float64, stored as float32; depth < 0.001 -> (0, 0, 0)
__global__ __launch_bounds__(256) void xyzmap_kernel(const float *__restrict__ depth, int H, int W, double fx, double fy,
                                                     double cx, double cy, float *__restrict__ xyz) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int v = (int)(i / W), u = (int)(i - (int64_t)v * W);
    const float z = depth[i];
    float X = 0.f, Y = 0.f, Z = 0.f;
    if (!(z < 0.001f)) {
        X = (float)__ddiv_rn(__dmul_rn(__dsub_rn((double)u, cx), (double)z), fx);
        Y = (float)__ddiv_rn(__dmul_rn(__dsub_rn((double)v, cy), (double)z), fy);
        Z = z;
    }
    xyz[3 * i] = X; xyz[3 * i + 1] = Y; xyz[3 * i + 2] = Z;
}

// depth2xyzmap_batch: float32 throughout, per-image intrinsics, invalid = z < 0.001 or z > zfar
__global__ __launch_bounds__(256) void xyzmap_batch_kernel(const float *__restrict__ depth, int64_t HW, int W,
                                                           const float *__restrict__ Ks /* B x 9 */, float zfar,
                                                           float *__restrict__ xyz) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const int b = blockIdx.y;
    const float *K = Ks + 9 * b;
    const int v = (int)(i / W), u = (int)(i - (int64_t)v * W);
    const float z = depth[(size_t)b * HW + i];
    float X = 0.f, Y = 0.f, Z = 0.f;
    if (!((z < 0.001f) || (z > zfar))) {
        X = __fdiv_rn(__fmul_rn(__fsub_rn((float)u, K[2]), z), K[0]);
        Y = __fdiv_rn(__fmul_rn(__fsub_rn((float)v, K[5]), z), K[4]);
        Z = z;
    }
    float *o = xyz + 3 * ((size_t)b * HW + i);
    o[0] = X; o[1] = Y; o[2] = Z;
}

// ---- the valid points of a back-projected image, in row-major order, as float64 scaled (metres -> millimetres): what
// run.py's loop hands to preprocess_source.  Count per 2,048-pixel block, one single-workgroup scan of the block totals,
// scatter with the block-local scan redone in LDS (the pattern of pedp_project.hip's compactions).
constexpr int SC_THREADS = 256, SC_PER = 8, SC_BLOCK = SC_THREADS * SC_PER;
__device__ __forceinline__ int sc_block_scan(int v, int *lds, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += lds[w];
    if (total) *total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + inc - v;
}
__global__ __launch_bounds__(SC_THREADS) void scene_count_kernel(const float *__restrict__ xyz, int64_t n, float z_min, int *__restrict__ block_tot) {
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * SC_BLOCK + (int64_t)threadIdx.x * SC_PER;
    int cnt = 0;
    for (int k = 0; k < SC_PER; ++k)
        if (base + k < n && xyz[3 * (base + k) + 2] >= z_min) ++cnt;
    int total;
    (void)sc_block_scan(cnt, lds, &total);
    if (threadIdx.x == 0) block_tot[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void scene_scan_kernel(int *__restrict__ block_tot, int n_blocks, int *__restrict__ out_total) {
    __shared__ int part[1024];
    const int per = (n_blocks + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(lo + per, n_blocks);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += block_tot[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 1024; ++i) { const int v = part[i]; part[i] = run; run += v; }
        *out_total = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { const int v = block_tot[i]; block_tot[i] = run; run += v; }
}
__global__ __launch_bounds__(SC_THREADS) void scene_scatter_kernel(const float *__restrict__ xyz, int64_t n, float z_min, double scale,
                                                                   const int *__restrict__ block_off, double *__restrict__ pts) {
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * SC_BLOCK + (int64_t)threadIdx.x * SC_PER;
    unsigned flags = 0;
    for (int k = 0; k < SC_PER; ++k)
        if (base + k < n && xyz[3 * (base + k) + 2] >= z_min) flags |= 1u << k;
    int64_t pos = block_off[blockIdx.x] + sc_block_scan(__popc(flags), lds, nullptr);
    for (int k = 0; k < SC_PER; ++k) {
        if (!(flags >> k & 1u)) continue;
        const float *q = xyz + 3 * (base + k);
        pts[3 * pos] = __dmul_rn((double)q[0], scale);
        pts[3 * pos + 1] = __dmul_rn((double)q[1], scale);
        pts[3 * pos + 2] = __dmul_rn((double)q[2], scale);
        ++pos;
    }
}

int stage_in(pedp_ctx_t c, const float *src, size_t n_in, size_t n_out, int mem, const float **d_in, float **d_out,
             float *out) {
    *d_in = src;
    *d_out = out;
    if (mem == PEDP_HOST) {
        int st = c->ray_in.reserve(sizeof(float) * n_in);
        if (st) return st;
        st = c->ray_out.reserve(sizeof(float) * n_out);
        if (st) return st;
        { int up_ = pedp_upload(c, c->ray_in.ptr, src, sizeof(float) * n_in); if (up_) return up_; }
        *d_in = (const float *)c->ray_in.ptr;
        *d_out = (float *)c->ray_out.ptr;
    }
    return PEDP_OK;
}

int stage_out(pedp_ctx_t c, float *out, const float *d_out, size_t n_out, int mem) {
    PEDP_HIP_CHECK(hipGetLastError());
    if (mem == PEDP_HOST) {
        { int dn_ = pedp_download(c, out, d_out, sizeof(float) * n_out); if (dn_) return dn_; }
        PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    return PEDP_OK;
}

int check_image(pedp_ctx_t c, const void *in, const void *out, int H, int W, int mem, const char *who) {
    PEDP_REQUIRE(c, "%s: null context", who);
    PEDP_REQUIRE(H >= 0 && W >= 0 && (int64_t)H * W < (int64_t)1 << 31, "%s: image size out of range", who);
    PEDP_REQUIRE(mem == PEDP_HOST || mem == PEDP_DEVICE, "%s: bad mem flag %d", who, mem);
    PEDP_REQUIRE((in && out) || (int64_t)H * W == 0, "%s: null arrays", who);
    return PEDP_OK;
}

template <int MODE>  // 0 erode, 1 bilateral
int run_stencil(pedp_ctx_t c, const float *depth, int H, int W, StencilArgs p, int mem, float *out, const char *who) {
    int rc = check_image(c, depth, out, H, W, mem, who);
    if (rc) return rc;
    PEDP_REQUIRE(p.radius >= 0 && p.radius <= 64, "%s: radius out of range", who);
    const size_t n = (size_t)H * W;
    if (n == 0) return PEDP_OK;
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    const float *d_in;
    float *d_out;
    rc = stage_in(c, depth, n, n, mem, &d_in, &d_out, out);
    if (rc) return rc;
    const int tiles_x = (W + 63) / 64, tiles_total = tiles_x * ((H + 15) / 16);
    const dim3 tiles(8u * (unsigned)((tiles_total + 7) / 8)), rows((W + 63) / 64, (H + 3) / 4);
    // large images: bands of 64 columns cut into segments so that about eight workgroups stand on every CU
    WalkArgs wk;
    wk.bands = tiles_x;
    const int blocks = (H + WALK_BH - 1) / WALK_BH;
    wk.segs = std::max(1, std::min(blocks, (2048 + wk.bands - 1) / wk.bands));  // (1,024 to 8,192 workgroups measured alike)
    wk.seg_rows = WALK_BH * ((blocks + wk.segs - 1) / wk.segs);
    wk.segs = (H + wk.seg_rows - 1) / wk.seg_rows;
    const dim3 walkers(8u * (unsigned)((wk.bands * wk.segs + 7) / 8));
    const bool walk = MODE == 0 && n >= (size_t)WIDE_PIXELS;
#define PEDP_STENCIL(RV)                                                                                                                \
    if (walk) hipLaunchKernelGGL(erode_walk_kernel<RV>, walkers, dim3(256), 0, c->stream, d_in, d_out, p, wk);                            \
    else if (MODE == 0) hipLaunchKernelGGL((erode_kernel<RV, 64, 4>), tiles, dim3(256), 0, c->stream, d_in, d_out, p, tiles_x, tiles_total); \
    else hipLaunchKernelGGL((bilateral_kernel<RV, 64, 4>), tiles, dim3(256), 0, c->stream, d_in, d_out, p, tiles_x, tiles_total)
    switch (p.radius) {
        case 1: PEDP_STENCIL(1); break;
        case 2: PEDP_STENCIL(2); break;
        case 3: PEDP_STENCIL(3); break;
        case 4: PEDP_STENCIL(4); break;
        default:
            if (MODE == 0) hipLaunchKernelGGL(erode_generic_kernel, rows, dim3(256), 0, c->stream, d_in, d_out, p);
            else hipLaunchKernelGGL(bilateral_generic_kernel, rows, dim3(256), 0, c->stream, d_in, d_out, p);
    }
#undef PEDP_STENCIL
    return stage_out(c, out, d_out, n, mem);
}

}  // namespace

extern "C" {

int pedp_erode_depth(pedp_ctx_t c, const float *depth, int H, int W, int radius, float depth_diff_thres,
                     float ratio_thres, float zfar, int mem, float *out) {
    StencilArgs p{H, W, radius, zfar, depth_diff_thres, ratio_thres};
    return run_stencil<0>(c, depth, H, W, p, mem, out, "pedp_erode_depth");
}

int pedp_bilateral_filter_depth(pedp_ctx_t c, const float *depth, int H, int W, int radius, float zfar, float sigmaD,
                                float sigmaR, int mem, float *out) {
    StencilArgs p{H, W, radius, zfar, sigmaD, sigmaR};
    return run_stencil<1>(c, depth, H, W, p, mem, out, "pedp_bilateral_filter_depth");
}

int pedp_depth2xyzmap(pedp_ctx_t c, const float *depth, int H, int W, const double K[9], int mem, float *xyz) {
    int rc = check_image(c, depth, xyz, H, W, mem, "pedp_depth2xyzmap");
    if (rc) return rc;
    PEDP_REQUIRE(K, "pedp_depth2xyzmap: null intrinsics");
    const size_t n = (size_t)H * W;
    if (n == 0) return PEDP_OK;
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    const float *d_in;
    float *d_out;
    rc = stage_in(c, depth, n, 3 * n, mem, &d_in, &d_out, xyz);
    if (rc) return rc;
    hipLaunchKernelGGL(xyzmap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_in, H, W, K[0], K[4],
                       K[2], K[5], d_out);
    return stage_out(c, xyz, d_out, 3 * n, mem);
}

int pedp_depth2xyzmap_batch(pedp_ctx_t c, const float *depths, int B, int H, int W, const float *Ks, float zfar,
                            int mem, float *xyz) {
    int rc = check_image(c, depths, xyz, H, W, mem, "pedp_depth2xyzmap_batch");
    if (rc) return rc;
    PEDP_REQUIRE(B >= 0 && B < 65536 && (int64_t)B * H * W < (int64_t)1 << 33, "pedp_depth2xyzmap_batch: batch out of range");
    const size_t hw = (size_t)H * W, n = hw * (size_t)B;
    if (n == 0) return PEDP_OK;
    PEDP_REQUIRE(Ks, "pedp_depth2xyzmap_batch: null intrinsics");
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    const float *d_in;
    float *d_out;
    rc = stage_in(c, depths, n, 3 * n, mem, &d_in, &d_out, xyz);
    if (rc) return rc;
    // intrinsics: B x 9 float32 on the host in both modes (a few bytes per image)
    int st = c->proj.reserve(sizeof(float) * 9 * (size_t)B);
    if (st) return st;
    PEDP_HIP_CHECK(hipMemcpyAsync(c->proj.ptr, Ks, sizeof(float) * 9 * (size_t)B, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(xyzmap_batch_kernel, dim3((unsigned)((hw + 255) / 256), (unsigned)B), dim3(256), 0, c->stream, d_in,
                       (int64_t)hw, W, (const float *)c->proj.ptr, zfar, d_out);
    return stage_out(c, xyz, d_out, 3 * n, mem);
}

int pedp_depth_to_scene(pedp_ctx_t c, const float *depth, int H, int W, int depth_mem, const pedp_depth_entry_params *prm,
                        float *d_filtered, float *d_xyz, double *d_points, int64_t *n_points) {
    PEDP_REQUIRE(c && prm && d_points && n_points, "pedp_depth_to_scene: null argument");
    PEDP_REQUIRE(depth_mem == PEDP_HOST || depth_mem == PEDP_DEVICE, "pedp_depth_to_scene: bad mem flag %d", depth_mem);
    PEDP_REQUIRE(H >= 0 && W >= 0 && (int64_t)H * W < (int64_t)1 << 30, "pedp_depth_to_scene: image size out of range");
    *n_points = 0;
    const size_t n = (size_t)H * W;
    if (n == 0) return PEDP_OK;
    PEDP_REQUIRE(depth, "pedp_depth_to_scene: null depth image");
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    // scratch: [raw image (host input)][eroded][filtered (if the caller wants none)][xyz (likewise)][block offsets][counter][K]
    const int n_blocks = (int)((n + SC_BLOCK - 1) / SC_BLOCK);
    auto a256 = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t s_img = a256(sizeof(float) * n), s_xyz = a256(sizeof(float) * 3 * n), s_off = a256(sizeof(int) * ((size_t)n_blocks + 1));
    int st = c->ray_in.reserve(3 * s_img + s_xyz + s_off + 512);
    if (st) return st;
    char *b = (char *)c->ray_in.ptr;
    const float *d_raw = depth;
    if (depth_mem == PEDP_HOST) {
        { int up_ = pedp_upload(c, b, depth, sizeof(float) * n); if (up_) return up_; }
        d_raw = (const float *)b;
    }
    float *d_er = (float *)(b + s_img);
    float *d_fl = d_filtered ? d_filtered : (float *)(b + 2 * s_img);
    float *d_x = d_xyz ? d_xyz : (float *)(b + 3 * s_img);
    int *boff = (int *)(b + 3 * s_img + s_xyz);
    int *counter = boff + n_blocks;
    float *d_K = (float *)(b + 3 * s_img + s_xyz + s_off);
    int rc = pedp_erode_depth(c, d_raw, H, W, prm->erode_radius, prm->erode_diff, prm->erode_ratio, prm->erode_zfar, PEDP_DEVICE, d_er);
    if (rc) return rc;
    rc = pedp_bilateral_filter_depth(c, d_er, H, W, prm->bilateral_radius, prm->bilateral_zfar, prm->sigmaD, prm->sigmaR, PEDP_DEVICE, d_fl);
    if (rc) return rc;
    float *h_K = (float *)((char *)c->pinned + 12288);
    for (int k = 0; k < 9; ++k) h_K[k] = prm->K[k];   // (every earlier user of this pinned block has waited for its copy)
    PEDP_HIP_CHECK(hipMemcpyAsync(d_K, h_K, sizeof(float) * 9, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(xyzmap_batch_kernel, dim3((unsigned)((n + 255) / 256), 1u), dim3(256), 0, c->stream, (const float *)d_fl, (int64_t)n, W,
                       (const float *)d_K, prm->xyz_zfar, d_x);
    hipLaunchKernelGGL(scene_count_kernel, dim3((unsigned)n_blocks), dim3(SC_THREADS), 0, c->stream, (const float *)d_x, (int64_t)n, prm->z_min, boff);
    hipLaunchKernelGGL(scene_scan_kernel, dim3(1), dim3(1024), 0, c->stream, boff, n_blocks, counter);
    hipLaunchKernelGGL(scene_scatter_kernel, dim3((unsigned)n_blocks), dim3(SC_THREADS), 0, c->stream, (const float *)d_x, (int64_t)n, prm->z_min,
                       prm->scale, (const int *)boff, d_points);
    PEDP_HIP_CHECK(hipGetLastError());
    int *h_n = (int *)((char *)c->pinned + 8192);
    PEDP_HIP_CHECK(hipMemcpyAsync(h_n, counter, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->stage_busy = false;
    *n_points = *h_n;
    return PEDP_OK;
}

}  // extern "C"
