/* TEST INFRASTRUCTURE -- CPU restatement of the reference's depth pre-filters, one scalar loop
 * nest per pixel in the reference's own order.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it.
 *
 *   pedp_oracle_erode_depth      Utils.py:356-383 (warp kernel erode_depth_kernel)
 *   pedp_oracle_bilateral_depth  Utils.py:304-345 (warp kernel bilateral_filter_depth_kernel)
 *   pedp_oracle_depth2xyzmap     Utils.py:401-420 (numpy; uvs=None)
 *   pedp_oracle_depth2xyzmap_batch Utils.py:423-442 (torch float32)
 *
 * Parity unpinned: warp-lang is absent and CUDA-only, the reference holds no fixtures for these
 * kernels.  warp's `float` is float32 and its literals are float32 constants; whether its CUDA
 * build contracts a*b+c into FMA is not recoverable from the reference, so this restatement (and
 * the HIP kernels) fix: no contraction, libm expf.  Compiled with -ffp-contract=off.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include "pedp_oracle.h"

void pedp_oracle_erode_depth(const float *depth, int H, int W, int radius, float depth_diff_thres, float ratio_thres,
                             float zfar, float *out, int nthreads) {
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
            const float d_ori = depth[(size_t)h * W + w];
            float res = 0.0f; /* out is zero-initialised (wp.zeros) */
            if (d_ori < 0.001f || d_ori >= zfar) res = 0.0f;
            float bad_cnt = 0.0f, total = 0.0f;
            for (int u = w - radius; u <= w + radius; ++u) {
                if (u < 0 || u >= W) continue;
                for (int v = h - radius; v <= h + radius; ++v) {
                    if (v < 0 || v >= H) continue;
                    const float cur = depth[(size_t)v * W + u];
                    total += 1.0f;
                    if (cur < 0.001f || cur >= zfar || fabsf(cur - d_ori) > depth_diff_thres) bad_cnt += 1.0f;
                }
            }
            if (bad_cnt / total > ratio_thres) res = 0.0f;
            else res = d_ori;
            out[(size_t)h * W + w] = res;
        }
}

void pedp_oracle_bilateral_depth(const float *depth, int H, int W, int radius, float zfar, float sigmaD, float sigmaR,
                                 float *out, int nthreads) {
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
            float *o = &out[(size_t)h * W + w];
            *o = 0.0f;
            float mean_depth = 0.0f;
            int num_valid = 0;
            for (int u = w - radius; u <= w + radius; ++u) {
                if (u < 0 || u >= W) continue;
                for (int v = h - radius; v <= h + radius; ++v) {
                    if (v < 0 || v >= H) continue;
                    const float cur = depth[(size_t)v * W + u];
                    if (cur >= 0.001f && cur < zfar) { num_valid += 1; mean_depth += cur; }
                }
            }
            if (num_valid == 0) continue;
            mean_depth /= (float)num_valid;
            const float centre = depth[(size_t)h * W + w];
            float sum_weight = 0.0f, sum = 0.0f;
            for (int u = w - radius; u <= w + radius; ++u) {
                if (u < 0 || u >= W) continue;
                for (int v = h - radius; v <= h + radius; ++v) {
                    if (v < 0 || v >= H) continue;
                    const float cur = depth[(size_t)v * W + u];
                    if (cur >= 0.001f && cur < zfar && fabsf(cur - mean_depth) < 0.01f) {
                        const float a = -(float)((u - w) * (u - w) + (h - v) * (h - v)) / (2.0f * sigmaD * sigmaD);
                        const float b = (centre - cur) * (centre - cur) / (2.0f * sigmaR * sigmaR);
                        const float weight = expf(a - b);
                        sum_weight += weight;
                        sum += weight * cur;
                    }
                }
            }
            if (sum_weight > 0.0f && num_valid > 0) *o = sum / sum_weight;
        }
}

void pedp_oracle_depth2xyzmap(const float *depth, int H, int W, const double *K, float *xyz) {
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) {
            const size_t i = (size_t)v * W + u;
            const float z = depth[i];
            float *o = xyz + 3 * i;
            if (z < 0.001f) { o[0] = o[1] = o[2] = 0.0f; continue; }
            o[0] = (float)(((double)u - K[2]) * (double)z / K[0]);
            o[1] = (float)(((double)v - K[5]) * (double)z / K[4]);
            o[2] = z;
        }
}

void pedp_oracle_depth2xyzmap_batch(const float *depths, int B, int H, int W, const float *Ks, float zfar, float *xyz) {
    for (int b = 0; b < B; ++b)
        for (int v = 0; v < H; ++v)
            for (int u = 0; u < W; ++u) {
                const size_t i = ((size_t)b * H + v) * W + u;
                const float *K = Ks + 9 * b;
                const float z = depths[i];
                float *o = xyz + 3 * i;
                if ((z < 0.001f) || (z > zfar)) { o[0] = o[1] = o[2] = 0.0f; continue; }
                o[0] = ((float)u - K[2]) * z / K[0];
                o[1] = ((float)v - K[5]) * z / K[4];
                o[2] = z;
            }
}
