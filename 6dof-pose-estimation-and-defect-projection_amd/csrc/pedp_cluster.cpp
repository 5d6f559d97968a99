// cluster_poses for libpedp_hip.so: host-only, float32.
//
// Replaces mycpp.cluster_poses (mycpp/src/app/pybind_api.cpp:24-68, geodesic distance
// mycpp/src/Utils.cpp:21-26; caller estimater.py:118 with 252 poses) without Eigen /
// Boost / pybind11.  Sequential greedy de-duplication, O(N * kept * S) 3x3 products:
// microseconds for the rotation grid, so no GPU kernel.
#include <cmath>
#include <cstdint>
#include "../../include/pedp.h"

void pedp_set_error(const char *fmt, ...);

namespace {

struct Rot {
    float m[9];
};

inline Rot rot_of(const float *pose) {
    Rot r;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) r.m[3 * a + b] = pose[4 * a + b];
    return r;
}

// rotation block of (pose * tf), full 4x4 row-by-column product like Eigen's Matrix4f
inline Rot rot_of_product(const float *pose, const float *tf) {
    Rot r;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            r.m[3 * a + b] = ((pose[4 * a] * tf[b] + pose[4 * a + 1] * tf[4 + b]) +
                              pose[4 * a + 2] * tf[8 + b]) + pose[4 * a + 3] * tf[12 + b];
    return r;
}

inline float geodesic(const Rot &a, const Rot &b) {
    float tr = 0.0f;
    for (int i = 0; i < 3; ++i) {
        float d = (a.m[3 * i] * b.m[3 * i] + a.m[3 * i + 1] * b.m[3 * i + 1]) + a.m[3 * i + 2] * b.m[3 * i + 2];
        tr = (i == 0) ? d : tr + d;
    }
    float c = (float)(((double)(tr - 1.0f)) / 2.0);
    c = std::fmax(std::fmin(c, 1.0f), -1.0f);
    return std::acos(c);
}

}  // namespace

extern "C" int pedp_cluster_poses(float angle_diff_deg, float dist_diff, const float *poses, int n,
                                  const float *syms, int s, int32_t *keep_idx, int *n_keep) {
    if (!n_keep || n < 0 || s < 0 || (n > 0 && (!poses || !keep_idx)) || (s > 0 && !syms)) {
        pedp_set_error("pedp_cluster_poses: bad arguments (n=%d, s=%d)", n, s);
        return PEDP_ERR_BAD_ARG;
    }
    *n_keep = 0;
    if (n == 0) return PEDP_OK;
    const float radian_thres = (float)((double)angle_diff_deg / 180.0 * M_PI);
    int nk = 0;
    keep_idx[nk++] = 0;
    for (int i = 1; i < n; ++i) {
        const float *cur = poses + 16 * i;
        bool isnew = true;
        for (int c = 0; c < nk && isnew; ++c) {
            const float *cl = poses + 16 * keep_idx[c];
            const float dx = cl[3] - cur[3], dy = cl[7] - cur[7], dz = cl[11] - cur[11];
            if (std::sqrt((dx * dx + dy * dy) + dz * dz) >= dist_diff) continue;
            const Rot rc = rot_of(cl);
            for (int k = 0; k < s; ++k) {
                if (geodesic(rot_of_product(cur, syms + 16 * k), rc) < radian_thres) {
                    isnew = false;
                    break;
                }
            }
        }
        if (isnew) keep_idx[nk++] = i;
    }
    *n_keep = nk;
    return PEDP_OK;
}

// Rigid transform of a host array (include/pedp.h): the holders' transform().  -ffp-contract=off: no fused multiply-add.
extern "C" int pedp_transform_points(const double T[16], const double *in, int64_t n, int rotate_only, double *out) {
    if (!T || n < 0 || (n > 0 && (!in || !out))) {
        pedp_set_error("pedp_transform_points: bad arguments (n=%lld)", (long long)n);
        return PEDP_ERR_BAD_ARG;
    }
    const double t0 = rotate_only ? 0.0 : T[3], t1 = rotate_only ? 0.0 : T[7], t2 = rotate_only ? 0.0 : T[11];
    if (rotate_only) {
        for (int64_t i = 0; i < n; ++i) {
            const double x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
            out[3 * i] = (T[0] * x + T[1] * y) + T[2] * z;
            out[3 * i + 1] = (T[4] * x + T[5] * y) + T[6] * z;
            out[3 * i + 2] = (T[8] * x + T[9] * y) + T[10] * z;
        }
        return PEDP_OK;
    }
    for (int64_t i = 0; i < n; ++i) {
        const double x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
        out[3 * i] = ((T[0] * x + T[1] * y) + T[2] * z) + t0;
        out[3 * i + 1] = ((T[4] * x + T[5] * y) + T[6] * z) + t1;
        out[3 * i + 2] = ((T[8] * x + T[9] * y) + T[10] * z) + t2;
    }
    return PEDP_OK;
}
