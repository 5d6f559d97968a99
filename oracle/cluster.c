/*
 * cluster.c -- ORACLE (test infrastructure, see pedp_oracle.h): float32 restatement
 * of mycpp.cluster_poses (mycpp/src/app/pybind_api.cpp:24-68) and
 * Utils::rotationGeodesicDistance (mycpp/src/Utils.cpp:21-26): greedy pose
 * de-duplication.  Pose 0 is always kept; a later pose is dropped if, for some kept
 * pose, the translations are closer than dist_diff AND for some symmetry transform
 * the geodesic rotation distance is below angle_diff degrees.
 * mycpp itself needs Eigen3 + Boost (mycpp/CMakeLists.txt:10-13), both absent here,
 * so it cannot be compiled as oracle/_ref.
 */
#include "pedp_oracle.h"
#include <math.h>

static float geodesic(const float R1[9], const float R2[9]) {
    /* ((R1 * R2^T).trace() - 1) / 2.0, clamped, acos */
    float tr = 0.0f;
    for (int i = 0; i < 3; ++i) {
        float d = (R1[3 * i] * R2[3 * i] + R1[3 * i + 1] * R2[3 * i + 1]) + R1[3 * i + 2] * R2[3 * i + 2];
        tr = (i == 0) ? d : tr + d;
    }
    float c = (float)(((double)(tr - 1.0f)) / 2.0);
    c = fmaxf(fminf(c, 1.0f), -1.0f);
    return acosf(c);
}

int pedp_oracle_cluster_poses(float angle_diff_deg, float dist_diff, const float *poses, int n,
                              const float *syms, int s, int32_t *keep_idx, int *n_keep) {
    if (n <= 0) { *n_keep = 0; return n == 0 ? 0 : -1; }
    const float radian_thres = (float)((double)angle_diff_deg / 180.0 * M_PI);
    int nk = 0;
    keep_idx[nk++] = 0;
    for (int i = 1; i < n; ++i) {
        const float *cur = poses + 16 * i;
        int isnew = 1;
        for (int c = 0; c < nk && isnew; ++c) {
            const float *cl = poses + 16 * keep_idx[c];
            float dx = cl[3] - cur[3], dy = cl[7] - cur[7], dz = cl[11] - cur[11];
            float nrm = sqrtf((dx * dx + dy * dy) + dz * dz);
            if (nrm >= dist_diff) continue;
            for (int k = 0; k < s; ++k) {
                const float *tf = syms + 16 * k;
                float R[9], Rc[9];
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) {
                        /* rotation block of cur * tf (4x4 product, k-ordered sum) */
                        R[3 * a + b] = ((cur[4 * a] * tf[b] + cur[4 * a + 1] * tf[4 + b]) +
                                        cur[4 * a + 2] * tf[8 + b]) + cur[4 * a + 3] * tf[12 + b];
                        Rc[3 * a + b] = cl[4 * a + b];
                    }
                if (geodesic(R, Rc) < radian_thres) { isnew = 0; break; }
            }
        }
        if (isnew) keep_idx[nk++] = i;
    }
    *n_keep = nk;
    return 0;
}
