// f4 -- the geometric consistency tests of the front end (reference MSCKF.add_camera_measurements,
// src/msckf/MSCKF.py:332-412): a descriptor match (tracked feature <-> keypoint of the newest image) is kept only
// if it agrees with EVERY earlier view of the feature,
//   * |t_12| >= 0.01 : epipolar test   score = [x2,1]^T F [x1,1],  F = K^-T [t_12]x R_12 K^-1      (:392-394), signed
//   * |t_12| <  0.01 : homography test H = K R_12 K^-1, mean of the two transfer errors          (:370-376)
// with T_12 = T_W_C1^-1 T_W_C2 (C1 = the earlier view's clone, C2 = the newest clone, :366).  One lane per
// match walks the feature's views in order and stops at the first failure like the reference's `break`.
// HBM-bound by definition: 20 bytes per (match, view) in, 5 bytes per match out.
#pragma once
#include <hip/hip_runtime.h>

namespace msckf {

struct AssocArgs {
    int F;
    const int* view_ptr;        // [F+1] sorted feature order (the tracks BEFORE the new view)
    const double* obs_uv;       // [sumM*2] pixels
    const int* obs_slot;        // [sumM]
    const double* cam_R;        // [N*9]
    const double* cam_t;        // [N*3]
    const double* matched_uv;   // [F*2] keypoint matched to the feature in the newest image (NaN: no match), sorted order
    double K[9], Kinv[9];
    double R2[9], t2[3];        // pose of the newest clone (T_W_C2)
    double thr_epipolar, thr_homography;
    unsigned char* result;      // [F] 0 kept, 1 epipolar failure, 2 homography failure, 3 no match
    int* fail_view;             // [F] view that failed, -1
};

__device__ __forceinline__ void mat3_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

__global__ __launch_bounds__(256) void k_assoc(AssocArgs p) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= p.F) return;
    const double mx = p.matched_uv[2 * f], my = p.matched_uv[2 * f + 1];
    if (!(mx == mx) || !(my == my)) { p.result[f] = 3; p.fail_view[f] = -1; return; }
    unsigned char res = 0;
    int fv = -1;
    const int v0 = p.view_ptr[f], v1 = p.view_ptr[f + 1];
    for (int o = v0; o < v1 && res == 0; ++o) {
        const int s = p.obs_slot[o];
        const double* R1 = p.cam_R + 9 * s;
        const double* t1 = p.cam_t + 3 * s;
        const double fx = p.obs_uv[2 * o], fy = p.obs_uv[2 * o + 1];
        // T_12 = T_W_C1^-1 T_W_C2: R12 = R1^T R2, t12 = R1^T (t2 - t1)
        double R12[9], t12[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) R12[3 * i + j] = R1[i] * p.R2[j] + R1[3 + i] * p.R2[3 + j] + R1[6 + i] * p.R2[6 + j];
            t12[i] = R1[i] * (p.t2[0] - t1[0]) + R1[3 + i] * (p.t2[1] - t1[1]) + R1[6 + i] * (p.t2[2] - t1[2]);
        }
        const double nt = sqrt(t12[0] * t12[0] + t12[1] * t12[1] + t12[2] * t12[2]);
        if (nt < 0.01) {
            double KR[9], H[9], KRt[9], Hi[9], R12t[9];
            mat3_mul(p.K, R12, KR);
            mat3_mul(KR, p.Kinv, H);                                   // H = K R12 K^-1
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) R12t[3 * i + j] = R12[3 * j + i];
            mat3_mul(p.K, R12t, KRt);
            mat3_mul(KRt, p.Kinv, Hi);                                 // H^-1 = K R12^T K^-1
            const double a0 = Hi[0] * mx + Hi[1] * my + Hi[2], a1 = Hi[3] * mx + Hi[4] * my + Hi[5], a2 = Hi[6] * mx + Hi[7] * my + Hi[8];
            const double b0 = H[0] * fx + H[1] * fy + H[2], b1 = H[3] * fx + H[4] * fy + H[5], b2 = H[6] * fx + H[7] * fy + H[8];
            const double e1x = mx - a0 / a2, e1y = my - a1 / a2, e2x = fx - b0 / b2, e2y = fy - b1 / b2;
            const double score = 0.5 * (sqrt(e1x * e1x + e1y * e1y) + sqrt(e2x * e2x + e2y * e2y));
            if (score > p.thr_homography) { res = 2; fv = o - v0; }
        } else {
            // F = K^-T [t12]x R12 K^-1;  score = m^T F x1 = (K^-1 m)^T [t12]x R12 (K^-1 x1)
            const double* Ki = p.Kinv;
            const double m0 = Ki[0] * mx + Ki[1] * my + Ki[2], m1 = Ki[3] * mx + Ki[4] * my + Ki[5], m2 = Ki[6] * mx + Ki[7] * my + Ki[8];
            const double x0 = Ki[0] * fx + Ki[1] * fy + Ki[2], x1 = Ki[3] * fx + Ki[4] * fy + Ki[5], x2 = Ki[6] * fx + Ki[7] * fy + Ki[8];
            const double y0 = R12[0] * x0 + R12[1] * x1 + R12[2] * x2, y1 = R12[3] * x0 + R12[4] * x1 + R12[5] * x2,
                         y2 = R12[6] * x0 + R12[7] * x1 + R12[8] * x2;
            const double c0 = t12[1] * y2 - t12[2] * y1, c1 = t12[2] * y0 - t12[0] * y2, c2 = t12[0] * y1 - t12[1] * y0;   // t12 x y
            const double score = m0 * c0 + m1 * c1 + m2 * c2;
            if (score > p.thr_epipolar) { res = 1; fv = o - v0; }
        }
    }
    p.result[f] = res;
    p.fail_view[f] = fv;
}

}  // namespace msckf
