// f1 -- feature selection and triangulation, the step in front of K1.
// Follows reference MSCKF.get_valid_features (src/msckf/MSCKF.py:458-495):
//   lost / too-short / parallax tests, intersection_of_lines (src/utils/geometry.py:274-303),
//   re-projection into the anchor clone (src/msckf/Camera.py:13-52) and the refresh of the
//   inverse-depth point (geometry.py:61-71).
// Eight lanes share one feature: each lane folds the lines i = sub, sub+8, ... into the 3x3
// normal matrix X and the right-hand side y, an 8-lane DPP sum combines them, and every lane
// then runs the same 3x3 Jacobi eigen-solve for pinv(X) (numpy's rcond = 1e-15 cut-off).
// HBM traffic: 7 doubles + 1 int per view in, 8 doubles + 1 byte per feature out; the kernel
// is launch-latency bound at MSCKF sizes (about 1 MB at 2000 x 10 views).
#pragma once
#include <hip/hip_runtime.h>

#include "wave_ops.h"

namespace msckf {

enum : unsigned char { SEL_VALID = 1, SEL_LOST = 2, SEL_REFRESHED = 4 };

struct SelectArgs {
    int F;
    const int* view_ptr;          // [F+1] CSR (sorted feature order)
    const int* obs_slot;          // [sumM] clone slot per view
    const double* line_base;      // [sumM*3]
    const double* line_dir;       // [sumM*3]
    const double* line_conf;      // [sumM]
    const int* lost_for;          // [F]
    const int* tracked_for;       // [F]
    const double* cam_R;          // [N*9]
    const double* cam_t;          // [N*3]
    double K[9], Kinv[9];
    int width, height, min_lost, min_tracked, use_parallax;
    double min_parallax_deg;
    unsigned char* flags;         // [F]
    double* idp_m;                // [F*3] refreshed in place
    double* idp_rho;              // [F]
    double* world;                // [F*3] triangulated point (NaN when none was computed)
};

// One Jacobi rotation annihilating a[p][q] of a symmetric 3x3 matrix (r is the third index).
#define MSCKF_JACOBI_ROT(app, aqq, apq, arp, arq, v0p, v0q, v1p, v1q, v2p, v2q)            \
    do {                                                                                    \
        if (apq != 0.0) {                                                                   \
            const double th_ = (aqq - app) / (2.0 * apq);                                   \
            const double t_ = (th_ >= 0.0 ? 1.0 : -1.0) / (fabs(th_) + sqrt(th_ * th_ + 1.0)); \
            const double c_ = 1.0 / sqrt(t_ * t_ + 1.0), s_ = t_ * c_;                      \
            app -= t_ * apq;                                                                \
            aqq += t_ * apq;                                                                \
            apq = 0.0;                                                                      \
            const double rp_ = arp, rq_ = arq;                                              \
            arp = c_ * rp_ - s_ * rq_;                                                      \
            arq = s_ * rp_ + c_ * rq_;                                                      \
            double x_, y_;                                                                  \
            x_ = v0p; y_ = v0q; v0p = c_ * x_ - s_ * y_; v0q = s_ * x_ + c_ * y_;           \
            x_ = v1p; y_ = v1q; v1p = c_ * x_ - s_ * y_; v1q = s_ * x_ + c_ * y_;           \
            x_ = v2p; y_ = v2q; v2p = c_ * x_ - s_ * y_; v2q = s_ * x_ + c_ * y_;           \
        }                                                                                   \
    } while (0)

__global__ __launch_bounds__(256) void k_select(SelectArgs p) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int fr = gid >> 3, sub = threadIdx.x & 7;
    const bool live = fr < p.F;
    const int f = live ? fr : p.F - 1;            // idle groups shadow the last feature (DPP needs all lanes)
    const int v0 = p.view_ptr[f], v1 = p.view_ptr[f + 1];
    const int M = v1 - v0;

    const bool lost = p.lost_for[f] >= p.min_lost;                          // MSCKF.py:463-465
    const bool too_short = lost && p.tracked_for[f] < p.min_tracked;        // :467-469
    bool enough = false;
    if (p.use_parallax && M > 1) {                                          // :472-477
        const double* a = p.line_dir + (size_t)v0 * 3;
        const double* b = p.line_dir + (size_t)(v1 - 1) * 3;
        const double na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        const double nb = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
        double dot = (a[0] / na) * (b[0] / nb) + (a[1] / na) * (b[1] / nb) + (a[2] / na) * (b[2] / nb);
        dot = fmin(1.0, fmax(-1.0, dot));                                   // geometry.py:249-253
        enough = acos(dot) * (180.0 / 3.14159265358979323846) > p.min_parallax_deg;
    }
    const bool tri = !too_short && (lost || enough);                        // :479

    // X = sum c (I - d d^T), y = sum c (I - d d^T) base          (geometry.py:285-297)
    double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0, y0 = 0, y1 = 0, y2 = 0;
    if (tri) {
        for (int i = v0 + sub; i < v1; i += 8) {
            const double* dp = p.line_dir + (size_t)i * 3;
            const double* bp = p.line_base + (size_t)i * 3;
            const double c = p.line_conf[i];
            const double n = sqrt(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2]);
            const double dx = dp[0] / n, dy = dp[1] / n, dz = dp[2] / n;
            const double db = dx * bp[0] + dy * bp[1] + dz * bp[2];
            xx += c * (1.0 - dx * dx); yy += c * (1.0 - dy * dy); zz += c * (1.0 - dz * dz);
            xy -= c * dx * dy; xz -= c * dx * dz; yz -= c * dy * dz;
            y0 += c * (bp[0] - dx * db); y1 += c * (bp[1] - dy * db); y2 += c * (bp[2] - dz * db);
        }
    }
    xx = row8_sum(xx); xy = row8_sum(xy); xz = row8_sum(xz);
    yy = row8_sum(yy); yz = row8_sum(yz); zz = row8_sum(zz);
    y0 = row8_sum(y0); y1 = row8_sum(y1); y2 = row8_sum(y2);
    if (!live || sub != 0) return;

    unsigned char flag = 0;
    double wx = __builtin_nan(""), wy = wx, wz = wx;
    if (too_short) {
        flag = SEL_LOST;
    } else if (tri) {
        flag = SEL_VALID | (lost ? SEL_LOST : 0);                           // :492-493
        // pinv(X) y through the eigen-decomposition X = V diag(l) V^T (cyclic Jacobi)
        double a00 = xx, a11 = yy, a22 = zz, a01 = xy, a02 = xz, a12 = yz;
        double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
        for (int sweep = 0; sweep < 12; ++sweep) {
            const double off = a01 * a01 + a02 * a02 + a12 * a12;
            if (off <= 1e-60 * (a00 * a00 + a11 * a11 + a22 * a22)) break;
            MSCKF_JACOBI_ROT(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21);
            MSCKF_JACOBI_ROT(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22);
            MSCKF_JACOBI_ROT(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22);
        }
        const double lmax = fmax(fabs(a00), fmax(fabs(a11), fabs(a22)));
        const double cut = 1e-15 * lmax;                                    // numpy.linalg.pinv rcond
        const double c0 = (fabs(a00) > cut) ? (v00 * y0 + v10 * y1 + v20 * y2) / a00 : 0.0;
        const double c1 = (fabs(a11) > cut) ? (v01 * y0 + v11 * y1 + v21 * y2) / a11 : 0.0;
        const double c2 = (fabs(a22) > cut) ? (v02 * y0 + v12 * y1 + v22 * y2) / a22 : 0.0;
        wx = v00 * c0 + v01 * c1 + v02 * c2;
        wy = v10 * c0 + v11 * c1 + v12 * c2;
        wz = v20 * c0 + v21 * c1 + v22 * c2;
        // anchor clone = clone of the first view (:481); W2Ci through the explicit inverse (geometry.py:35-37)
        const int s = p.obs_slot[v0];
        const double* R = p.cam_R + (size_t)s * 9;
        const double* t = p.cam_t + (size_t)s * 3;
        const double i00 = R[4] * R[8] - R[5] * R[7], i01 = R[2] * R[7] - R[1] * R[8], i02 = R[1] * R[5] - R[2] * R[4];
        const double i10 = R[5] * R[6] - R[3] * R[8], i11 = R[0] * R[8] - R[2] * R[6], i12 = R[2] * R[3] - R[0] * R[5];
        const double i20 = R[3] * R[7] - R[4] * R[6], i21 = R[1] * R[6] - R[0] * R[7], i22 = R[0] * R[4] - R[1] * R[3];
        const double det = R[0] * i00 + R[1] * i10 + R[2] * i20;
        const double qx = wx - t[0], qy = wy - t[1], qz = wz - t[2];
        const double cx = (i00 * qx + i01 * qy + i02 * qz) / det;
        const double cy = (i10 * qx + i11 * qy + i12 * qz) / det;
        const double cz = (i20 * qx + i21 * qy + i22 * qz) / det;
        if (cz > 0.0) {                                                     // Camera.py:18
            const double hx = p.K[0] * cx + p.K[1] * cy + p.K[2] * cz;
            const double hy = p.K[3] * cx + p.K[4] * cy + p.K[5] * cz;
            const double hz = p.K[6] * cx + p.K[7] * cy + p.K[8] * cz;
            const double u = hx / hz, v = hy / hz;                          // Camera.py:20-21
            if (!(u < 0.0 || u >= (double)p.width || v < 0.0 || v >= (double)p.height)) {   // :24-26
                const double ex = p.Kinv[0] * u + p.Kinv[1] * v + p.Kinv[2];                // MSCKF.py:486
                const double ey = p.Kinv[3] * u + p.Kinv[4] * v + p.Kinv[5];
                const double ez = p.Kinv[6] * u + p.Kinv[7] * v + p.Kinv[8];
                const double gx = R[0] * ex + R[1] * ey + R[2] * ez;                        // :487
                const double gy = R[3] * ex + R[4] * ey + R[5] * ez;
                const double gz = R[6] * ex + R[7] * ey + R[8] * ez;
                // m = (cos phi sin theta, -sin phi, cos phi cos theta) = W_v / |W_v|  (geometry.py:64-67)
                const double gn = sqrt(gx * gx + gy * gy + gz * gz);
                p.idp_m[(size_t)f * 3 + 0] = gx / gn;
                p.idp_m[(size_t)f * 3 + 1] = gy / gn;
                p.idp_m[(size_t)f * 3 + 2] = gz / gn;
                p.idp_rho[f] = 1.0 / cz;                                    // geometry.py:61-62
                flag |= SEL_REFRESHED;
            }
        }
    }
    p.flags[f] = flag;
    p.world[(size_t)f * 3 + 0] = wx;
    p.world[(size_t)f * 3 + 1] = wy;
    p.world[(size_t)f * 3 + 2] = wz;
}

#undef MSCKF_JACOBI_ROT

}  // namespace msckf
