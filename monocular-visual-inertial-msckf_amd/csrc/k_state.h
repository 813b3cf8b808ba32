// f2 / f3 -- the covariance steps either side of the update, so that P stays in HBM
// between frames:
//   k_propagate        MSCKF.process_imu covariance half      (src/msckf/MSCKF.py:236-244)
//   k_augment          MSCKF.state_augmentation               (:258-265)
//   k_compact          MSCKF.remove_cameras                   (:751-757)
//   k_symmetrize_tail  the (P + P^T)/2 of :244 on the clone block
// All three are O(d^2) element-wise / 15-term dot-product kernels (d = 15 + 6N <= a few
// hundred): HBM-bound by definition and launch-latency bound in practice.
#pragma once
#include <hip/hip_runtime.h>

namespace msckf {

struct PropagateArgs {
    double* P;            // [d][d] in place
    int d;
    double Phi[225];      // 15x15 transition (observability-constrained, MSCKF.py:218-233)
    double Q[225];        // 15x15 discrete noise (:237)
};

// One workgroup.  T = Phi P[:15, :] lives in LDS (15 x d doubles) while the IMU rows and
// columns of P are rewritten.
__global__ __launch_bounds__(256) void k_propagate(PropagateArgs p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int d = p.d, tid = threadIdx.x;
    double* T = smem;                 // [15][d]
    double* A = T + 15 * d;           // [15][15]  Phi P_II Phi^T + Q before symmetrisation
    for (int idx = tid; idx < 15 * d; idx += 256) {
        const int i = idx / d, j = idx - i * d;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 15; ++k) s = fma(p.Phi[i * 15 + k], p.P[(size_t)k * d + j], s);
        T[idx] = s;
    }
    __syncthreads();
    if (tid < 225) {
        const int i = tid / 15, j = tid - i * 15;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 15; ++k) s = fma(T[i * d + k], p.Phi[j * 15 + k], s);
        A[tid] = s + p.Q[tid];                                           // :238
    }
    __syncthreads();
    if (tid < 225) {
        const int i = tid / 15, j = tid - i * 15;
        p.P[(size_t)i * d + j] = 0.5 * (A[i * 15 + j] + A[j * 15 + i]);  // :244 on the IMU block
    }
    for (int idx = tid; idx < 15 * (d - 15); idx += 256) {
        const int i = idx / (d - 15), j = 15 + idx - i * (d - 15);
        const double v = T[i * d + j];
        p.P[(size_t)i * d + j] = v;                                      // :241
        p.P[(size_t)j * d + i] = v;                                      // :242
    }
}

// (P + P^T)/2 on the rows/columns >= lo, in place (thread (i, j), i < j, owns the pair).
__global__ __launch_bounds__(256) void k_symmetrize_tail(double* P, int d, int lo) {
    const int j = lo + blockIdx.x * 16 + (threadIdx.x & 15);
    const int i = lo + blockIdx.y * 16 + (threadIdx.x >> 4);
    if (i < j && j < d) {
        const double v = 0.5 * (P[(size_t)i * d + j] + P[(size_t)j * d + i]);
        P[(size_t)i * d + j] = v;
        P[(size_t)j * d + i] = v;
    }
}

struct AugmentArgs {
    const double* P;      // [d][d]
    double* out;          // [d+6][d+6]
    int d;
    double J[90];         // 6x15: the non-zero columns of J (MSCKF.py:258-261)
};

// out = sym(M P M^T), M = [I; J]  (MSCKF.py:262-265).  One thread per output element.
__global__ __launch_bounds__(256) void k_augment(AugmentArgs p) {
    const int d = p.d, n = d + 6;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const int i = idx / n, j = idx - i * n;
    double v;
    if (i < d && j < d) {
        v = 0.5 * (p.P[(size_t)i * d + j] + p.P[(size_t)j * d + i]);
    } else if (i >= d && j >= d) {
        // S[d+a][d+b] = sum_k (J P)[a][k] J[b][k]; both orders for the symmetrisation
        const int a = i - d, b = j - d;
        double sab = 0.0, sba = 0.0;
        for (int k = 0; k < 15; ++k) {
            double ja = 0.0, jb = 0.0;
            for (int l = 0; l < 15; ++l) {
                ja = fma(p.J[a * 15 + l], p.P[(size_t)l * d + k], ja);
                jb = fma(p.J[b * 15 + l], p.P[(size_t)l * d + k], jb);
            }
            sab = fma(ja, p.J[b * 15 + k], sab);
            sba = fma(jb, p.J[a * 15 + k], sba);
        }
        v = 0.5 * (sab + sba);
    } else {
        const int a = (i >= d) ? i - d : j - d;       // new clone row
        const int c = (i >= d) ? j : i;               // old state column
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < 15; ++k) {
            s1 = fma(p.J[a * 15 + k], p.P[(size_t)k * d + c], s1);      // (J P)[a][c]
            s2 = fma(p.P[(size_t)c * d + k], p.J[a * 15 + k], s2);      // (P J^T)[c][a]
        }
        v = 0.5 * (s1 + s2);
    }
    p.out[(size_t)i * n + j] = v;
}

// out[i][j] = P[keep[i]][keep[j]]: rows and columns of removed clones dropped (MSCKF.py:754-757).
__global__ __launch_bounds__(256) void k_compact(const double* P, int d, const int* keep, int n, double* out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const int i = idx / n, j = idx - i * n;
    out[idx] = P[(size_t)keep[i] * d + keep[j]];
}

}  // namespace msckf
