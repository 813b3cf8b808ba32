// K6-K7 as separate launches behind K5 (rounds 1-3): Kalman gain and Joseph-form covariance update.  Since round 4 the
// update runs as k_gstream.h's sequential block update beside the root sweep; what is still used of this file: the in-wave
// elimination k_chol16 lent k_gstream.h, and the whole chain for windows of more than 82 clones, MSCKF_GAIN_STREAM=0 and the
// retry of a timed-out fused launch (msckf_get_result).
//   reference MSCKF.py:604-607 : S = T P T^T + R_n ; K = P T^T S^-1 ; dx = K r_n
//   reference MSCKF.py:612-614 : P+ = (I-KT) P (I-KT)^T + K R_n K^T ; P+ <- (P+ + P+^T)/2
// with R_n = sigma^2 I (Q^T (sigma^2 I) Q, MSCKF.py:598) and T = [0 | R] acting
// on the 6N clone columns only (T_H[:, :15] == 0).
//
// Dense d x d products run on the FP64 matrix cores (v_mfma_f64_16x16x4_f64),
// one wavefront per 16x16 output tile.  S is factored by a single workgroup in
// LDS (S = L L^T, SPD thanks to + sigma^2 I) instead of the reference's explicit
// inverse; K = Y S^-1 is two triangular sweeps per row, one wavefront per row.
#pragma once
#include <hip/hip_runtime.h>
#include "wave_ops.h"

namespace msckf {

typedef double v4d __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const double* A; int lda;     // A[M][K]
    const double* B; int ldb;     // transB ? B[N][K] : B[K][N]
    const double* C0; int ldc0;   // optional addend
    double* C; int ldc;
    int M, N, K;
    double alpha, beta, diag_add; // C = alpha*A*op(B) + beta*C0 + diag_add*I
    int transB;
    int tri;                      // 1: B[N][K] is upper triangular (k >= n), transB only
                                  // 2: A[M][K] is upper triangular (k >= m)
};

// One workgroup of GEMM_WAVES wavefronts per 16x16 tile of C: the K range is cut into GEMM_WAVES * 4 contiguous
// pieces -- one per wavefront and 16-lane group of the MFMA (the sum over k is order free) -- so a lane streams
// contiguous doubles of its A row (and of its B row when transB), and a 180-deep product is 12 dependent MFMAs per
// wavefront instead of 45.  The partial tiles meet in LDS; wavefront w finishes rows {lane >> 4 + 4 w}.
#ifndef MSCKF_GEMM_WAVES
#define MSCKF_GEMM_WAVES 8
#endif
constexpr int GEMM_WAVES = MSCKF_GEMM_WAVES;   // 4 or 8
__global__ __launch_bounds__(64 * GEMM_WAVES) void k_gemm_f64(GemmArgs g) {
    __shared__ double sAcc[GEMM_WAVES][4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const int r = lane & 15, grp = wv * 4 + (lane >> 4);
    // the addend of this wavefront's share of the epilogue: requested first, used last
    const int ocol = n0 + (lane & 15), orow = m0 + (lane >> 4) + 4 * wv;
    const bool ook = wv < 4 && orow < g.M && ocol < g.N;              // (the first four wavefronts finish the tile)
    double c0 = 0.0;
    if (g.C0 && ook) c0 = g.C0[(size_t)orow * g.ldc0 + ocol];
    int kbeg = 0;
    if (g.tri == 1) kbeg = n0;
    else if (g.tri == 2) kbeg = m0;
    const int klen = g.K - kbeg;
    const int kq = (klen + 4 * GEMM_WAVES - 1) / (4 * GEMM_WAVES);   // MFMA steps per wavefront
    // wavefront w covers k in [kbeg + 4 kq w, + 4 kq); at step u its four 16-lane groups take k = .. + 4 u + group, so one
    // load instruction touches 16 rows x 32 contiguous bytes (round 2: every group streamed its own piece, 64 different
    // cache lines per instruction -- the products were bound by the texture addresser, not by the matrix cores)
    const int k_lo = kbeg + wv * 4 * kq + (lane >> 4);
    const int k_hi = g.K;
    (void)grp;
    const int arow = m0 + r, bcol = n0 + r;
    const bool aok = arow < g.M, bok = bcol < g.N;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    // GEMM_TRIP k-steps per trip, all loads of a trip issued before its first MFMA (K <= 256 is one trip)
    constexpr int GEMM_TRIP = 16;
    if (g.transB) {
        for (int s0 = 0; s0 < kq; s0 += GEMM_TRIP) {
            double av[GEMM_TRIP], bv[GEMM_TRIP];
#pragma unroll
            for (int u = 0; u < GEMM_TRIP; ++u) {
                const int k = k_lo + 4 * (s0 + u);
                const bool kok = (s0 + u < kq) && (k < k_hi);
                av[u] = (aok && kok) ? g.A[(size_t)arow * g.lda + k] : 0.0;
                bv[u] = (bok && kok) ? g.B[(size_t)bcol * g.ldb + k] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < GEMM_TRIP; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
        }
    } else {
        for (int s0 = 0; s0 < kq; s0 += GEMM_TRIP) {
            double av[GEMM_TRIP], bv[GEMM_TRIP];
#pragma unroll
            for (int u = 0; u < GEMM_TRIP; ++u) {
                const int k = k_lo + 4 * (s0 + u);
                const bool kok = (s0 + u < kq) && (k < k_hi);
                av[u] = (aok && kok) ? g.A[(size_t)arow * g.lda + k] : 0.0;
                bv[u] = (bok && kok) ? g.B[(size_t)k * g.ldb + bcol] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < GEMM_TRIP; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
        }
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int i = 0; i < 4; ++i) sAcc[wv][i][lane] = acc[i];
    __syncthreads();
    if (ook) {
        double x = 0.0;
#pragma unroll
        for (int w = 0; w < GEMM_WAVES; ++w) x += sAcc[w][wv & 3][lane];   // register wv of every partial tile: rows (lane >> 4) + 4 wv
        x *= g.alpha;
        if (g.C0) x += g.beta * c0;
        if (orow == ocol) x += g.diag_add;
        g.C[(size_t)orow * g.ldc + ocol] = x;
    }
}

// Joseph covariance update (MSCKF.py:612-614) in ONE launch.  With Y = P T_H^T and S = T_H P T_H^T + sigma^2 I the form
//     P+ = (I - K T_H) P (I - K T_H)^T + sigma^2 K K^T
// expands to  Pn = P - K Y^T - Y K^T + K S K^T  =  P - K Y^T + W K^T,  W = K S - Y  (the residual of K = Y S^-1: the Joseph
// correction), and P_out = (Pn + Pn^T) / 2.  One workgroup per PAIR of 16 x 16 tiles (I, J), I <= J: both tiles of Pn
// over the virtual product depth 2 dc ([-K | W] [Y | K]^T, cut into 32 contiguous pieces as in k_gemm_f64), the halves
// of the symmetrisation meet in LDS and both mirror tiles of P_out leave bit-equal.  Round 2 ran B2, D, Pn and the
// symmetrisation as four launches of ~5 us each.
struct JosephArgs {
    const double* P; int ldp;     // prior covariance (d x d)
    const double* K;              // gain (d x dc)
    const double* Y;              // P T_H^T (d x dc)
    const double* W;              // K S - Y (d x dc)
    double* Pout; int ldo;
    int d, dc;
};
__global__ __launch_bounds__(64 * GEMM_WAVES) void k_joseph_f64(JosephArgs g) {
    __shared__ double sAcc[2][GEMM_WAVES][4][64];
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (ti > tj) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m0 = ti * 16, n0 = tj * 16;
    const int r = lane & 15, grp = wv * 4 + (lane >> 4);
    const int K2 = 2 * g.dc;
    const int kq = (K2 + 4 * GEMM_WAVES - 1) / (4 * GEMM_WAVES);
    const int k_lo = wv * 4 * kq + (lane >> 4);      // (k = k_lo + 4 u: 16 rows x 32 contiguous bytes per load instruction)
    (void)grp;
    constexpr int TRIP = 16;                        // 2 dc <= 512: one trip
    const bool ra = m0 + r < g.d, rb = n0 + r < g.d;
    const size_t oa = (size_t)(m0 + r) * g.dc, ob = (size_t)(n0 + r) * g.dc;
    v4d acc1 = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
    for (int s0 = 0; s0 < kq; s0 += TRIP) {
        double a1[TRIP], b1[TRIP], a2[TRIP], b2[TRIP];
#pragma unroll
        for (int u = 0; u < TRIP; ++u) {
            const int k = k_lo + 4 * (s0 + u);
            const bool kok = (s0 + u < kq) && (k < K2);
            const bool first = k < g.dc;               // [-K | W] [Y | K]^T
            const int kk = first ? k : k - g.dc;
            // tile (I, J): rows m0.., columns n0..; tile (J, I): rows n0.., columns m0..  (unconditional loads from
            // selected / clamped addresses: guarded loads compile to an exec-mask branch each)
            const double* Ap = first ? g.K : g.W;
            const double* Bp = first ? g.Y : g.K;
            const double sgn = first ? -1.0 : 1.0;
            const bool va = kok && ra, vb = kok && rb;
            const size_t ia = va ? oa + kk : 0, ib = vb ? ob + kk : 0;
            const double xa1 = Ap[ia], xb1 = Bp[ib], xa2 = Ap[ib], xb2 = Bp[ia];
            a1[u] = va ? sgn * xa1 : 0.0;
            b1[u] = vb ? xb1 : 0.0;
            a2[u] = vb ? sgn * xa2 : 0.0;
            b2[u] = va ? xb2 : 0.0;
        }
#pragma unroll
        for (int u = 0; u < TRIP; ++u) {
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[u], b2[u], acc2, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { sAcc[0][wv][i][lane] = acc1[i]; sAcc[1][wv][i][lane] = acc2[i]; }
    __syncthreads();
    // element (i, j) of tile (I, J): i = (lane >> 4) + 4 reg, j = lane & 15 -- the first four wavefronts finish it (reg = wv)
    if (wv < 4) {
        const int i = (lane >> 4) + 4 * wv, j = lane & 15;
        const int gi = m0 + i, gj = n0 + j;
        if (gi < g.d && gj < g.d) {
            double x1 = 0.0, x2 = 0.0;
            const int l2 = (j & 3) * 16 + i, r2 = j >> 2;            // element (j, i) of tile (J, I)
#pragma unroll
            for (int w = 0; w < GEMM_WAVES; ++w) { x1 += sAcc[0][w][wv][lane]; x2 += sAcc[1][w][r2][l2]; }
            const double pn1 = g.P[(size_t)gi * g.ldp + gj] + x1;     // Pn[gi][gj]
            const double pn2 = g.P[(size_t)gj * g.ldp + gi] + x2;     // Pn[gj][gi]
            const double o = 0.5 * (pn1 + pn2);
            g.Pout[(size_t)gi * g.ldo + gj] = o;
            g.Pout[(size_t)gj * g.ldo + gi] = o;
        }
    }
}

// The same product on the f32 matrix cores (v_mfma_f32_16x16x4_f32) for MSCKF_DTYPE_F32: the Joseph covariance
// update (MSCKF.py:612-614) with fp32 operands and fp32 accumulation.  Operands may be stored as double (the
// fp64 state: P, K, Y, T -- rounded to float as they are loaded) or float (the update's own intermediates
// B2, D, Pn); the result is stored as float or double.
typedef float v4f __attribute__((ext_vector_type(4)));

struct Gemm32Args {
    const void* A; int lda; int a_f32;    // A[M][K]
    const void* B; int ldb; int b_f32;    // transB ? B[N][K] : B[K][N]
    const void* C0; int ldc0; int c0_f32; // optional addend
    void* C; int ldc; int c_f32;
    int M, N, K;
    float alpha, beta;
    int transB;
    int tri;                              // 1: B[N][K] upper triangular (k >= n), transB only
};

__device__ __forceinline__ float ld32(const void* p, size_t i, int f32) {
    return f32 ? static_cast<const float*>(p)[i] : (float)static_cast<const double*>(p)[i];
}

template <bool AF, bool C0F>      // A / C0 stored as float (else double); B is always one of the fp64 operands
__global__ __launch_bounds__(64 * GEMM_WAVES) void k_gemm_f32(Gemm32Args g) {
    // as k_gemm_f64: GEMM_WAVES wavefronts per tile, the K range in GEMM_WAVES * 4 contiguous pieces, partial tiles summed in LDS
    __shared__ float sAcc[GEMM_WAVES][4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const int r = lane & 15, grp = wv * 4 + (lane >> 4);
    // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg; wavefront w < 4 finishes register w
    const int ocol = n0 + (lane & 15), orow = m0 + 4 * (lane >> 4) + (wv & 3);
    const bool ook = wv < 4 && orow < g.M && ocol < g.N;
    float c0 = 0.f;
    if (g.C0 && ook) c0 = ld32(g.C0, (size_t)orow * g.ldc0 + ocol, C0F);
    const int kbeg = (g.tri == 1) ? n0 : 0;
    const int klen = g.K - kbeg;
    const int kq = (klen + 4 * GEMM_WAVES - 1) / (4 * GEMM_WAVES);
    const int k_lo = kbeg + wv * 4 * kq + (lane >> 4);     // k = k_lo + 4 u (as k_gemm_f64: 16 rows x 4 consecutive k per load)
    const int k_hi = g.K;
    (void)grp;
    const int arow = m0 + r, bcol = n0 + r;
    const bool aok = arow < g.M, bok = bcol < g.N;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int TRIP = 16;
    for (int s0 = 0; s0 < kq; s0 += TRIP) {
        float av[TRIP], bv[TRIP];
#pragma unroll
        for (int u = 0; u < TRIP; ++u) {
            const int k = k_lo + 4 * (s0 + u);
            const bool kok = (s0 + u < kq) && (k < k_hi);
            av[u] = (aok && kok) ? ld32(g.A, (size_t)arow * g.lda + k, AF) : 0.f;
            bv[u] = (bok && kok) ? ld32(g.B, (size_t)bcol * g.ldb + k, 0) : 0.f;      // B[N][K] (transB form), double
        }
#pragma unroll
        for (int u = 0; u < TRIP; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) sAcc[wv][i][lane] = acc[i];
    __syncthreads();
    if (ook) {
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < GEMM_WAVES; ++w) x += sAcc[w][wv & 3][lane];
        x *= g.alpha;
        if (g.C0) x += g.beta * c0;
        static_cast<float*>(g.C)[(size_t)orow * g.ldc + ocol] = x;
    }
}

// P_out = (Pn + Pn^T) / 2 from the fp32 Pn, stored as double  (reference MSCKF.py:614)
__global__ void k_symmetrize_f32(const float* Pn, double* Pout, int d, int ld) {
    const int i = blockIdx.y * 16 + threadIdx.y, j = blockIdx.x * 16 + threadIdx.x;
    if (i < d && j < d) Pout[(size_t)i * ld + j] = (double)(0.5f * (Pn[(size_t)i * ld + j] + Pn[(size_t)j * ld + i]));
}

// Cholesky S = L L^T of an n x n SPD matrix by one workgroup.  Works on the
// packed lower triangle in LDS when it fits (use_lds), else in place in HBM/L2.
// Writes L (row-major, ld = n), U = L^T (row-major) and invd[j] = 1 / L[j][j].
struct CholArgs {
    const double* S; int lds_;    // input, row-major, leading dimension lds_
    double* L; double* U; double* invd; int n;
    int use_lds;
    double* work;                 // n*(n+1)/2 doubles when !use_lds
    int* status;                  // [0] set to 1 when a pivot is not positive
    long long* stamps;            // debug builds (CHOL16_STAMPS) only: s_memtime stamps per step and wavefront
    double diag_rel;              // k_chol16: factor S + diag_rel * trace(S) / n * I (0: S itself; nothing sets it since round 5)
    unsigned long long* done_flag;   // k_chol16, optional: done_val | rows of U (16 per step) that are final and visible device-wide:
    unsigned long long done_val;     //   a kernel of another stream (k_root_gain) follows the factor row block by row block
};

template <int T>
__global__ __launch_bounds__(T) void k_chol(CholArgs c) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int t = threadIdx.x, n = c.n;
    double* A = c.use_lds ? smem : c.work;      // packed lower by rows: (i,j) at i(i+1)/2 + j
    if (t == 0) c.status[0] = 0;                // thread 0 also writes the failure flag: program order
    for (int i = t / 32; i < n; i += T / 32)
        for (int j = t % 32; j <= i; j += 32) A[i * (i + 1) / 2 + j] = c.S[(size_t)i * c.lds_ + j];
    __syncthreads();
    const int tx = t % 32, ty = t / 32;          // 32 x (T/32) thread tile over the trailing block
    for (int k = 0; k < n; ++k) {
        const double piv = A[k * (k + 1) / 2 + k];
        if (!(piv > 0.0)) { if (t == 0) c.status[0] = 1; break; }
        const double dinv = 1.0 / sqrt(piv);
        __syncthreads();
        for (int i = k + t; i < n; i += T) A[i * (i + 1) / 2 + k] *= dinv;   // column k, incl. diagonal -> sqrt(piv)
        __syncthreads();
        for (int i = k + 1 + ty; i < n; i += T / 32) {
            const double lik = A[i * (i + 1) / 2 + k];
            double* row = A + i * (i + 1) / 2;
            for (int j = k + 1 + tx; j <= i; j += 32) row[j] -= lik * A[j * (j + 1) / 2 + k];
        }
        __syncthreads();
    }
    __syncthreads();
    for (int i = t / 32; i < n; i += T / 32)
        for (int j = t % 32; j < n; j += 32) {
            const double x = (j <= i) ? A[i * (i + 1) / 2 + j] : 0.0;
            c.L[(size_t)i * n + j] = x;
            c.U[(size_t)j * n + i] = x;
        }
    for (int j = t; j < n; j += T) c.invd[j] = 1.0 / A[j * (j + 1) / 2 + j];
}

// K row i = Y row i * S^-1 : forward sweep with U = L^T rows (L x = y), backward
// sweep with L rows (L^T k = x).  One wavefront per row; the vector lives in
// registers (lane l holds elements l, l+64, ...), the pivot element travels by
// v_readlane.  Also dx[i] = K[i,:] . z.
struct SolveArgs {
    const double* Y; int ldy;     // [d][n]
    const double* L; const double* U; const double* invd; int n;
    const double* Lp;             // optional: L packed by rows, (i, j) at i(i+1)/2 + j (k_chol_tile writes it)
    const double* z; int zstride; // r_n (column of the compressed block)
    double* Kg; int ldk;          // [d][n]
    double* dx; int d;
};

template <int NREG>
__global__ __launch_bounds__(64) void k_solve(SolveArgs s) {
    const int row = blockIdx.x, lane = threadIdx.x, n = s.n;
    double x[NREG];
#pragma unroll
    for (int m = 0; m < NREG; ++m) {
        const int i = lane + 64 * m;
        x[m] = (i < n) ? s.Y[(size_t)row * s.ldy + i] : 0.0;
    }
    // forward: for j: xj = x[j] / L[j][j]; x[i] -= L[i][j] xj (i > j); L[i][j] = U[j][i]
#pragma unroll
    for (int mj = 0; mj < NREG; ++mj) {
        for (int jj = 0; jj < 64; ++jj) {
            const int j = mj * 64 + jj;
            if (j >= n) break;
            const double xj = readlane_d(x[mj], jj) * s.invd[j];
            if (lane == jj) x[mj] = xj;
            const double* urow = s.U + (size_t)j * n;
#pragma unroll
            for (int m = 0; m < NREG; ++m) {
                if (m < mj) continue;
                const int i = lane + 64 * m;
                if (i > j && i < n) x[m] -= urow[i] * xj;
            }
        }
    }
    // backward: for j = n-1..0: kj = x[j] / L[j][j]; x[i] -= L[j][i] kj (i < j)
#pragma unroll
    for (int mj = NREG - 1; mj >= 0; --mj) {
        for (int jj = 63; jj >= 0; --jj) {
            const int j = mj * 64 + jj;
            if (j >= n) continue;
            const double kj = readlane_d(x[mj], jj) * s.invd[j];
            if (lane == jj) x[mj] = kj;
            const double* lrow = s.L + (size_t)j * n;
#pragma unroll
            for (int m = 0; m < NREG; ++m) {
                if (m > mj) continue;
                const int i = lane + 64 * m;
                if (i < j) x[m] -= lrow[i] * kj;
            }
        }
    }
    double dot = 0.0;
#pragma unroll
    for (int m = 0; m < NREG; ++m) {
        const int i = lane + 64 * m;
        if (i < n) {
            s.Kg[(size_t)row * s.ldk + i] = x[m];
            dot += x[m] * s.z[(size_t)i * s.zstride];
        }
    }
    dot = wave_sum(dot);
    if (lane == 0) s.dx[row] = dot;
}

// Register-resident tiled Cholesky for n <= 4 * CHOL_TILE_MAX_NT.  The whole lower triangle lives
// in registers as 4x4 tiles (tile e = t + T*s of a column-wise enumeration of the lower tiles, last column first, up
// to three per thread), so the trailing update reads only the current 4-column panel from LDS
// (32 doubles per tile and step) and never re-reads or re-writes the matrix itself -- the LDS
// traffic of the packed-triangle version (k_chol_blk: 80 LDS operations per tile and step, 79 of
// its 125 us) was the bound.  Step k:
//   P(k)  owners of the tiles (i, k), i > k, solve X L11^T = A against the published 4x4 diagonal
//         factor and publish their rows of the panel;                         barrier
//   T(k)  every tile (i, j), j > k, subtracts panel_i panel_j^T; the owner of tile (k+1, k+1) takes
//         it first, factors it (right-looking, so the divisions chain only through the pivots)
//         and publishes L11 for the next step while the others are still updating;   barrier
// No global memory is touched inside the loop (a barrier would wait for outstanding stores).
constexpr int CHOL_TILE_MAX_NT = 47;      // 47 * 48 / 2 = 1128 tiles <= 3 * 512

template <int T, int NS = (CHOL_TILE_MAX_NT * (CHOL_TILE_MAX_NT + 1) / 2 + T - 1) / T>
__global__ __launch_bounds__(T) void k_chol_tile(CholArgs c) {
    // panel, laid out [column q][row-in-tile u][tile row]: lanes with consecutive tile indices read
    // consecutive doubles (a [row][4] layout put 64 lanes on 4 banks: 16-way conflicts, 120 us)
    constexpr int NTP = CHOL_TILE_MAX_NT + 1;
    __shared__ __attribute__((aligned(16))) double sP[16 * NTP];
    __shared__ __attribute__((aligned(16))) double sD[20];                         // L11 (4x4) + 1/diag
    __shared__ double sInv[4 * CHOL_TILE_MAX_NT];                                  // 1/L_jj of every factored column
    __shared__ int sBad;
    const int t = threadIdx.x, n = c.n;
    const int nt = (n + 3) >> 2, ntile = nt * (nt + 1) / 2;
    int ti[NS], tj[NS];
    double a[NS][4][4];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = t + T * s;
        ti[s] = -1; tj[s] = -1;
        if (e < ntile) {
            // Tiles are enumerated column by column from the LAST column backwards (column nt-1-m starts at
            // m (m + 1) / 2): a tile (i, j) is live while k < j, so at every step the live tiles are exactly the
            // first A_k = (nt-k-1)(nt-k)/2 of the enumeration -- packed into the lowest register slots of all
            // threads (61 slot-steps per wavefront instead of 121 with a row-major order) -- and the panel of
            // step k is one run of consecutive threads (one or two wavefronts execute the panel solve, not all).
            int m = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
            while (m * (m + 1) / 2 > e) --m;
            while ((m + 1) * (m + 2) / 2 <= e) ++m;
            tj[s] = nt - 1 - m; ti[s] = tj[s] + (e - m * (m + 1) / 2);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = 4 * ti[s] + u, j = 4 * tj[s] + v;
                double x = (i == j) ? 1.0 : 0.0;                      // identity padding beyond n
                if (ti[s] >= 0 && i < n && j < n) x = c.S[(size_t)i * c.lds_ + j];
                a[s][u][v] = x;
            }
    }
    if (t == 0) { sBad = 0; c.status[0] = 0; }
    __syncthreads();
    // in-register factor of a diagonal tile + publication (right-looking: every update of the
    // remaining entries is independent, only the four pivots form a chain)
    auto factor_diag = [&](double (&d)[4][4], int kd) {
        int bad = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const double piv = d[p][p];
            if (!(piv > 0.0)) bad = 1;
            const double di = (piv > 1e-200 && piv < 1e200) ? fast_rsqrt(piv) : 1.0 / sqrt(piv);
            d[p][p] = fast_norm(piv, di);
            sD[16 + p] = di;
            sInv[4 * kd + p] = di;
#pragma unroll
            for (int b = p + 1; b < 4; ++b) d[b][p] *= di;
#pragma unroll
            for (int b = p + 1; b < 4; ++b)
#pragma unroll
                for (int q = p + 1; q <= b; ++q) d[b][q] = fma(-d[b][p], d[q][p], d[b][q]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v <= u; ++v) sD[4 * u + v] = d[u][v];
        if (bad) sBad = 1;
    };
    auto update_diag_tile = [&](double (&d)[4][4], int ri) {
        double Li[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) Li[u][q] = sP[(4 * q + u) * NTP + ri];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v <= u; ++v) {
                double x = d[u][v];
#pragma unroll
                for (int q = 0; q < 4; ++q) x = fma(-Li[u][q], Li[v][q], x);
                d[u][v] = x;
            }
    };
    auto update_tile = [&](double (&d)[4][4], int ri, int rj) {
        double Li[4][4], Lj[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                Li[u][q] = sP[(4 * q + u) * NTP + ri];
                Lj[u][q] = sP[(4 * q + u) * NTP + rj];
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                double x = d[u][v];
#pragma unroll
                for (int q = 0; q < 4; ++q) x = fma(-Li[u][q], Lj[v][q], x);
                d[u][v] = x;
            }
    };
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (ti[s] == 0 && tj[s] == 0) factor_diag(a[s], 0);
    __syncthreads();
#ifdef CHOL_PROF
    long long pc[6] = {0, 0, 0, 0, 0, 0};
    long long tp = __builtin_readcyclecounter();
#define CHOL_TICK(q) do { const long long tn_ = __builtin_readcyclecounter(); pc[q] += tn_ - tp; tp = tn_; } while (0)
#else
#define CHOL_TICK(q) do { } while (0)
#endif
    for (int k = 0; k < nt; ++k) {
        if (sBad) break;                           // uniform: read after a barrier
        // ---- P(k): the panel below the diagonal tile -------------------------------------
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (tj[s] == k && ti[s] > k) {
                double l10 = sD[4], l20 = sD[8], l21 = sD[9], l30 = sD[12], l31 = sD[13], l32 = sD[14];
                double i0 = sD[16], i1 = sD[17], i2 = sD[18], i3 = sD[19];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double x0 = a[s][u][0] * i0;
                    const double x1 = fma(-x0, l10, a[s][u][1]) * i1;
                    const double x2 = fma(-x1, l21, fma(-x0, l20, a[s][u][2])) * i2;
                    const double x3 = fma(-x2, l32, fma(-x1, l31, fma(-x0, l30, a[s][u][3]))) * i3;
                    a[s][u][0] = x0; a[s][u][1] = x1; a[s][u][2] = x2; a[s][u][3] = x3;
                    double* row = sP + u * NTP + ti[s];
                    row[0] = x0; row[4 * NTP] = x1; row[8 * NTP] = x2; row[12 * NTP] = x3;
                }
            }
        }
        CHOL_TICK(0);
        __syncthreads();
        CHOL_TICK(1);
        // ---- T(k): trailing update, next diagonal tile first ----------------------------------
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (ti[s] == k + 1 && tj[s] == k + 1) {
                // the chain every other wavefront ends up waiting for: issue it ahead of the co-resident
                // wavefronts' trailing updates (without the priority its ~90 dependent FP64 instructions
                // took 4000 cycles per step, most of them waiting for an issue slot)
                __builtin_amdgcn_s_setprio(3);
                update_diag_tile(a[s], k + 1);
                factor_diag(a[s], k + 1);
                __builtin_amdgcn_s_setprio(0);
            }
        }
        CHOL_TICK(2);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (tj[s] > k && !(ti[s] == k + 1 && tj[s] == k + 1)) update_tile(a[s], ti[s], tj[s]);
        }
        CHOL_TICK(3);
        __syncthreads();
        CHOL_TICK(4);
    }
#ifdef CHOL_PROF
    if ((t & 63) == 0) printf("chol wave %d: panel %lld  bar1 %lld  diag %lld  trailing %lld  bar2 %lld (cycles, %d steps)\n", t >> 6,
                              pc[0], pc[1], pc[2], pc[3], pc[4], nt);
#endif
    if (sBad) { if (t == 0) c.status[0] = 1; return; }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (ti[s] < 0) continue;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = 4 * ti[s] + u, j = 4 * tj[s] + v;
                if (i < n && j <= i) {
                    const double x = a[s][u][v];
                    c.L[(size_t)i * n + j] = x;
                    c.U[(size_t)j * n + i] = x;
                    // packed copy for k_solve_lds with columns scaled to a unit diagonal (L D^-1)
                    if (c.work) c.work[i * (i + 1) / 2 + j] = (i == j) ? 1.0 : x * sInv[j];
                    if (i == j) c.invd[j] = sInv[j];
                }
            }
    }
}

// K = Y S^-1 with the packed factor L resident in LDS (n(n+1)/2 doubles): one
// wavefront per row of Y, WAVES rows per workgroup.
// MODE: 3 = both sweeps (K = Y S^-1), 1 = forward only (X = Y L^-T), 2 = backward only (K = X L^-1); the
// one-sided forms serve the two-block factorisation of windows wider than one register-tiled Cholesky
// (launch_gain_blocked) and need the unit-diagonal packed factor.  Rows may be updated in place (Kg == Y).
template <int V> struct CTagS { static constexpr int value = V; };

template <int NREG, int WAVES, int ROWS = 1, bool UNIT = false, int MODE = 3>
__global__ __launch_bounds__(64 * WAVES) void k_solve_lds(SolveArgs s) {
    static_assert(MODE == 3 || UNIT, "one-sided sweeps use the unit-diagonal packed factor");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n = s.n, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    double* lp = smem;                          // packed lower: (i, j) at i(i+1)/2 + j
    double* sinv = smem + (size_t)n * (n + 1) / 2;
    constexpr bool unit = UNIT;                 // the packed copy s.Lp has unit diagonal (columns scaled by 1/L_jj)
    if (UNIT) {                                 // flat copy, ALL loads of a thread in flight at once: one trip to L2, not four
        const int np = n * (n + 1) / 2;
        constexpr int PER = (64 * 3 * (64 * 3 + 1) / 2 + 64 * WAVES - 1) / (64 * WAVES);   // n <= 192
        if (NREG <= 3 && np <= PER * 64 * WAVES) {
            double r[PER];
#pragma unroll
            for (int q = 0; q < PER; ++q) { const int idx = t + q * 64 * WAVES; r[q] = (idx < np) ? s.Lp[idx] : 0.0; }
#pragma unroll
            for (int q = 0; q < PER; ++q) { const int idx = t + q * 64 * WAVES; if (idx < np) lp[idx] = r[q]; }
        } else {
#pragma unroll 8
            for (int idx = t; idx < np; idx += 64 * WAVES) lp[idx] = s.Lp[idx];
        }
    } else {
        for (int i = wv; i < n; i += WAVES)
            for (int j = lane; j <= i; j += 64) lp[i * (i + 1) / 2 + j] = s.L[(size_t)i * n + j];
    }
    for (int j = t; j < n; j += 64 * WAVES) sinv[j] = s.invd[j];
    __syncthreads();
    // ROWS rows of Y per wavefront: the rows are independent, so their pivot -> readlane -> FMA chains
    // interleave, and each column of L is read from LDS once for all of them
    const int row0 = (blockIdx.x * WAVES + wv) * ROWS;
    if (row0 >= s.d) return;
    double x[ROWS][NREG];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int m = 0; m < NREG; ++m) {
            const int i = lane + 64 * m;
            x[r][m] = (row0 + r < s.d && i < n) ? s.Y[(size_t)(row0 + r) * s.ldy + i] : 0.0;
        }
    // Both sweeps are FULLY unrolled: with the step index a compile-time constant the LDS reads carry their column /
    // row in the instruction's immediate offset (no address arithmetic per step), the pivot lane of v_readlane is an
    // immediate, and only the register that holds the diagonal needs a lane mask.  A step is ~13 instructions.
    int pb[NREG], pcol[NREG];                   // packed row starts i (i + 1) / 2 of this lane's rows (rows >= n: row 0, never used); columns
#pragma unroll
    for (int m = 0; m < NREG; ++m) { const int i = lane + 64 * m; pb[m] = (i < n) ? i * (i + 1) / 2 : 0; pcol[m] = (i < n) ? i : 0; }
    // forward sweep  L x = y  (column j + 2 is requested before step j's readlane -> FMA chain runs)
    if constexpr ((MODE & 1) != 0) {
        auto colf = [&](auto tagm, int j, double (&lv)[NREG]) {             // rows of column j, unmasked
            constexpr int mj = decltype(tagm)::value;
#pragma unroll
            for (int m = 0; m < NREG; ++m) lv[m] = (m >= mj) ? lp[pb[m] + j] : 0.0;
        };
        auto fwd_seg = [&](auto tagm) {
            constexpr int mj = decltype(tagm)::value;
            double lb[3][NREG];                                              // columns j, j + 1, j + 2 in slots j % 3, ... (no register moves)
            colf(tagm, mj * 64, lb[0]); colf(tagm, mj * 64 + 1, lb[1]);
#pragma unroll
            for (int jj = 0; jj < 64; ++jj) {
                const int j = mj * 64 + jj;
                if (j < n) {
                    colf(tagm, j + 2, lb[(jj + 2) % 3]);                     // (past column n - 1: words of the factor's tail, never used)
                    double lv[NREG];
#pragma unroll
                    for (int m = 0; m < NREG; ++m) lv[m] = (m == mj) ? ((lane > jj) ? lb[jj % 3][m] : 0.0) : lb[jj % 3][m];
                    const double sj = unit ? 1.0 : sinv[j];
#pragma unroll
                    for (int r = 0; r < ROWS; ++r) {
                        const double xj = unit ? readlane_d(x[r][mj], jj) : readlane_d(x[r][mj], jj) * sj;
                        if (!unit && lane == jj) x[r][mj] = xj;
#pragma unroll
                        for (int m = 0; m < NREG; ++m)
                            if (m >= mj) x[r][m] -= lv[m] * xj;
                    }
                }
            }
        };
        fwd_seg(CTagS<0>{});
        if constexpr (NREG > 1) fwd_seg(CTagS<1>{});
        if constexpr (NREG > 2) fwd_seg(CTagS<2>{});
        if constexpr (NREG > 3) fwd_seg(CTagS<3>{});
    }
    // unit-diagonal factor L' = L D^-1:  L x = y  <=>  L' (D x) = y  and  L^T k = x  <=>  L'^T k = D^-1 x,
    // so the two sweeps need no per-step scaling, just D^-2 in between (D^-1 after a forward-only sweep
    // and before a backward-only one)
    if (unit) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int m = 0; m < NREG; ++m) {
                const int i = lane + 64 * m;
                const double di = (i < n) ? sinv[i] : 0.0;
                x[r][m] *= (MODE == 3) ? di * di : di;
            }
    }
    // backward sweep  L^T k = x  (row j of L: contiguous in the packed factor; row j - 2 requested ahead)
    if constexpr ((MODE & 2) != 0) {
        int nb = n;
        asm volatile("" : "+s"(nb));                                        // (keeps the forward sweep's 192 predicates from being kept alive)
        auto rowb = [&](auto tagm, int j, double (&lv)[NREG]) {
            constexpr int mj = decltype(tagm)::value;
            const double* rowj = lp + (j >= 0 ? j * (j + 1) / 2 : 0);
#pragma unroll
            for (int m = 0; m < NREG; ++m) lv[m] = (m <= mj) ? rowj[pcol[m]] : 0.0;   // (lanes at / right of the diagonal: masked below)
        };
        auto bwd_seg = [&](auto tagm) {
            constexpr int mj = decltype(tagm)::value;
            if (mj * 64 >= nb) return;
            const int jtop = min(63, nb - 1 - mj * 64);
            // rows j, j - 1, j - 2 in slots j % 3, ...: the first two are requested here, whatever row the sweep starts at
            double lb[3][NREG];
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if ((jtop % 3) == q) { rowb(tagm, mj * 64 + jtop, lb[q]); rowb(tagm, mj * 64 + jtop - 1, lb[(q + 2) % 3]); }
#pragma unroll
            for (int jj = 63; jj >= 0; --jj) {
                const int j = mj * 64 + jj;
                if (j < nb) {
                    rowb(tagm, j - 2, lb[(jj + 1) % 3]);                     // (jj - 2) mod 3
                    double lv[NREG];
#pragma unroll
                    for (int m = 0; m < NREG; ++m) lv[m] = (m == mj) ? ((lane < jj) ? lb[jj % 3][m] : 0.0) : lb[jj % 3][m];
                    const double sj = unit ? 1.0 : sinv[j];
#pragma unroll
                    for (int r = 0; r < ROWS; ++r) {
                        const double kj = unit ? readlane_d(x[r][mj], jj) : readlane_d(x[r][mj], jj) * sj;
                        if (!unit && lane == jj) x[r][mj] = kj;
#pragma unroll
                        for (int m = 0; m < NREG; ++m)
                            if (m <= mj) x[r][m] -= lv[m] * kj;
                    }
                }
            }
        };
        if constexpr (NREG > 3) bwd_seg(CTagS<3>{});
        if constexpr (NREG > 2) bwd_seg(CTagS<2>{});
        if constexpr (NREG > 1) bwd_seg(CTagS<1>{});
        bwd_seg(CTagS<0>{});
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        if (row0 + r >= s.d) break;
        double dot = 0.0;
#pragma unroll
        for (int m = 0; m < NREG; ++m) {
            const int i = lane + 64 * m;
            if (i < n) {
                s.Kg[(size_t)(row0 + r) * s.ldk + i] = x[r][m];
                if (s.dx) dot += x[r][m] * s.z[(size_t)i * s.zstride];
            }
        }
        if (s.dx) {
            dot = wave_sum(dot);
            if (lane == 0) s.dx[row0 + r] = dot;
        }
    }
}

// dx = K z  (MSCKF.py:607) for the blocked solve: one wavefront per row of K.
__global__ __launch_bounds__(256) void k_matvec(const double* K, int ldk, int n, const double* z, int zstride, double* dx, int d) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= d) return;
    double dot = 0.0;
    for (int i = lane; i < n; i += 64) dot = fma(K[(size_t)row * ldk + i], z[(size_t)i * zstride], dot);
    dot = wave_sum(dot);
    if (lane == 0) dx[row] = dot;
}

// P_out = (Pn + Pn^T) / 2   (reference MSCKF.py:614)
__global__ void k_symmetrize(const double* Pn, double* Pout, int d, int ld) {
    const int i = blockIdx.y * 16 + threadIdx.y, j = blockIdx.x * 16 + threadIdx.x;
    if (i < d && j < d) Pout[(size_t)i * ld + j] = 0.5 * (Pn[(size_t)i * ld + j] + Pn[(size_t)j * ld + i]);
}

}  // namespace msckf

namespace msckf {

// ---------------------------------------------------------------------------------------------------
// Cholesky S = L L^T for n <= 16 * CHOL16_MAX_NB on 16 x 16 blocks held as FP64 matrix-core accumulators
// (v_mfma_f64_16x16x4_f64: lane (g, c) = (lane >> 4, lane & 15) holds rows {g + 4 i}, i < 4, of column c).
// One workgroup of CHOL16_W = 12 wavefronts.  A wavefront owns at most ONE diagonal block (a dedicated register
// block: no selection on the critical path) and up to CHOL16_NS off-diagonal blocks, which are enumerated block
// column by block column from the LAST column backwards (live blocks pack into the lowest register slots) and
// dealt round-robin; a block column has at most 11 panel blocks, so every wavefront has at most one of them and
// the table CHOL16_DIAG_OWNER picks, per column, a wavefront that has none for the diagonal block.
// Step k:
//   D  the owner of block (k, k) eliminates it in-wave, one pivot at a time, in the square-root-free form
//      a_rc -= a_rp a_cp / a_pp: the pivot comes by v_readlane, its reciprocal (rcp seed + one Newton step) is the
//      only thing on the dependent chain -- column p travels along the 16-lane rows by DPP row broadcast and to
//      the other row groups through LDS while the reciprocal is being formed.  Column p and 1 / a_pp are
//      published in LDS and a progress word counts the pivots.  The 16 square roots are ONE v_rsq_f64 at the
//      end (lane c scales column c);
//   P  the owners of the panel blocks (i, k) FOLLOW pivot by pivot (no barrier: they poll the progress word) with
//      the same rank-1 step on their block, scale by 1 / l_cc at the end, write X = A L_kk^-T to LDS in the
//      matrix-core operand layout (double-buffered over k) and to the outputs;               ONE barrier
//   T  every block (i, j), j > k, subtracts X_i X_j^T with four MFMAs (two blocks interleaved); the diagonal
//      blocks first, so the owner of (k+1, k+1) reaches D of the next step early.
// The per-column dependent chain is readlane -> rcp -> 2 fma -> fma instead of the ~1100 cycles per column of the
// 4 x 4 register-tile kernel, and there is one barrier per 16 columns.
// Outputs as k_chol_tile: L, U = L^T, 1 / l_jj, and the packed copy with unit diagonal for k_solve_lds.
constexpr int CHOL16_MAX_NB = 12;            // n <= 192
constexpr int CHOL16_W = 12;                 // wavefronts
constexpr int CHOL16_NS = 6;                 // off-diagonal blocks per wavefront: 66 / 12
constexpr long long CHOL16_UNSET = 0x7ff8dead0000beefLL;   // a NaN no arithmetic produces
// owner of the diagonal block of the column with m = nb - 1 - j columns behind it: not among the owners
// (m (m - 1) / 2 + q) % 12, q < m, of that column's panel blocks, and all different
__device__ constexpr int CHOL16_DIAG_OWNER[CHOL16_MAX_NB] = {5, 2, 4, 1, 11, 3, 10, 8, 0, 9, 7, 6};
template <int V> struct CTag { static constexpr int value = V; };
#ifdef CHOL16_STAMPS
#define CHOL16_STAMP(i) do { if (c.stamps && lane == 0) c.stamps[(k * W + wv) * 4 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define CHOL16_STAMP(i) do { } while (0)
#endif

template <int P>
__device__ __forceinline__ double row_bcast16(double x) {      // lane P of every 16-lane row, to all lanes of the row
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + P, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + P, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ unsigned lds_addr(const volatile void* p) { return (unsigned)(size_t)p; }

__global__ __launch_bounds__(64 * CHOL16_W) void k_chol16(CholArgs c) {
    constexpr int NBM = CHOL16_MAX_NB, W = CHOL16_W, NS = CHOL16_NS;
    // What the followers of step k need, in half k & 1: per pivot p the row of multipliers m_c = -a_cp / a_pp (zero
    // for c <= p) in [p][0..15], with [p][16] as the progress word -- it holds CHOL16_UNSET until the row is there
    // (one 17-lane ds_write_b64: lane 16's own multiplier is a zero) -- and 1 / l_cc in sRi (zero until published).
    // The owner of the diagonal block resets the other half for the next step.
    __shared__ __attribute__((aligned(16))) double sW[2][16][17];
    __shared__ __attribute__((aligned(16))) double sRi[2][16];
    // X_i of step k in buffer k & 1: [block row][column][row], the matrix-core operand order
    __shared__ __attribute__((aligned(16))) double sX[2 * NBM * 256];
    constexpr int XB = NBM * 256;
    __shared__ int sBad;
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = lane >> 4, cc = lane & 15;
    const int n = c.n, nb = (n + 15) >> 4, noff = nb * (nb - 1) / 2;
    const unsigned sw_addr = lds_addr(&sW[0][0][0]);
    const int cc4 = cc * 4;
    double dshift = 0.0;
    if (c.diag_rel > 0.0) {
        // trace(S), summed in a fixed order (wavefront sums, then wavefront 0 over the W of them): bit-reproducible
        double x = 0.0;
        for (int i = t; i < n; i += 64 * W) x += c.S[(size_t)i * c.lds_ + i];
        x = wave_sum(x);
        if (lane == 0) sRi[0][wv] = x;
        __syncthreads();
        double tr = 0.0;
        for (int w = 0; w < W; ++w) tr += sRi[0][w];
        dshift = fmax(c.diag_rel * tr / (double)n, 1e-150);          // (all-zero S: nothing was accepted, the rows stay negligible)
        __syncthreads();
    }
    auto load_block = [&](int bi_, int bj_) {
        v4d a;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 16 * bi_ + g + 4 * i, col = 16 * bj_ + cc;
            double x = (r == col) ? 1.0 : 0.0;                               // identity padding beyond n
            if (bi_ >= 0 && r < n && col < n) x = c.S[(size_t)r * c.lds_ + col] + ((r == col) ? dshift : 0.0);
            a[i] = x;
        }
        return a;
    };
    // outputs of a finished block (scaled by ri = 1 / l_cc of its columns)
    auto store_block = [&](const v4d& a, int bi_, int bj_, double ri) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 16 * bi_ + g + 4 * i, col = 16 * bj_ + cc;
            if (r < n && col <= r) {
                const double x = a[i];
                c.L[(size_t)r * n + col] = x;
                if (c.done_flag)                                             // (read by another kernel while this one runs: write-through)
                    __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)(unsigned long long*)(c.U + (size_t)col * n + r),
                                       (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else c.U[(size_t)col * n + r] = x;
                if (c.work) c.work[r * (r + 1) / 2 + col] = (r == col) ? 1.0 : x * ri;
                if (r == col) c.invd[col] = ri;
            }
        }
    };
    int jd = -1;                                                             // my diagonal block
#pragma unroll
    for (int m = 0; m < NBM; ++m)
        if (m < nb && CHOL16_DIAG_OWNER[m] == wv) jd = nb - 1 - m;
    v4d dg = load_block(jd, jd);
    int bi[NS], bj[NS];
    v4d acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = wv + W * s;
        bi[s] = -1; bj[s] = -1;
        if (e < noff) {
            int m = (int)((sqrtf(8.0f * (float)e + 1.0f) + 1.0f) * 0.5f);   // m (m - 1) / 2 <= e < m (m + 1) / 2
            while (m * (m - 1) / 2 > e) --m;
            while (m * (m + 1) / 2 <= e) ++m;
            bj[s] = nb - 1 - m; bi[s] = bj[s] + 1 + (e - m * (m - 1) / 2);
        }
        acc[s] = load_block(bi[s], bj[s]);
    }
    if (t == 0) { sBad = 0; c.status[0] = 0; }
    if (t < 32) { (&sRi[0][0])[t] = 0.0; sW[t >> 4][t & 15][16] = __longlong_as_double(CHOL16_UNSET); }
    __syncthreads();

    const int oo = g * 16 + cc;                  // operand element (row cc, column 4 q + g) of a block at [q * 64 + oo]
    for (int k = 0; k < nb; ++k) {
        CHOL16_STAMP(0);
        // (done_flag: block row k - 1 of U went out behind the last barrier; every wavefront waits for its stores here, the
        //  barrier of this step orders the waits, and thread 0 then publishes the 16 k rows that are final and visible)
        if (c.done_flag && k > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // The trailing update of step k - 1 (X blocks in buffer (k - 1) & 1) is split: T1 = my diagonal block and my block
        // of column k, right here; T2 = the rest, ahead of following THIS step's pivots -- a pivot costs the followers
        // less than its owner, they catch up -- or, for the owner of the diagonal block, behind its elimination (T2 must
        // only be done by this step's barrier, behind which that buffer is written again).  Leaving the owner's T2 for
        // the NEXT step (three buffers) was tried: no faster.
        const double* xb = sX + ((k + 1) & 1) * XB;
        auto update = [&](v4d& a, int i_, int j_) {
            const double* xi = xb + i_ * 256 + oo;
            const double* xj = xb + j_ * 256 + oo;
            double av[4], bv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { av[q] = -xi[q * 64]; bv[q] = xj[q * 64]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) a = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], a, 0, 0, 0);
        };
        auto update2 = [&](v4d& a, int ia, int ja, v4d& b, int ib, int jb) {
            const double* xi = xb + ia * 256 + oo;
            const double* xj = xb + ja * 256 + oo;
            const double* yi = xb + ib * 256 + oo;
            const double* yj = xb + jb * 256 + oo;
#pragma unroll
            for (int h = 0; h < 2; ++h) {                                    // two halves: 8 operand registers less
                double av[2], bv[2], cv[2], dv[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    av[q] = -xi[(2 * h + q) * 64]; bv[q] = xj[(2 * h + q) * 64];
                    cv[q] = -yi[(2 * h + q) * 64]; dv[q] = yj[(2 * h + q) * 64];
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    a = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], a, 0, 0, 0);
                    b = __builtin_amdgcn_mfma_f64_16x16x4f64(cv[q], dv[q], b, 0, 0, 0);
                }
            }
        };
        const int thr = k;
        auto t2 = [&](auto tagp) {                                           // slots 2 pr, 2 pr + 1: blocks right of column thr
            constexpr int PR = decltype(tagp)::value;
            const bool l0 = bj[2 * PR] > thr, l1 = bj[2 * PR + 1] > thr;
            if (l0 && l1) update2(acc[2 * PR + 1], bi[2 * PR + 1], bj[2 * PR + 1], acc[2 * PR], bi[2 * PR], bj[2 * PR]);
            else if (l0) update(acc[2 * PR], bi[2 * PR], bj[2 * PR]);
            else if (l1) update(acc[2 * PR + 1], bi[2 * PR + 1], bj[2 * PR + 1]);
        };
        int sp = -1;
#pragma unroll
        for (int s = 0; s < NS; ++s) if (bj[s] == k) sp = s;
        if (k > 0) {
            if (jd >= k) update(dg, jd, jd);
#pragma unroll
            for (int s = 0; s < NS; ++s) if (s == sp) update(acc[s], bi[s], k);
        }
        v4d asel = acc[0];
        int i0 = -1;
        double ri_out = 1.0;
        if (jd != k && k > 0) { t2(CTag<2>{}); t2(CTag<1>{}); t2(CTag<0>{}); }
        if (jd == k) {
            // ---- D: the diagonal block -----------------------------------------------------------------------------
            double d[4] = {dg[0], dg[1], dg[2], dg[3]};
            if (lane < 16) { sW[(k + 1) & 1][lane][16] = __longlong_as_double(CHOL16_UNSET); sRi[(k + 1) & 1][lane] = 0.0; }
            const unsigned w_out = sw_addr + (k & 1) * (16 * 17 * 8) + lane * 8;
            // Software pipeline over the pivots.  Entering pivot P, row P of the (symmetric) trailing square -- a[cc][P] for
            // the lanes of column cc, which sits in register IP of row group GP -- is already on its way by ds_bpermute
            // and the pivot by v_readlane: both were issued in pivot P - 1 right behind the update of THAT register, ahead
            // of the other three.  The exact chain is fmac -> readlane -> rcp -> fma -> fma -> fmac.  The order of the LDS
            // instructions is fixed, [multipliers of P][row P + 1], so that one s_waitcnt lgkmcnt(0) in the next pivot
            // covers exactly the ds_bpermute pair.
            int rlo, rhi;                                                    // row P in flight
            double piv;
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"                 // (matrix-core write -> read, unseen by the hazard recogniser)
                         "ds_bpermute_b32 %0, %2, %3\n\tds_bpermute_b32 %1, %2, %4"
                         : "=&v"(rlo), "=&v"(rhi) : "v"(cc4), "v"(__double2loint(d[0])), "v"(__double2hiint(d[0])) : "memory");
            piv = readlane_d(d[0], 0);
            auto pivot = [&](auto tagp) {
                constexpr int P = decltype(tagp)::value;
                constexpr int GN = (P + 1) & 3, IN = ((P + 1) >> 2) & 3;     // where row / pivot P + 1 live
                int ccl = cc;
                asm volatile("" : "+v"(ccl));                                // (keeps 32 lane masks from being hoisted into SGPRs)
                double r = __builtin_amdgcn_rcp(piv);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rlo), "+v"(rhi), "+v"(r)::"memory");   // (behind the v_rcp_f64)
                const double lc = __hiloint2double(rhi, rlo);
                const double lcm = (ccl > P) ? -lc : 0.0;
                const double e = fma(-piv, r, 1.0);                          // one Newton step, folded into the multiplier
                const double w = lcm * r;
                const double w2 = fma(w, e, w);
                if constexpr (P < 15) {
                    // a[r][c] -= a[r][P] a[c][P] / a[P][P]: the register of row P + 1 first, and its row and pivot go on their way
                    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(d[IN]) : "v"(w2), "i"(P));
                    // the multipliers for the followers (lanes 0..15: columns; lane 16: a zero, the progress word), then row P + 1
                    asm volatile("s_mov_b64 exec, 0x1ffff\n\tds_write_b64 %2, %3 offset:%4\n\ts_mov_b64 exec, -1\n\t"
                                 "ds_bpermute_b32 %0, %5, %6 offset:%8\n\tds_bpermute_b32 %1, %5, %7 offset:%8"
                                 : "=&v"(rlo), "=&v"(rhi)
                                 : "v"(w_out), "v"(w2), "i"(P * 17 * 8), "v"(cc4), "v"(__double2loint(d[IN])), "v"(__double2hiint(d[IN])), "i"(64 * GN)
                                 : "memory");
                    piv = readlane_d(d[IN], 16 * GN + P + 1);
                    asm volatile("s_nop 1\n\t"
                                 "v_fmac_f64_dpp %0, %0, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %1, %1, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %2, %2, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
                                 : "+v"(d[(IN + 1) & 3]), "+v"(d[(IN + 2) & 3]), "+v"(d[(IN + 3) & 3]) : "v"(w2), "i"(P));
                }
            };
            __builtin_amdgcn_s_setprio(3);
            pivot(CTag<0>{}); pivot(CTag<1>{}); pivot(CTag<2>{}); pivot(CTag<3>{});
            pivot(CTag<4>{}); pivot(CTag<5>{}); pivot(CTag<6>{}); pivot(CTag<7>{});
            pivot(CTag<8>{}); pivot(CTag<9>{}); pivot(CTag<10>{}); pivot(CTag<11>{});
            pivot(CTag<12>{}); pivot(CTag<13>{}); pivot(CTag<14>{}); pivot(CTag<15>{});
            // the pivots are the diagonal as it stands now (entry (c, c) is final once column c is eliminated): lane (g, c)
            // picks register c >> 2 and takes it from row group c & 3
            double pivs;
            {
                const double sel = (cc < 8) ? ((cc < 4) ? d[0] : d[1]) : ((cc < 12) ? d[2] : d[3]);
                const int src4 = (16 * (cc & 3) + cc) * 4;
                const int plo = __builtin_amdgcn_ds_bpermute(src4, __double2loint(sel)), phi = __builtin_amdgcn_ds_bpermute(src4, __double2hiint(sel));
                pivs = __hiloint2double(phi, plo);
            }
            const bool bad = __ballot(!(pivs > 1e-200 && pivs < 1e200)) != 0ull;   // (a NaN pivot carries through to the later ones)
            const double ri = bad ? 1.0 : fast_rsqrt(pivs);                  // lane c: 1 / l_cc -- sixteen roots in one go
            if (bad && lane == 0) sBad = 1;
            asm volatile("" ::: "memory");
            if (g == 0) sRi[k & 1][cc] = ri;
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) dg[i] = d[i] * ri;
            CHOL16_STAMP(1);
            if (k > 0) { t2(CTag<2>{}); t2(CTag<1>{}); t2(CTag<0>{}); }
            store_block(dg, k, k, ri);
        } else if (sp >= 0) {
            // ---- P: the panel block of this wavefront (at most one) follows the pivots -------------------------------
            CHOL16_STAMP(1);
            i0 = bi[0];
#pragma unroll
            for (int s = 1; s < NS; ++s) if (s == sp) { asel = acc[s]; i0 = bi[s]; }
            double a0[4] = {asel[0], asel[1], asel[2], asel[3]};
            const unsigned w_in = sw_addr + (k & 1) * (16 * 17 * 8);
            const unsigned w_cc = w_in + cc * 8;
            // Four pivots at a time: one poll (the progress word of the chunk's last pivot, by ONE lane -- eleven wavefronts
            // polling with all 64 keep the LDS busy and hold up the owner of the diagonal block), four rows of
            // multipliers read back to back, sixteen updates from registers.
            auto follow4 = [&](auto tagp) {
                constexpr int P0 = decltype(tagp)::value;
                constexpr int PL = (P0 + 3 < 14) ? P0 + 3 : 14;              // the last pivot with multipliers
                double* ap = a0;                                             // (a variable named only in asm operands is not captured)
                const unsigned win_l = w_in, wcc_l = w_cc;
                int spins = 0;
                do {
                    double flag;
                    asm volatile("s_mov_b64 exec, 1\n\tds_read_b64 %0, %1 offset:%2\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)"
                                 : "=v"(flag) : "v"(win_l), "i"(PL * 17 * 8 + 128) : "memory");
                    const int flo = __builtin_amdgcn_readfirstlane(__double2loint(flag));
                    const int fhi = __builtin_amdgcn_readfirstlane(__double2hiint(flag));
                    if (flo != (int)(CHOL16_UNSET & 0xffffffffLL) || fhi != (int)(CHOL16_UNSET >> 32)) break;
                    __builtin_amdgcn_s_sleep(2);
                } while (++spins < (1 << 22));
                double m0, m1, m2, m3 = 0.0;
                asm volatile("ds_read_b64 %0, %3 offset:%4\n\tds_read_b64 %1, %3 offset:%5\n\tds_read_b64 %2, %3 offset:%6"
                             : "=&v"(m0), "=&v"(m1), "=&v"(m2) : "v"(wcc_l), "i"(P0 * 17 * 8), "i"((P0 + 1) * 17 * 8), "i"((P0 + 2) * 17 * 8) : "memory");
                if constexpr (P0 + 3 < 15) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(m3) : "v"(wcc_l), "i"((P0 + 3) * 17 * 8) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3)::"memory");
#define CHOL16_FOLLOW1(PP, M)                                                                                            \
    asm volatile("s_nop 1\n\t"                                                                                          \
                 "v_fmac_f64_dpp %0, %0, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"                             \
                 "v_fmac_f64_dpp %1, %1, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"                             \
                 "v_fmac_f64_dpp %2, %2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"                             \
                 "v_fmac_f64_dpp %3, %3, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"                                 \
                 : "+v"(ap[0]), "+v"(ap[1]), "+v"(ap[2]), "+v"(ap[3]) : "v"(M), "i"(PP))
                CHOL16_FOLLOW1(P0, m0);
                CHOL16_FOLLOW1(P0 + 1, m1);
                CHOL16_FOLLOW1(P0 + 2, m2);
                if constexpr (P0 + 3 < 15) CHOL16_FOLLOW1(P0 + 3, m3);
#undef CHOL16_FOLLOW1
            };
#ifdef CHOL16_STAMPS
#define CHOL16_CSTAMP(i) do { if (c.stamps && lane == 0 && k == 0) c.stamps[12 * W * 4 + wv * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define CHOL16_CSTAMP(i) do { } while (0)
#endif
            CHOL16_CSTAMP(0);
            follow4(CTag<0>{}); CHOL16_CSTAMP(1); follow4(CTag<4>{}); CHOL16_CSTAMP(2); follow4(CTag<8>{}); CHOL16_CSTAMP(3); follow4(CTag<12>{});
            CHOL16_CSTAMP(4);
            {
                const unsigned ri_addr = lds_addr(&sRi[k & 1][cc]);
                int spins = 0;
                do {
                    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(ri_out) : "v"(ri_addr) : "memory");
                    if (ri_out != 0.0) break;
                    __builtin_amdgcn_s_sleep(1);
                } while (++spins < (1 << 22));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) asel[i] = a0[i] * ri_out;
            double* dst = sX + (k & 1) * XB + i0 * 256 + cc * 16 + g;
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[4 * i] = asel[i];                // [column][row]
            CHOL16_CSTAMP(5);
        } else {
            CHOL16_STAMP(1);
        }
        CHOL16_STAMP(2);
        __syncthreads();
        CHOL16_STAMP(3);
        if (sBad) break;                                   // uniform: read after a barrier
        if (c.done_flag && k > 0 && t == 0)
            __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)c.done_flag, c.done_val | (unsigned)(16 * k),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (i0 >= 0) store_block(asel, i0, k, ri_out);     // the finished panel block, off the critical path
    }
    __syncthreads();
    if (sBad && t == 0) c.status[0] = 1;
    if (c.done_flag) {
        // write-through stores of U -> every wavefront's wait -> barrier -> the count (MI355X_MICROARCH.md, inter-workgroup
        // visibility); a failed factorisation publishes too: its reader must not wait for rows that never come
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0)
            __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)c.done_flag, c.done_val | (unsigned)(16 * nb),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace msckf
