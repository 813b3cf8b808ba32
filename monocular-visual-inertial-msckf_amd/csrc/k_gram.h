// K5 for tracks that span more than 15 clone slots: compression in the INFORMATION form.
//   reference MSCKF.py:594-598 : Q, R = qr(H_X) ; T_H = R ; r_n = Q^T r_o
// The band pipeline (k_lsweep / k_sweep / k_wsweep) covers tracks of up to 15 slots; a track of 16 - 31 views fills most of
// the 6N columns, its rows cannot be folded through 60- or 90-column tiles, and the merge tree that used to take such
// batches (k_fold) is eight times slower per row.  K6-K7 (k_gstream.h) consume the compressed system 16 rows at a time and
// need nothing of it but  T^T T = H^T H  and  T^T r_n = H^T r  -- ANY square root of the augmented Gram matrix
//     [H r]^T [H r] = [[G, b], [b^T, c]] = U^T U,   U = [[T, r_n], [0, rho]]
// serves, so for these tracks the compression is a rank-q update of G per feature on the matrix cores and ONE Cholesky
// factorisation (k_chol16, n = 6N + 1 <= 192) instead of a Householder sweep over every row.
// Numerics: G is formed in fp64 (error ~1e-16 |G|); the update P+ = (P^-1 + G / sigma^2)^-1 sees a perturbation dG through
// P+ dG P+ / sigma^2, i.e. ~1e-12 relative at the sizes of this problem however ill-conditioned H is (the stack has rank
// 6N - 4: the Cholesky runs on G + eps I, eps = 1e-14 trace(G) / n -- a measurement of the unobservable directions with
// standard deviation sigma / sqrt(eps) = 1e7 sigma; checked in NumPy against the reference's own outputs: <= 4e-12 on dx and
// P+ for every golden fixture, recipe B with cond(H) ~ 1e16 included).  The band pipeline stays Householder: its tiles make
// the rows of R available one by one while the sweep runs, which is what lets K6-K7 run beside it.
#pragma once
#include <hip/hip_runtime.h>
#include "wave_ops.h"
#include "k_gain.h"

namespace msckf {

constexpr int GRAM_WAVES = 16;
constexpr int GRAM_ROWS = 60;                 // projected rows of a track: q <= 2 * 31 - 3 = 59, in MFMA steps of 4
constexpr int GRAM_MAX_NT = 12;               // 16-column tiles of the augmented system: 6N + 1 <= 192

struct GramArgs {
    const int* view_ptr;          // sorted CSR
    const int* obs_slot;
    const long long* blk_off;     // [F] K4 stack blocks: row-major q x (6M + 1), [H_o | r_o]
    const double* stack;
    const int* rank;              // q = 2M - rank
    const unsigned char* accepted;
    int f0, nf;                   // the wide tracks: sorted features [f0, f0 + nf)
    int dc;                       // clone columns 6N; the rhs is column dc of the augmented system
    int nt;                       // tiles: (dc + 1 + 15) / 16
    double* part;                 // [gridDim.x][nt (nt + 1) / 2][256]: partial tiles (ta <= tb), accumulator layout
};

__host__ __device__ inline int gram_ld(int nt) { return 16 * (nt | 1); }     // odd multiple of 16: the four rows of an MFMA step on different banks
__host__ __device__ inline size_t gram_lds_bytes(int nt) { return ((size_t)GRAM_ROWS * gram_ld(nt) + 192) * 8; }   // image | column map | tile mask

// One workgroup of 16 wavefronts takes features f0 + blockIdx.x, + gridDim.x, ...: the feature's block is scattered into a
// dense q x (6N + 1) image in LDS (columns of clones it does not see are zero), every wavefront accumulates its share of
// the upper tile pairs G(ta, tb) += Hd[:, ta]^T Hd[:, tb] (pairs with a tile the track does not touch are skipped).
__global__ __launch_bounds__(64 * GRAM_WAVES) void k_gram(GramArgs p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = lane >> 4, cc = lane & 15;
    const int nt = p.nt, LD = gram_ld(nt), npairs = nt * (nt + 1) / 2;
    double* Hd = smem;                                          // [GRAM_ROWS][LD]
    int* sCol = reinterpret_cast<int*>(smem + (size_t)GRAM_ROWS * LD);          // [<= 32 * 6 + 1] compact column -> column of the image
    typedef __attribute__((address_space(3))) int gram_lds_int;
    volatile gram_lds_int* sMask = (volatile gram_lds_int*)(gram_lds_int*)(int*)(smem + (size_t)GRAM_ROWS * LD + 128);   // tiles the track touches
    constexpr int MAXP = (GRAM_MAX_NT * (GRAM_MAX_NT + 1) / 2 + GRAM_WAVES - 1) / GRAM_WAVES;     // pairs per wavefront
    v4d acc[MAXP];
#pragma unroll
    for (int j = 0; j < MAXP; ++j) acc[j] = v4d{0.0, 0.0, 0.0, 0.0};
    // my pairs: index pi = wv + 16 j in the row-major enumeration of ta <= tb
    int pa[MAXP], pb[MAXP];
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
        int pi = wv + GRAM_WAVES * j, ta = 0;
        if (pi < npairs) { while (pi >= nt - ta) { pi -= nt - ta; ++ta; } pa[j] = ta; pb[j] = ta + pi; }
        else { pa[j] = -1; pb[j] = -1; }
    }
    for (int fi = blockIdx.x; fi < p.nf; fi += gridDim.x) {
        const int f = p.f0 + fi;
        if (p.accepted[f] != 1) continue;                       // (uniform) gate-rejected / not selected: no rows
        const int v0 = p.view_ptr[f], M = p.view_ptr[f + 1] - v0;
        const int q = 2 * M - p.rank[f], ldb = 6 * M + 1, kq = (q + 3) >> 2;
        __syncthreads();                                        // the previous image has been read
        for (int e = t; e < GRAM_ROWS * LD; e += 64 * GRAM_WAVES) Hd[e] = 0.0;
        if (t <= 6 * M) sCol[t] = (t == 6 * M) ? p.dc : 6 * p.obs_slot[v0 + t / 6] + t % 6;
        if (t == 0) {
            int m = 1 << (p.dc >> 4);
            for (int v = 0; v < M; ++v) { const int c0 = 6 * p.obs_slot[v0 + v]; m |= (1 << (c0 >> 4)) | (1 << ((c0 + 5) >> 4)); }
            sMask[0] = m;
        }
        __syncthreads();
        const double* blk = p.stack + p.blk_off[f];
        for (int e = t; e < q * ldb; e += 64 * GRAM_WAVES) {
            const int L = e / ldb, c = e - L * ldb;
            Hd[L * LD + sCol[c]] = blk[e];
        }
        __syncthreads();
        const int mask = sMask[0];
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            if (pa[j] >= 0 && ((mask >> pa[j]) & (mask >> pb[j]) & 1)) {
                const double* ha = Hd + g * LD + 16 * pa[j] + cc;
                const double* hb = Hd + g * LD + 16 * pb[j] + cc;
                v4d a = acc[j];
                for (int u = 0; u < kq; ++u) a = __builtin_amdgcn_mfma_f64_16x16x4f64(ha[4 * u * LD], hb[4 * u * LD], a, 0, 0, 0);
                acc[j] = a;
            }
        }
    }
    double* out = p.part + (size_t)blockIdx.x * npairs * 256;
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
        const int pi = wv + GRAM_WAVES * j;
        if (pi < npairs) {
#pragma unroll
            for (int i = 0; i < 4; ++i) out[(size_t)pi * 256 + 64 * i + lane] = acc[j][i];
        }
    }
}

// Sum of the partial tiles in a fixed order (bit-reproducible), written to both halves of the symmetric n x n matrix S.
__global__ __launch_bounds__(256) void k_gram_reduce(const double* part, int nparts, int nt, int n, double* S, double* Uz) {
    const int npairs = nt * (nt + 1) / 2;
    int pi = blockIdx.x, ta = 0;
    while (pi >= nt - ta) { pi -= nt - ta; ++ta; }
    const int tb = ta + pi;
    const int t = threadIdx.x, lane = t & 63, i = t >> 6;
    double s = 0.0;
    for (int k = 0; k < nparts; ++k) s += part[((size_t)k * npairs + blockIdx.x) * 256 + t];
    // element (m = g + 4 i, nn = c) of tile (ta, tb): G[16 ta + m][16 tb + nn]
    const int r = 16 * ta + (lane >> 4) + 4 * i, c = 16 * tb + (lane & 15);
    if (r < n && c < n) {
        S[(size_t)r * n + c] = s;
        S[(size_t)c * n + r] = s;
        if (Uz) { Uz[(size_t)r * n + c] = 0.0; Uz[(size_t)c * n + r] = 0.0; }     // (the factor's buffer: k_chol16 fills its upper triangle)
    }
}

}  // namespace msckf
