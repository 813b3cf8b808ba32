// K1-K3 (+K4 write): one wavefront per tracked feature.
//
//   K1  per-view residual and Jacobians           reference MSCKF.py:505-544, Camera.py:54-67
//   K2  left-nullspace projection of the H_f block reference MSCKF.py:554-559
//   K3  chi-square gate against the prior P        reference MSCKF.py:561-568
//   K4  the accepted block [H_o | r_o] is written to the stack in compact form
//       (only the 6M clone columns the track touches)  reference MSCKF.py:581-588
//
// Lane L < 2M owns measurement row L (view L>>1, image axis L&1).  The feature
// Jacobian H_f (2M x 3) is reduced by three column-pivoted Householder
// reflectors held in registers (one V row per lane, wavefront reductions over
// DPP); with Q^T = I - V T^T V^T the projected clone Jacobian is
//     H_o = D - V Z,   Z = T^T (V^T D)   (3 x 6M)
// where D is the block-diagonal 2M x 6M matrix of the per-view 2x6 blocks, so
// V^T D needs only a 2-lane exchange per view.  The gate matrix
// S = H_o P_sub H_o^T + sigma^2 I is built from E = H_o P_sub with lanes over
// the 6M columns (each P_sub column is read once), then eliminated in LDS with
// the residual appended as an extra row, which leaves -gamma in the corner.
#pragma once
#include <hip/hip_runtime.h>
#include "wave_ops.h"
#include "feature_dpp_groups.h"

namespace msckf {

// ---- long tracks: a two-level nullspace basis (round 5) -----------------------------------------------------------------
// delta x and P+ do not depend on WHICH orthonormal basis of null(H_f^T) projects a track (reference MSCKF.py:554-559 takes
// scipy's).  A track that spans more than SPLIT_GSLOTS clone slots is cut into groups of views, each within
// SPLIT_GSLOTS slots, and Q is built in two levels, Q = Q_1 Q_2:
//   level 1: Q_1 = blockdiag(Q_1g), Q_1g^T H_f,g = [R_g; 0] -- three Householder reflectors over group g's rows alone.
//            Rows 3.. of Q_1g^T [H_x,g | r_g] are projected measurements that touch ONLY the group's clone slots: an
//            ordinary track of <= 10 views to the 60-column band pipeline (its "narrow block", q_g = 2 M_g - 3 rows);
//   level 2: the 3 carry rows of every group ([R_g | their rows of Q_1g^T H_x | r]) are stacked and reduced by three
//            more (column-pivoted) reflectors: the rows that remain, 3 (groups - 1) of them, touch every slot of the track
//            -- the "remainder block", a few rows per long track instead of 2 M - 3.
// A 30-view track becomes 3 x 17 narrow rows + 6 remainder rows; the gate (K3) is taken over all of them together and does
// not depend on the basis either, so it runs on the one-level basis as before.
constexpr int SPLIT_MAXG = 6;            // groups per long track
constexpr int SPLIT_GSLOTS = 10;         // clone slots a group may span
struct __attribute__((aligned(8))) SplitRec {
    int child[SPLIT_MAXG];               // block index (K5 order) of group g's narrow block; -1: a one-view group has none
    int wide;                            // block index of the remainder block
    unsigned char gv[SPLIT_MAXG + 1];    // group g = views [gv[g], gv[g + 1]) of the track (views in slot order)
    unsigned char ng;                    // groups (>= 2)
    int rows_cap;                        // 3 ng: the rows the remainder block may hold (its place in the dense matrix is found by k_rem_scatter)
};

struct FeatureArgs {
    int F;                       // features in this launch
    int f0;                      // ... sorted features [f0, f0 + F): workgroup b takes feature f0 + b
    int ldp;                     // leading dimension of P
    const int* view_ptr;         // [F+1] CSR, in SORTED feature order
    const double* obs_uv;        // [sumM*2]
    const int* obs_slot;         // [sumM]
    const double* idp_base;      // [F*3]
    const double* idp_m;         // [F*3]
    const double* idp_rho;       // [F]
    const double* cam_R;         // [N*9] R_W_Ci
    const double* cam_t;         // [N*3]
    const double* cam_R0;        // [N*9] null-state
    const double* cam_t0;        // [N*3]
    const double* P;             // [d*ldp]
    const double* chi2;          // [n_chi2]
    int n_chi2;
    double g[3];
    double Kinv[9];
    double sigma2;
    const long long* blk_off;    // [F] offset (scalars) of the feature's stack block
    void* stack;                 // blocks: row-major, q = 2M - rank rows x (6M+1) columns [H_o | r_o] (capacity 2M rows)
    int stack_f32;               // scalars of the stack: 0 = double, 1 = float (MSCKF_DTYPE_F32)
    int* rank;                   // [F] rank of H_f (rows < rank are not part of the projection)
    unsigned char* accepted;     // [F] (sorted order): 1 accepted, 0 gate-rejected, 2 not-SPD, 3 not selected
    const unsigned char* select; // optional [F] flags of k_select: features without bit 0 are skipped
    double* gamma;               // [F]
    long long* stamps;           // optional diagnostics (8 per feature), may be null
    long long zero_idx;          // index (scalars) of the 8 zero words behind the last block (k_lsweep loads them for absent entries)
    const SplitRec* split;       // k_feature<64, true>: one record per feature of the launch (long tracks)
    int* rank_h;                 // optional mirrors of rank / accepted in pinned host memory (the one-shot call: the gate results
    unsigned char* acc_h;        //   are on the host when the stream has drained, without a copy command behind K7)
};

// The 6M columns of a track's clone block are worked on in chunks of whole views, at most 64 columns each (one
// lane per column): views per chunk / chunks for a track of M views.
__host__ __device__ inline int feature_chunk_views(int M) {
    // (whole views: a chunk takes at most 10 of them -- 6 M / 64 chunks are one too few where M is 1 (mod 10) beyond 20: 21 views in two
    //  chunks made one of 11 views = 66 columns on 64 lanes, and the gate lost the last two columns of view 10 -- round 4, found by
    //  tests/test_gpu_hostpath.py::test_upload_paths_at_their_boundaries through a 300 px outlier the gate let pass)
    const int a = (6 * M + 63) / 64, b = (M + 9) / 10;
    const int nch = a > b ? a : b;
    return (M + nch - 1) / nch;
}
// LDS doubles needed for a track of M views.  Two layouts (k_feature<RMAX>):
//   all columns at once (launches whose tracks have at most 15 views): slots, D rows, V, Z, E (R2 x (6M+1)), E Z^T; the
//   elimination's staging tile reuses E (13.2 KB at 10 views: 12 wavefronts per CU; 27 KB at 15 views: 5);
//   column chunks (longer tracks): slots, D rows, V, Z, one chunk of E / H_o (+ the rhs column), r_o -- S lives in registers.
__host__ __device__ inline int feature_lds_doubles(int M, bool chunked) {
    const int R2 = 2 * M, C6 = 6 * M;
    const int head = (M + 2) / 2 + R2 * 6 + R2 * 3 + 3 * C6;
    if (!chunked) {
        const int stage = R2 * (6 * feature_chunk_views(M) + 1), elim = (R2 + 1) * (R2 + 3);   // one chunk of E | the elimination's tile (reuses it)
        return head + (stage > elim ? stage : elim) + 3 * R2 + 8;               // + E Z^T (behind both)
    }
    const int stage = R2 * (6 * feature_chunk_views(M) + 1) + (R2 + 2), elim = (R2 + 1) * (R2 + 3);   // (the elimination's tile takes the staging area's place)
    return head + (stage > elim ? stage : elim) + 8;
}

// ... of the split form (k_feature<64, true>): the chunked layout with a 61-column staging tile (a group has up to 10 views) +
// V1, V2 rows, Z1, Z2, the view -> group map and the carry list
// nwv > 1: that many wavefronts per track, each with a staging tile of its own (the first also holds the sum of the partial gate
// matrices and the elimination's tile)
__host__ __device__ inline int feature_split_estride(int M) {
    const int R2 = 2 * M;
    const int cv = feature_chunk_views(M), ld = 6 * (cv > SPLIT_GSLOTS ? cv : SPLIT_GSLOTS) + 1;
    const int stage = R2 * ld, elim = (R2 + 1) * (R2 + 3);
    return (stage > elim ? stage : elim) + 8;
}
__host__ __device__ inline int feature_split_lds_doubles(int M, int nwv = 1) {
    const int R2 = 2 * M, C6 = 6 * M;
    const int head = (M + 2) / 2 + R2 * 6 + R2 * 3 + 3 * C6;
    const int cv = feature_chunk_views(M), ld = 6 * (cv > SPLIT_GSLOTS ? cv : SPLIT_GSLOTS) + 1;
    const int tail = (R2 + 2) + 8 + 6 * R2 + 6 * C6 + (M + 3 * SPLIT_MAXG + 4) / 2 + 2;
    if (nwv > 1) return head + nwv * feature_split_estride(M) + tail;
    return head + R2 * ld + tail;
}

// The workgroup IS one wavefront (64 threads): its LDS instructions execute in program order, so a phase boundary needs no
// s_barrier and, above all, no s_waitcnt vmcnt(0) -- __syncthreads() drained the P_sub reads the gate has in flight across
// K4.  What is left is keeping the compiler from moving LDS accesses across the boundary.
__device__ __forceinline__ void wave_sync() { asm volatile("" ::: "memory"); }

template <int V> struct FTag { static constexpr int value = V; };

// The elimination's rank-1 update of NJ column quads in ONE statement (one s_nop 1 for the DPP hazard instead of one per FMA):
// eb_j += (lane LK of ea_j's row) * wB, then ea_j += (lane LK of ea_j's row) * wA -- ea_j, the broadcast source, last.
#define MSCKF_FD "row_newbcast:%c[lk] row_mask:0xf bank_mask:0xf\n\t"
template <int LK> __device__ __forceinline__ void elim_pairs(double (&b)[1], double (&a)[1], double wB, double wA) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %[wb] " MSCKF_FD "v_fmac_f64_dpp %1, %1, %[wa] " MSCKF_FD
                 : "+v"(b[0]), "+v"(a[0]) : [wb] "v"(wB), [wa] "v"(wA), [lk] "i"(LK));
}
template <int LK> __device__ __forceinline__ void elim_pairs(double (&b)[2], double (&a)[2], double wB, double wA) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %2, %[wb] " MSCKF_FD "v_fmac_f64_dpp %1, %3, %[wb] " MSCKF_FD
                 "v_fmac_f64_dpp %2, %2, %[wa] " MSCKF_FD "v_fmac_f64_dpp %3, %3, %[wa] " MSCKF_FD
                 : "+v"(b[0]), "+v"(b[1]), "+v"(a[0]), "+v"(a[1]) : [wb] "v"(wB), [wa] "v"(wA), [lk] "i"(LK));
}
template <int LK> __device__ __forceinline__ void elim_pairs(double (&b)[3], double (&a)[3], double wB, double wA) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %3, %[wb] " MSCKF_FD "v_fmac_f64_dpp %1, %4, %[wb] " MSCKF_FD "v_fmac_f64_dpp %2, %5, %[wb] " MSCKF_FD
                 "v_fmac_f64_dpp %3, %3, %[wa] " MSCKF_FD "v_fmac_f64_dpp %4, %4, %[wa] " MSCKF_FD "v_fmac_f64_dpp %5, %5, %[wa] " MSCKF_FD
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) : [wb] "v"(wB), [wa] "v"(wA), [lk] "i"(LK));
}
template <int LK> __device__ __forceinline__ void elim_pairs(double (&b)[4], double (&a)[4], double wB, double wA) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %4, %[wb] " MSCKF_FD "v_fmac_f64_dpp %1, %5, %[wb] " MSCKF_FD "v_fmac_f64_dpp %2, %6, %[wb] " MSCKF_FD
                 "v_fmac_f64_dpp %3, %7, %[wb] " MSCKF_FD
                 "v_fmac_f64_dpp %4, %4, %[wa] " MSCKF_FD "v_fmac_f64_dpp %5, %5, %[wa] " MSCKF_FD "v_fmac_f64_dpp %6, %6, %[wa] " MSCKF_FD
                 "v_fmac_f64_dpp %7, %7, %[wa] " MSCKF_FD
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])
                 : [wb] "v"(wB), [wa] "v"(wA), [lk] "i"(LK));
}
// ... and of the lower row slot alone (pivots 16+): eb_j += (lane LK of eb_j's row) * wB
template <int LK> __device__ __forceinline__ void elim_self(double (&b)[1], double wB) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[wb] " MSCKF_FD : "+v"(b[0]) : [wb] "v"(wB), [lk] "i"(LK));
}
template <int LK> __device__ __forceinline__ void elim_self(double (&b)[2], double wB) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[wb] " MSCKF_FD "v_fmac_f64_dpp %1, %1, %[wb] " MSCKF_FD
                 : "+v"(b[0]), "+v"(b[1]) : [wb] "v"(wB), [lk] "i"(LK));
}
template <int LK> __device__ __forceinline__ void elim_self(double (&b)[3], double wB) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[wb] " MSCKF_FD "v_fmac_f64_dpp %1, %1, %[wb] " MSCKF_FD "v_fmac_f64_dpp %2, %2, %[wb] " MSCKF_FD
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]) : [wb] "v"(wB), [lk] "i"(LK));
}
template <int LK> __device__ __forceinline__ void elim_self(double (&b)[4], double wB) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %[wb] " MSCKF_FD "v_fmac_f64_dpp %1, %1, %[wb] " MSCKF_FD "v_fmac_f64_dpp %2, %2, %[wb] " MSCKF_FD
                 "v_fmac_f64_dpp %3, %3, %[wb] " MSCKF_FD
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : [wb] "v"(wB), [lk] "i"(LK));
}
// ... and of a row slot whose source is ANOTHER register set: d_j += (lane LK of a_j's row) * w  (k_feature<64>: four row slots)
template <int LK> __device__ __forceinline__ void elim_from(double (&d)[1], const double (&a)[1], double w) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %[w] " MSCKF_FD : "+v"(d[0]) : "v"(a[0]), [w] "v"(w), [lk] "i"(LK));
}
template <int LK> __device__ __forceinline__ void elim_from(double (&d)[2], const double (&a)[2], double w) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %2, %[w] " MSCKF_FD "v_fmac_f64_dpp %1, %3, %[w] " MSCKF_FD
                 : "+v"(d[0]), "+v"(d[1]) : "v"(a[0]), "v"(a[1]), [w] "v"(w), [lk] "i"(LK));
}
template <int LK> __device__ __forceinline__ void elim_from(double (&d)[3], const double (&a)[3], double w) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %3, %[w] " MSCKF_FD "v_fmac_f64_dpp %1, %4, %[w] " MSCKF_FD "v_fmac_f64_dpp %2, %5, %[w] " MSCKF_FD
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), [w] "v"(w), [lk] "i"(LK));
}
template <int LK> __device__ __forceinline__ void elim_from(double (&d)[4], const double (&a)[4], double w) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %4, %[w] " MSCKF_FD "v_fmac_f64_dpp %1, %5, %[w] " MSCKF_FD "v_fmac_f64_dpp %2, %6, %[w] " MSCKF_FD
                 "v_fmac_f64_dpp %3, %7, %[w] " MSCKF_FD
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), [w] "v"(w), [lk] "i"(LK));
}
#undef MSCKF_FD
// column quads [J0, J0 + NJ) of a 16-quad row slot, four per statement
template <int LK, int J0, int NJ>
__device__ __forceinline__ void elim_from16(double (&d)[16], const double (&a)[16], double w) {
    if constexpr (NJ > 0) {
        constexpr int N1 = NJ < 4 ? NJ : 4;
        double dd[N1], aa[N1];
#pragma unroll
        for (int i = 0; i < N1; ++i) { dd[i] = d[J0 + i]; aa[i] = a[J0 + i]; }
        elim_from<LK>(dd, aa, w);
#pragma unroll
        for (int i = 0; i < N1; ++i) d[J0 + i] = dd[i];
        elim_from16<LK, J0 + N1, NJ - N1>(d, a, w);
    }
}
template <int LK, int J0, int NJ>
__device__ __forceinline__ void elim_self16(double (&d)[16], double w) {
    if constexpr (NJ > 0) {
        constexpr int N1 = NJ < 4 ? NJ : 4;
        double dd[N1];
#pragma unroll
        for (int i = 0; i < N1; ++i) dd[i] = d[J0 + i];
        elim_self<LK>(dd, w);
#pragma unroll
        for (int i = 0; i < N1; ++i) d[J0 + i] = dd[i];
        elim_self16<LK, J0 + N1, NJ - N1>(d, w);
    }
}
template <int K, int KEND, typename Fn>
__device__ __forceinline__ void feature_static_for(Fn&& fn) {
    if constexpr (K < KEND) { fn(FTag<K>{}); feature_static_for<K + 1, KEND>(fn); }
}
// RMAX > 2 * max track length of the launch (rows of the gate matrix held per lane).
// NWV > 1 (split form only): NWV wavefronts per track -- every wavefront runs K1 / K2 (cheap, all in registers), the view groups
// (K4) and the gate's column chunks are dealt over the wavefronts, the partial gate matrices are summed through LDS and wavefront
// 0 eliminates.  For FEW long tracks: one wavefront per track is one wavefront's latency (140 us at 30 views) in front of the
// band pipeline's leaves.
template <int RMAX, bool SPLIT = false, int NWV = 1>
__global__ __launch_bounds__(64 * NWV, NWV > 1 ? 1 : (RMAX <= 24 ? 3 : 2)) void k_feature(FeatureArgs p) {      // (<24>: <= 168 registers, the LDS footprint allows 12 wavefronts per CU; <32>: 219 registers, 8 per CU -- bounded to 168 it spills and is slower)
    // CHUNKED: the gate works on column chunks (tracks of 16+ views, k_feature<64>).  Tracks of up to 15 views keep all 6M
    // columns of E in LDS: 27 KB at 15 views -- five wavefronts per CU instead of the chunked form's nine, and still the
    // faster one since the all-columns form was rebuilt in round 3 (per block at 15 views: 61 us chunked).
    constexpr bool CHUNKED = RMAX > 32;
    static_assert(!SPLIT || CHUNKED, "the split form is an instance of the chunked kernel");
    static_assert(NWV == 1 || SPLIT, "several wavefronts per track: the split form only");
    constexpr bool ONE_CHUNK = RMAX <= 24;        // k_feature<24>: tracks of up to 10 views, all 60 columns in one chunk
    constexpr int MAXVK = ONE_CHUNK ? 10 : (RMAX - 2) / 2;   // longest track of this instance
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int f = blockIdx.x + p.f0;
    const int lane = threadIdx.x & 63, wv = NWV > 1 ? (int)(threadIdx.x >> 6) : 0;
    const bool wfirst = wv == 0;                   // the wavefront that writes what all of them computed alike
    auto xsync = [&]() { if constexpr (NWV > 1) __syncthreads(); else wave_sync(); };
    const int v0 = p.view_ptr[f];
    const int M = p.view_ptr[f + 1] - v0;
    const int R2 = 2 * M, C6 = 6 * M;
    int* sSlot = reinterpret_cast<int*>(smem);                  // [M] clone slot of every view
    double* sA = smem + (M + 2) / 2;       // [R2][6]   OC-projected clone block rows (D)
    double* sV = sA + R2 * 6;              // [R2][3]   Householder vectors
    double* sZ = sV + R2 * 3;              // [3][C6]

    if (f == 0 && lane < 8 && wfirst) {        // the stack's zero words (saves the host a memset launch in front of this kernel)
        if (p.stack_f32) static_cast<float*>(p.stack)[p.zero_idx + lane] = 0.0f;
        else static_cast<double*>(p.stack)[p.zero_idx + lane] = 0.0;
    }
    // sums over the wavefront of values that are zero outside the 2M row lanes: two 16-lane rows are enough up to 16 views
    auto wsum = [&](double x) -> double {
        if constexpr (RMAX <= 34) {
            x = row16_sum(x);
            return readlane_d(x, 0) + readlane_d(x, 16);
        } else {
            return wave_sum(x);
        }
    };
    if (p.select && !(p.select[f] & 1)) {      // not in valid_features (MSCKF.py:453-455): no rows, not a rejection
        if (lane == 0 && wfirst) {
            p.rank[f] = 0; p.gamma[f] = 0.0; p.accepted[f] = 3;
            if (p.acc_h) { p.rank_h[f] = 0; p.acc_h[f] = 3; }
            if constexpr (SPLIT) {
                const SplitRec& sr = p.split[blockIdx.x];
                for (int g = 0; g < sr.ng; ++g) if (sr.child[g] >= 0) { p.rank[sr.child[g]] = 3; p.accepted[sr.child[g]] = 3; }
                p.rank[sr.wide] = 0; p.accepted[sr.wide] = 3;
            }
        }
        return;
    }
    long long tq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (p.stamps) tq[0] = wall_clock64();
    // ---------------- K1: one measurement row per lane -----------------------
    double res = 0.0, a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, h0 = 0, h1 = 0, h2 = 0;
    const int view = lane >> 1;
    if (lane < R2) {
        const int o = v0 + view;
        const int s = p.obs_slot[o];
        if ((lane & 1) == 0 && wfirst) sSlot[view] = s;
        const double* R = p.cam_R + 9 * s;
        const double* t = p.cam_t + 3 * s;
        const double* R0 = p.cam_R0 + 9 * s;
        const double* t0 = p.cam_t0 + 3 * s;
        const double rho = p.idp_rho[f];
        const double wx = rho * (p.idp_base[3 * f + 0] - t[0]) + p.idp_m[3 * f + 0];
        const double wy = rho * (p.idp_base[3 * f + 1] - t[1]) + p.idp_m[3 * f + 1];
        const double wz = rho * (p.idp_base[3 * f + 2] - t[2]) + p.idp_m[3 * f + 2];
        // Ci_f = R^T w   (MSCKF.py:516)
        const double px = R[0] * wx + R[3] * wy + R[6] * wz;
        const double py = R[1] * wx + R[4] * wy + R[7] * wz;
        const double pz = R[2] * wx + R[5] * wy + R[8] * wz;
        // W_f = R Ci_f + t (MSCKF.py:517)
        const double Wx = R[0] * px + R[1] * py + R[2] * pz + t[0];
        const double Wy = R[3] * px + R[4] * py + R[5] * pz + t[1];
        const double Wz = R[6] * px + R[7] * py + R[8] * pz + t[2];
        const double u = p.obs_uv[2 * o], v = p.obs_uv[2 * o + 1];
        const double zx = p.Kinv[0] * u + p.Kinv[1] * v + p.Kinv[2];
        const double zy = p.Kinv[3] * u + p.Kinv[4] * v + p.Kinv[5];
        const double zw = p.Kinv[6] * u + p.Kinv[7] * v + p.Kinv[8];
        // (one division each for 1 / pz, 1 / zw and 1 / den; the other quotients of Camera.py are products with them: an IEEE
        //  division is ~15 instructions here, and ~1e-16 relative is what the products differ by)
        const double ia = 1.0 / pz;
        const double jb = -px * (ia * ia);
        const double jc = -py * (ia * ia);
        const double izw = 1.0 / zw;
        // u = [R0^T g ; (W_f - t0) x g]   (MSCKF.py:528-530)
        const double gx = p.g[0], gy = p.g[1], gz = p.g[2];
        const double u0 = R0[0] * gx + R0[3] * gy + R0[6] * gz;
        const double u1 = R0[1] * gx + R0[4] * gy + R0[7] * gz;
        const double u2 = R0[2] * gx + R0[5] * gy + R0[8] * gz;
        const double ex = Wx - t0[0], ey = Wy - t0[1], ez = Wz - t0[2];
        const double u3 = ey * gz - ez * gy;
        const double u4 = ez * gx - ex * gz;
        const double u5 = ex * gy - ey * gx;
        const double den = u0 * u0 + u1 * u1 + u2 * u2 + u3 * u3 + u4 * u4 + u5 * u5;
        double J0, J1, J2;   // this lane's row of J (Camera.py:57-58)
        if ((lane & 1) == 0) {
            res = zx * izw - px * ia;
            J0 = ia; J1 = 0.0; J2 = jb;
            a0 = -jb * py; a1 = -ia * pz + jb * px; a2 = ia * py;          // J skew(Ci_f), row 0
        } else {
            res = zy * izw - py * ia;
            J0 = 0.0; J1 = ia; J2 = jc;
            a0 = ia * pz - jc * py; a1 = jc * px; a2 = -ia * px;           // row 1
        }
        // H_f = J R^T (this row);  H_x[:,3:] = -H_f   (Camera.py:62,66; MSCKF.py:536)
        h0 = J0 * R[0] + J1 * R[1] + J2 * R[2];
        h1 = J0 * R[3] + J1 * R[4] + J2 * R[5];
        h2 = J0 * R[6] + J1 * R[7] + J2 * R[8];
        a3 = -h0; a4 = -h1; a5 = -h2;
        if (den > 1e-6) {      // observability-constrained projection (MSCKF.py:532-534)
            const double au = a0 * u0 + a1 * u1 + a2 * u2 + a3 * u3 + a4 * u4 + a5 * u5;
            const double aud = au * (1.0 / den);
            a0 -= aud * u0; a1 -= aud * u1; a2 -= aud * u2;
            a3 -= aud * u3; a4 -= aud * u4; a5 -= aud * u5;
        }
    }

    if (p.stamps) tq[1] = wall_clock64();
    // ---------------- K2: column-pivoted Householder QR of H_f ---------------
    // After step k the lane k holds R_kk; rows >= rank span the left null space.
    double vv0 = 0, vv1 = 0, vv2 = 0;       // this lane's row of V
    double beta0 = 0, beta1 = 0, beta2 = 0;
    int rank = 0;
    double r00 = 0.0;
    const double tol_scale = 2.220446049250313e-16 * (double)(R2 > 3 ? R2 : 3);
    {
        double c0 = h0, c1 = h1, c2 = h2;   // working columns (rows >= k)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k >= R2) break;
            const bool act = (lane >= k) && (lane < R2);
            // column norms of the remaining columns over rows >= k
            double n0 = wsum(act ? c0 * c0 : 0.0);
            double n1 = (k < 2) ? wsum(act ? c1 * c1 : 0.0) : -1.0;
            double n2 = (k < 1) ? wsum(act ? c2 * c2 : 0.0) : -1.0;
            // pivot: bring the largest remaining column to the front (c0)
            if (n1 > n0 && n1 >= n2) { double t_ = c0; c0 = c1; c1 = t_; n0 = n1; }
            else if (n2 > n0 && n2 > n1) { double t_ = c0; c0 = c2; c2 = t_; n0 = n2; }
            // (rsqrt / rcp seeds + one Newton step each, as in the sweeps: a few 1e-16; a zero column gives NaN and leaves below)
            const double ry = fast_rsqrt(n0);
            const double nrm = n0 * ry;
            if (k == 0) r00 = nrm;
            if (!(nrm > tol_scale * r00) || nrm == 0.0) break;   // numerically rank deficient
            const double xk = lane_bcast(c0, k);
            const double alpha = (xk > 0.0) ? -nrm : nrm;
            const double vk = act ? ((lane == k) ? (xk - alpha) : c0) : 0.0;
            const double beta = ry * fast_rcp(nrm + fabs(xk));   // 1 / (nrm (nrm + |xk|))
            // apply to the remaining columns
            if (k < 2) {
                const double d1 = wsum(vk * (act ? c1 : 0.0));
                if (act) c1 -= beta * d1 * vk;
            }
            if (k < 1) {
                const double d2 = wsum(vk * (act ? c2 : 0.0));
                if (act) c2 -= beta * d2 * vk;
            }
            if (k == 0) { vv0 = vk; beta0 = beta; }
            else if (k == 1) { vv1 = vk; beta1 = beta; }
            else { vv2 = vk; beta2 = beta; }
            rank = k + 1;
            // shift: next step works on (c1, c2)
            c0 = c1; c1 = c2; c2 = 0.0;
        }
    }
    // T of the compact WY form Q = I - V T V^T (forward, columnwise)
    const double v01 = wsum(vv0 * vv1);
    const double v02 = wsum(vv0 * vv2);
    const double v12 = wsum(vv1 * vv2);
    const double T00 = beta0, T11 = beta1, T22 = beta2;
    const double T01 = -beta1 * T00 * v01;
    const double T02 = -beta2 * (T00 * v02 + T01 * v12);
    const double T12 = -beta2 * T11 * v12;

    // residual: r_o = r - V T^T V^T r
    const double wr0 = wsum(vv0 * res), wr1 = wsum(vv1 * res), wr2 = wsum(vv2 * res);
    const double zr0 = T00 * wr0;
    const double zr1 = T01 * wr0 + T11 * wr1;
    const double zr2 = T02 * wr0 + T12 * wr1 + T22 * wr2;
    const double ro = res - (vv0 * zr0 + vv1 * zr1 + vv2 * zr2);

    // W = V^T D (2-lane exchange per view), Z = T^T W
    {
        const double av[6] = {a0, a1, a2, a3, a4, a5};
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double w0 = vv0 * av[a], w1 = vv1 * av[a], w2 = vv2 * av[a];
            w0 += dpp_move<0xB1>(w0);
            w1 += dpp_move<0xB1>(w1);
            w2 += dpp_move<0xB1>(w2);
            if (lane < R2 && (lane & 1) == 0 && wfirst) {
                const int c = 6 * view + a;
                sZ[c] = T00 * w0;
                sZ[C6 + c] = T01 * w0 + T11 * w1;
                sZ[2 * C6 + c] = T02 * w0 + T12 * w1 + T22 * w2;
            }
        }
        if (lane < R2 && wfirst) {
#pragma unroll
            for (int a = 0; a < 6; ++a) sA[lane * 6 + a] = av[a];
            sV[lane * 3 + 0] = vv0; sV[lane * 3 + 1] = vv1; sV[lane * 3 + 2] = vv2;
        }
    }
    xsync();

    if (p.stamps) tq[2] = wall_clock64();
    const int q = R2 - rank;
    double srow[RMAX];
    int n_wide_rows = 0;                   // SPLIT: rows of the track's remainder block
    if constexpr (!CHUNKED) {
    // Tracks of up to 15 views (k_feature<24>: 11, <32>: 15).  K4 and the gate run over column chunks of whole views, at most
    // 60 columns = one per lane (one chunk up to 10 views, two from 11 on): sE holds the chunk's columns of H_o (K4 staging),
    // then of E = H_o P_sub; S never exists in LDS before the elimination's tile, which takes sE's place at the end.
    const int CV = ONE_CHUNK ? M : feature_chunk_views(M);   // views per column chunk
    const int ldE = 6 * CV + 1;
    const int ldb = C6 + 1;
    double* sE = sZ + 3 * C6;              // [R2][ldE]
    // Z, the D rows and V once more in REGISTERS, 16 entries per register and every 16-lane row holding the same 16: the
    // lane-invariant factor of an FMA then comes by the DPP row broadcast (feature_dpp_groups.h) instead of one LDS
    // broadcast read per FMA -- ~460 of the ~700 LDS instructions of a 10-view block were such reads.
    double zq[3][FEAT_ZR], aq[FEAT_AR], vq[FEAT_VR];
    {
        const int l15 = lane & 15;
        constexpr int ZRU = (6 * MAXVK + 15) / 16, ARU = (12 * MAXVK + 15) / 16, VRU = (6 * MAXVK + 15) / 16;   // registers in use
        static_assert(ZRU <= FEAT_ZR && ARU <= FEAT_AR && VRU <= FEAT_VR && MAXVK <= FEAT_DPP_MAXV, "tables of feature_dpp_groups.h");
#pragma unroll
        for (int i = 0; i < ZRU; ++i) {
            const int cz = min(16 * i + l15, C6 - 1);
            zq[0][i] = sZ[cz]; zq[1][i] = sZ[C6 + cz]; zq[2][i] = sZ[2 * C6 + cz];
        }
#pragma unroll
        for (int i = 0; i < ARU; ++i) aq[i] = sA[min(16 * i + l15, R2 * 6 - 1)];
#pragma unroll
        for (int i = 0; i < VRU; ++i) vq[i] = sV[min(16 * i + l15, R2 * 3 - 1)];
    }
    // x[i] -= V[L0 + i] . (w0, w1, w2) for the rows of one group
    auto corr = [&](auto tagl, double (&x)[4], double w0, double w1, double w2) {
        if constexpr (decltype(tagl)::value < 2 * MAXVK) FeatCorrRows<decltype(tagl)::value>::run(x, w0, w1, w2, vq);
    };
    // the upper triangle of S, dealt round-robin over the lanes: rows L and R2 - 1 - L together hold R2 + 1 entries j >= L, so
    // entry idx of the R2 / 2 row pairs is one division by R2 + 1 away.  An entry's D part lies in ONE chunk (its view's).
    constexpr int NRES = ((RMAX / 2) * (RMAX + 1) + 63) / 64;      // R2 <= RMAX - 2, R2 even
    double res[NRES];
#pragma unroll
    for (int it = 0; it < NRES; ++it) res[it] = 0.0;
    const int ntri = (R2 >> 1) * (R2 + 1);
    const float inv_p = 1.0f / (float)(R2 + 1);
    auto entry = [&](int idx, int& L, int& j) {                  // idx < ntri -> (L, j), j >= L
        const int pr = (int)(((float)idx + 0.5f) * inv_p), q_ = idx - pr * (R2 + 1);
        const bool lo = q_ < R2 - pr;
        L = lo ? pr : R2 - 1 - pr;
        j = lo ? pr + q_ : q_ - 1;
    };
    double* sEz = sE + R2 * ldE;                                  // [R2][3] E Z^T, summed over the chunks
#ifndef MSCKF_FEAT_PD
#define MSCKF_FEAT_PD 4
#endif
    constexpr int PD = MSCKF_FEAT_PD;                             // views of lookahead of the P_sub column reads
    double pb[PD + 1][6];
    for (int vc0 = 0; vc0 < (ONE_CHUNK ? 1 : M); vc0 += (ONE_CHUNK ? 1 : CV)) {      // (provably one trip for k_feature<24>)
        const int nv = ONE_CHUNK ? M : min(CV, M - vc0);          // views of this chunk
        const int c0 = 6 * vc0, cw = 6 * nv;                      // its columns [c0, c0 + cw), cw <= 60
        const bool last = vc0 + nv >= M;
        const int cwp = cw + (last ? 1 : 0);                      // + the rhs column
        // ALL lanes run the DPP groups below (a DPP broadcast reads garbage from a source lane EXEC has switched off): lanes
        // past the chunk's last column work on a copy of it and do not store
        const bool cok = lane < cw;
        const int lc = min(lane, cw - 1), c = c0 + lc;
        const int vwc = c / 6, ac = c - 6 * vwc;
        const int colg = 15 + 6 * sSlot[vwc] + ac;
        // The gate's first pass walks column c of P_sub (L2 reads, ~0.6 us each way): its first PD views are requested HERE,
        // ahead of K4, the later ones PD views ahead of their use (a view's 30 FMAs last ~0.1 us; with one view of lookahead
        // the pass was ten load latencies long).
        auto fetch = [&](auto tagv) {
            constexpr int VW = decltype(tagv)::value;
            if (VW < M) {                                            // (uniform)
                const double* prow = p.P + (size_t)(15 + 6 * sSlot[VW]) * p.ldp + colg;
#pragma unroll
                for (int a = 0; a < 6; ++a) pb[VW % (PD + 1)][a] = prow[(size_t)a * p.ldp];
            }
        };
        fetch(FTag<0>{});
        if constexpr (PD > 1) fetch(FTag<1>{});
        if constexpr (PD > 2) fetch(FTag<2>{});
        if constexpr (PD > 3) fetch(FTag<3>{});
        static_assert(PD >= 1 && PD <= 4, "the prologue above issues PD views");
        // ---------------- K4: the chunk's columns of the compact block [H_o | r_o] ----------------
        // staged in sE (lane c holds Z[:, c] and walks the rows: H_o[L][c] = D[L][c] - V[L,:] Z[:, c], D[L][c] != 0 only for the
        // two rows of c's own view), then out as contiguous row segments of the stack block (row-major q x (6M + 1), the q
        // projected rows only)
        {
            const double z0 = sZ[c], z1 = sZ[C6 + c], z2 = sZ[2 * C6 + c];
            const double d0 = sA[(2 * vwc) * 6 + ac], d1 = sA[(2 * vwc + 1) * 6 + ac];   // the two rows of c's own view
            auto rows = [&](auto tagl) {
                constexpr int L0 = decltype(tagl)::value;
                if (L0 < R2) {
                    double x[4] = {0.0, 0.0, 0.0, 0.0};
                    corr(tagl, x, z0, z1, z2);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (L0 + i < R2 && cok) sE[(L0 + i) * ldE + lc] = x[i];
                }
            };
            rows(FTag<0>{}); rows(FTag<4>{}); rows(FTag<8>{}); rows(FTag<12>{}); rows(FTag<16>{}); rows(FTag<20>{});
            rows(FTag<24>{}); rows(FTag<28>{});
            if (cok) {                                               // + D: only the two rows of c's own view have an entry in column c
                sE[(2 * vwc) * ldE + lc] += d0;
                sE[(2 * vwc + 1) * ldE + lc] += d1;
            }
            if (last && lane < R2) sE[lane * ldE + cw] = ro;
            wave_sync();
            const int nel = q * cwp;
            if (cw == C6) {
                // one chunk: the q projected rows are ONE contiguous range of sE and of the stack block
                const double* srcrows = sE + rank * ldE;
                if (p.stack_f32) {
                    float* blk = static_cast<float*>(p.stack) + p.blk_off[f];
                    for (int e = lane; e < nel; e += 64) blk[e] = (float)srcrows[e];
                } else {
                    double* blk = static_cast<double*>(p.stack) + p.blk_off[f];
                    int e = lane;
                    for (; e + 192 < nel; e += 256) {             // four LDS reads in flight per trip
                        const double x0 = srcrows[e], x1 = srcrows[e + 64], x2 = srcrows[e + 128], x3 = srcrows[e + 192];
                        blk[e] = x0; blk[e + 64] = x1; blk[e + 128] = x2; blk[e + 192] = x3;
                    }
                    for (; e < nel; e += 64) blk[e] = srcrows[e];
                }
            } else {
                const float inv_cwp = 1.0f / (float)cwp;
                for (int e = lane; e < nel; e += 64) {
                    const int L = (int)(((float)e + 0.5f) * inv_cwp), cc = e - L * cwp;      // e / cwp, exact for e < 2^20
                    const double x = sE[(rank + L) * ldE + cc];
                    const long long dst = p.blk_off[f] + (long long)L * ldb + c0 + cc;
                    if (p.stack_f32) static_cast<float*>(p.stack)[dst] = (float)x;
                    else static_cast<double*>(p.stack)[dst] = x;
                }
            }
            wave_sync();                       // sE is rewritten by the gate below
        }
        if (p.stamps && vc0 == 0) tq[3] = wall_clock64();
        // ---------------- K3: gate.  Pass 1, one column c of P_sub per lane: E = D P_sub (block rows) - V (Z P_sub) -------------
        {
            double zp0 = 0, zp1 = 0, zp2 = 0;
            auto view = [&](auto tagv) {
                constexpr int VW = decltype(tagv)::value;
                if (VW < M) {                                            // (uniform)
                    fetch(FTag<VW + PD>{});
                    double e0 = 0.0, e1 = 0.0;
                    if constexpr (VW < MAXVK) FeatGateView<VW>::run(zp0, zp1, zp2, e0, e1, pb[VW % (PD + 1)], zq, aq);
                    if (cok) { sE[(2 * VW) * ldE + lc] = e0; sE[(2 * VW + 1) * ldE + lc] = e1; }
                }
            };
            view(FTag<0>{}); view(FTag<1>{}); view(FTag<2>{}); view(FTag<3>{}); view(FTag<4>{}); view(FTag<5>{});
            view(FTag<6>{}); view(FTag<7>{}); view(FTag<8>{}); view(FTag<9>{}); view(FTag<10>{});
            view(FTag<11>{}); view(FTag<12>{}); view(FTag<13>{}); view(FTag<14>{});
            // E -= V ZP  (same column, all rows)
            auto rows = [&](auto tagl) {
                constexpr int L0 = decltype(tagl)::value;
                if (L0 < R2) {
                    double x[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) x[i] = sE[min(L0 + i, R2 - 1) * ldE + lc];
                    corr(tagl, x, zp0, zp1, zp2);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (L0 + i < R2 && cok) sE[(L0 + i) * ldE + lc] = x[i];
                }
            };
            rows(FTag<0>{}); rows(FTag<4>{}); rows(FTag<8>{}); rows(FTag<12>{}); rows(FTag<16>{}); rows(FTag<20>{});
            rows(FTag<24>{}); rows(FTag<28>{});
        }
        wave_sync();
        if (p.stamps && vc0 == 0) tq[4] = wall_clock64();
        // ---------------- pass 2, ALL lanes (rounds 1-3 had one lane per row of S: 20 of 64 busy at 10 views) -----------------
        //   (a) the chunk's part of E Z^T, one (row, t) pair per lane;
        //   (b) the D part of the entries S[L][j], j >= L, whose view lies in this chunk: E[L, view of j] . D[j]
        for (int idx = lane; idx < 3 * R2; idx += 64) {
            const int L = idx / 3, t3 = idx - 3 * L;
            const double* er = sE + L * ldE;
            const double* zr = sZ + t3 * C6 + c0;
            double e0 = 0.0, e1 = 0.0, e2 = 0.0;
            for (int v6 = 0; v6 < cw; v6 += 6) {                 // a view's six columns at a time: twelve LDS reads in flight
                double ev[6], zv[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) { ev[a] = er[v6 + a]; zv[a] = zr[v6 + a]; }
                e0 += ev[0] * zv[0]; e1 += ev[1] * zv[1]; e2 += ev[2] * zv[2];
                e0 += ev[3] * zv[3]; e1 += ev[4] * zv[4]; e2 += ev[5] * zv[5];
            }
            sEz[idx] = (vc0 == 0 ? 0.0 : sEz[idx]) + ((e0 + e1) + e2);
        }
#pragma unroll
        for (int it = 0; it < NRES; ++it) {
            const int idx = lane + 64 * it;
            if (idx < ntri) {
                int L, j;
                entry(idx, L, j);
                const int vw = (j >> 1) - vc0;
                if (vw >= 0 && vw < nv) {
                    double sacc = 0.0;
#pragma unroll
                    for (int a = 0; a < 6; ++a) sacc += sE[L * ldE + 6 * vw + a] * sA[j * 6 + a];
                    res[it] = sacc;
                }
            }
        }
        wave_sync();                                                // the next chunk restages sE / the tile below replaces it
    }
    // S = (D parts) - (E Z^T) V^T + sigma^2 I, mirrored into the elimination's tile sT (which lies over sE) with the rhs border:
    // column R2 = r_o, row R2 = r_o^T, corner 0
    {
        double* sT = sE;
        const int ldT = R2 + 3;
#pragma unroll
        for (int it = 0; it < NRES; ++it) {
            const int idx = lane + 64 * it;
            if (idx < ntri) {
                int L, j;
                entry(idx, L, j);
                double x = res[it] - (sEz[L * 3 + 0] * sV[j * 3 + 0] + sEz[L * 3 + 1] * sV[j * 3 + 1] + sEz[L * 3 + 2] * sV[j * 3 + 2]);
                if (j == L) x += p.sigma2;
                res[it] = x;
            }
        }
        wave_sync();                                                // (sEz lies behind the tile: (R2 + 1)(R2 + 3) <= R2 ldE is not guaranteed)
#pragma unroll
        for (int it = 0; it < NRES; ++it) {
            const int idx = lane + 64 * it;
            if (idx < ntri) {
                int L, j;
                entry(idx, L, j);
                sT[L * ldT + j] = res[it];
                sT[j * ldT + L] = res[it];
            }
        }
        if (lane < R2) { sT[lane * ldT + R2] = ro; sT[R2 * ldT + lane] = ro; }
        if (lane == R2) sT[R2 * ldT + R2] = 0.0;
    }
    if (p.stamps) tq[5] = wall_clock64();
    } else {
    const int CV = feature_chunk_views(M);   // views per column chunk
    const int ldE = 6 * ((SPLIT && CV < SPLIT_GSLOTS) ? SPLIT_GSLOTS : CV) + 1;
    // [R2][ldE] one column chunk of H_o (K4 staging) / of E = H_o P_sub (gate): one per wavefront
    double* sE0 = sZ + 3 * C6;
    const int estride = NWV > 1 ? feature_split_estride(M) : 0;
    double* sE = sE0 + wv * estride;
    double* sRo = NWV > 1 ? sE0 + NWV * estride : sE0 + R2 * ldE;           // [R2]      r_o
    if constexpr (SPLIT) {
        // ---------------- K2 + K4 of a long track: two-level basis, G narrow blocks + one remainder block ----------------
        double* sU = sRo + (R2 + 2);           // [R2][3] level-1 reflectors (V1 rows)
        double* sW = sU + 3 * R2;              // [R2][3] level-2 reflectors (V2 rows, zero outside the carry rows)
        double* sZ1 = sW + 3 * R2;             // [3][C6] T1_g^T V1_g^T D, group g's columns
        double* sZ2 = sZ1 + 3 * C6;            // [3][C6] T2^T V2^T H1
        int* sGrp = reinterpret_cast<int*>(sZ2 + 3 * C6);      // [M] group of a view | [3 SPLIT_MAXG] the carry rows in order
        int* sCarry = sGrp + M;
        const SplitRec sr = p.split[blockIdx.x];
        const int ng = sr.ng;
        int grp = 0, g0v = 0;
        for (int g = 0; g < ng; ++g)
            if (view >= sr.gv[g] && view < sr.gv[g + 1]) { grp = g; g0v = sr.gv[g]; }
        const bool inr = lane < R2;
        const int lr = lane - 2 * g0v;         // row within the group
        if (inr && (lane & 1) == 0 && wfirst) sGrp[view] = grp;
        int ncarry = 0;
        for (int g = 0; g < ng; ++g) {
            const int rows = min(3, 2 * (sr.gv[g + 1] - sr.gv[g]));
            for (int i = 0; i < rows; ++i) { if (lane == 0 && wfirst) sCarry[ncarry] = 2 * sr.gv[g] + i; ++ncarry; }
        }
        // sums over the rows of ONE group, every lane receiving its own group's (ng <= 6 whole-wave sums)
        auto gsum = [&](double x) -> double {
            double mine = 0.0;
            for (int g = 0; g < ng; ++g) {
                const double sg = wave_sum(grp == g ? x : 0.0);
                if (grp == g) mine = sg;
            }
            return mine;
        };
        // level 1: three reflectors per group, no pivoting (any triangularisation of H_f,g serves)
        double hh[3] = {inr ? h0 : 0.0, inr ? h1 : 0.0, inr ? h2 : 0.0};
        double uu[3] = {0.0, 0.0, 0.0}, ub[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const bool act = inr && lr >= k;
            const double ck = hh[k];
            const double n0 = gsum(act ? ck * ck : 0.0);
            double xk = 0.0;                                    // the pivot row's entry, per group
            for (int g = 0; g < ng; ++g) {
                const double xg = readlane_d(ck, min(2 * sr.gv[g] + k, 63));
                if (grp == g) xk = xg;
            }
            const bool live = n0 > 0.0;                         // (a one-view group has no rows left at k = 2; a zero column: identity)
            const double ry = fast_rsqrt(live ? n0 : 1.0);
            const double nrm = n0 * ry;
            const double alpha = (xk > 0.0) ? -nrm : nrm;
            const double vk = (act && live) ? ((lr == k) ? (xk - alpha) : ck) : 0.0;
            const double beta = live ? ry * fast_rcp(nrm + fabs(xk)) : 0.0;
#pragma unroll
            for (int j = k + 1; j < 3; ++j) {
                const double dj = gsum(vk * hh[j]);
                hh[j] -= beta * dj * vk;
            }
            if (act && live) hh[k] = (lr == k) ? alpha : 0.0;
            uu[k] = vk; ub[k] = beta;
        }
        // T1 of the group (compact WY, as above) and the level-1 residual r1 = r - V1 T1^T V1^T r
        const double g01 = gsum(uu[0] * uu[1]), g02 = gsum(uu[0] * uu[2]), g12 = gsum(uu[1] * uu[2]);
        const double A00 = ub[0], A11 = ub[1], A22 = ub[2];
        const double A01 = -ub[1] * A00 * g01;
        const double A02 = -ub[2] * (A00 * g02 + A01 * g12);
        const double A12 = -ub[2] * A11 * g12;
        const double q0 = gsum(uu[0] * res), q1 = gsum(uu[1] * res), q2 = gsum(uu[2] * res);
        const double r1 = res - (uu[0] * (A00 * q0) + uu[1] * (A01 * q0 + A11 * q1) + uu[2] * (A02 * q0 + A12 * q1 + A22 * q2));
        {   // Z1 = T1^T (V1^T D): the two rows of a view sit in one group
            const double av[6] = {a0, a1, a2, a3, a4, a5};
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double w0 = uu[0] * av[a], w1 = uu[1] * av[a], w2 = uu[2] * av[a];
                w0 += dpp_move<0xB1>(w0);
                w1 += dpp_move<0xB1>(w1);
                w2 += dpp_move<0xB1>(w2);
                if (inr && (lane & 1) == 0 && wfirst) {
                    const int c = 6 * view + a;
                    sZ1[c] = A00 * w0;
                    sZ1[C6 + c] = A01 * w0 + A11 * w1;
                    sZ1[2 * C6 + c] = A02 * w0 + A12 * w1 + A22 * w2;
                }
            }
        }
        // level 2: the carry rows (the first three rows of every group: [R_g | ...]) reduced by column-pivoted reflectors, as
        // many as the one-level factorisation found the rank to be (the same |R_kk| up to rounding: the column norms of what
        // is left are invariant under Q_1)
        const bool isc = inr && lr < 3;
        double ww[3] = {0.0, 0.0, 0.0}, wb[3] = {0.0, 0.0, 0.0};
        {
            double c0 = isc ? hh[0] : 0.0, c1 = isc ? hh[1] : 0.0, c2 = isc ? hh[2] : 0.0;
            int L2[3] = {0, 0, 0};
            { int n = 0; for (int g = 0; g < ng && n < 3; ++g) { const int rows = min(3, 2 * (sr.gv[g + 1] - sr.gv[g])); for (int i = 0; i < rows && n < 3; ++i) L2[n++] = 2 * sr.gv[g] + i; } }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k >= rank) break;
                const bool act = isc && lane >= L2[k];
                double n0 = wave_sum(act ? c0 * c0 : 0.0);
                double n1 = (k < 2) ? wave_sum(act ? c1 * c1 : 0.0) : -1.0;
                double n2 = (k < 1) ? wave_sum(act ? c2 * c2 : 0.0) : -1.0;
                if (n1 > n0 && n1 >= n2) { double t_ = c0; c0 = c1; c1 = t_; n0 = n1; }
                else if (n2 > n0 && n2 > n1) { double t_ = c0; c0 = c2; c2 = t_; n0 = n2; }
                if (!(n0 > 0.0)) break;
                const double ry = fast_rsqrt(n0);
                const double nrm = n0 * ry;
                const double xk = readlane_d(c0, L2[k]);
                const double alpha = (xk > 0.0) ? -nrm : nrm;
                const double vk = act ? ((lane == L2[k]) ? (xk - alpha) : c0) : 0.0;
                const double beta = ry * fast_rcp(nrm + fabs(xk));
                if (k < 2) { const double d1 = wave_sum(vk * (act ? c1 : 0.0)); if (act) c1 -= beta * d1 * vk; }
                if (k < 1) { const double d2 = wave_sum(vk * (act ? c2 : 0.0)); if (act) c2 -= beta * d2 * vk; }
                ww[k] = vk; wb[k] = beta;
                c0 = c1; c1 = c2; c2 = 0.0;
            }
        }
        const double e01 = wave_sum(ww[0] * ww[1]), e02 = wave_sum(ww[0] * ww[2]), e12 = wave_sum(ww[1] * ww[2]);
        const double B00 = wb[0], B11 = wb[1], B22 = wb[2];
        const double B01 = -wb[1] * B00 * e01;
        const double B02 = -wb[2] * (B00 * e02 + B01 * e12);
        const double B12 = -wb[2] * B11 * e12;
        const double s0 = wave_sum(ww[0] * r1), s1 = wave_sum(ww[1] * r1), s2 = wave_sum(ww[2] * r1);
        const double r2 = r1 - (ww[0] * (B00 * s0) + ww[1] * (B01 * s0 + B11 * s1) + ww[2] * (B02 * s0 + B12 * s1 + B22 * s2));
        if (inr && wfirst) {
            sU[lane * 3 + 0] = uu[0]; sU[lane * 3 + 1] = uu[1]; sU[lane * 3 + 2] = uu[2];
            sW[lane * 3 + 0] = ww[0]; sW[lane * 3 + 1] = ww[1]; sW[lane * 3 + 2] = ww[2];
        }
        xsync();
        // Z2 = T2^T (V2^T H1), lanes over the columns: column c of view v meets the (up to three) carry rows of v's group only
        for (int c = lane + 64 * wv; c < C6; c += 64 * NWV) {
            const int v = c / 6, a = c - 6 * v, g = sGrp[v];
            const int L0 = 2 * sr.gv[g], rows = min(3, 2 * (sr.gv[g + 1] - sr.gv[g]));
            const double z0 = sZ1[c], z1 = sZ1[C6 + c], z2 = sZ1[2 * C6 + c];
            double t0 = 0.0, t1 = 0.0, t2 = 0.0;
            for (int i = 0; i < rows; ++i) {
                const int L = L0 + i;
                double hx = -(sU[L * 3 + 0] * z0 + sU[L * 3 + 1] * z1 + sU[L * 3 + 2] * z2);
                if ((L >> 1) == v) hx += sA[L * 6 + a];
                t0 += sW[L * 3 + 0] * hx; t1 += sW[L * 3 + 1] * hx; t2 += sW[L * 3 + 2] * hx;
            }
            sZ2[c] = B00 * t0;
            sZ2[C6 + c] = B01 * t0 + B11 * t1;
            sZ2[2 * C6 + c] = B02 * t0 + B12 * t1 + B22 * t2;
        }
        xsync();
        // K4: one group of columns at a time through the staging tile (lanes over rows), then out as contiguous row segments:
        // group g's narrow block (its rows 3.., its columns + r1) and the remainder block's columns of the group (the carry
        // rows behind the `rank` pivot rows, + r2 with the last group)
        n_wide_rows = ncarry - rank;
        const int ldw = C6 + 1;
        const long long wide_off = p.blk_off[sr.wide];
        for (int g = wv; g < ng; g += NWV) {            // (the groups dealt over the wavefronts; each has its own tile)
            const int gv0 = sr.gv[g], nvg = sr.gv[g + 1] - gv0;
            const int c0g = 6 * gv0, cw = 6 * nvg;
            const bool lastg = g == ng - 1;
            if (inr) {
                const double av[6] = {a0, a1, a2, a3, a4, a5};
                double* erow = sE + lane * ldE;
                const bool mine = grp == g;
                for (int vw = gv0; vw < gv0 + nvg; ++vw) {
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        const int c = 6 * vw + a;
                        double x = 0.0;
                        if (mine) {
                            x = -(uu[0] * sZ1[c] + uu[1] * sZ1[C6 + c] + uu[2] * sZ1[2 * C6 + c]);
                            if (vw == view) x += av[a];
                        }
                        if (isc) x -= ww[0] * sZ2[c] + ww[1] * sZ2[C6 + c] + ww[2] * sZ2[2 * C6 + c];
                        erow[c - c0g] = x;
                    }
                }
                erow[cw] = isc ? r2 : r1;
            }
            wave_sync();
            if (sr.child[g] >= 0) {                       // narrow block: rows 2 gv0 + 3 .. of the tile, cw + 1 columns, contiguous
                const int qg = 2 * nvg - 3, cwp = cw + 1, nel = qg * cwp;
                const long long boff = p.blk_off[sr.child[g]];
                const float inv_cwp = 1.0f / (float)cwp;
                for (int e = lane; e < nel; e += 64) {
                    const int L = (int)(((float)e + 0.5f) * inv_cwp), cc = e - L * cwp;
                    const double x = sE[(2 * gv0 + 3 + L) * ldE + cc];
                    if (p.stack_f32) static_cast<float*>(p.stack)[boff + e] = (float)x;
                    else static_cast<double*>(p.stack)[boff + e] = x;
                }
            }
            {                                             // remainder block: the carry rows behind the pivot rows
                const int cwp = cw + (lastg ? 1 : 0), nel = n_wide_rows * cwp;
                const float inv_cwp = 1.0f / (float)cwp;
                for (int e = lane; e < nel; e += 64) {
                    const int L = (int)(((float)e + 0.5f) * inv_cwp), cc = e - L * cwp;
                    const double x = sE[sCarry[rank + L] * ldE + cc];
                    const long long dst = wide_off + (long long)L * ldw + c0g + cc;
                    if (p.stack_f32) static_cast<float*>(p.stack)[dst] = (float)x;
                    else static_cast<double*>(p.stack)[dst] = x;
                }
            }
            wave_sync();
        }
    }
    // ---------------- K4 + K3, one column chunk (whole views, <= 64 columns) at a time -------------------
    // K4: the chunk's columns of H_o = D - V Z (and r_o with the last chunk) are staged in sE and leave as
    //     contiguous row segments of the stack block (row-major q x (6M+1), the q projected rows only).
    // K3: E = H_o P_sub for the chunk's columns (one lane per column; each P entry is read once per feature),
    //     then every lane (= row L of S) adds the chunk's part of S[L][:] = E[L,:] H_o^T to its REGISTER row:
    //     S never exists in LDS and sE holds one chunk only (17 KB per wavefront at 15 views instead of 36).
    const int ldb = C6 + 1;
    if (lane < R2 && wfirst) sRo[lane] = ro;
#pragma unroll
    for (int j = 0; j < RMAX; ++j) srow[j] = 0.0;
    double ez0 = 0.0, ez1 = 0.0, ez2 = 0.0;            // E[L,:] Z^T, over all chunks
    for (int vc0 = CV * wv; vc0 < M; vc0 += CV * NWV) { // (NWV > 1: the chunks dealt over the wavefronts, partial sums below)
        const int nv = min(CV, M - vc0);                // views of this chunk
        const int c0 = 6 * vc0, cw = 6 * nv;
        const bool last = vc0 + nv >= M;
        const int cwp = cw + (last ? 1 : 0);            // + the rhs column
        // ---- K4 ---- (the split form has written its blocks above)
        if constexpr (!SPLIT) {
        if (lane < R2) {
            const double av[6] = {a0, a1, a2, a3, a4, a5};
            double* erow = sE + lane * ldE;
            for (int vw = vc0; vw < vc0 + nv; ++vw) {
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    const int c = 6 * vw + a;
                    double x = -(vv0 * sZ[c] + vv1 * sZ[C6 + c] + vv2 * sZ[2 * C6 + c]);
                    if (vw == view) x += av[a];
                    erow[c - c0] = x;
                }
            }
            if (last) erow[cw] = ro;
        }
        wave_sync();
        {
            const int nel = q * cwp;
            const float inv_cwp = 1.0f / (float)cwp;
            for (int e = lane; e < nel; e += 64) {
                const int L = (int)(((float)e + 0.5f) * inv_cwp), cc = e - L * cwp;      // e / cwp, exact for e < 2^20
                const double x = sE[(rank + L) * ldE + cc];
                const long long dst = p.blk_off[f] + (long long)L * ldb + c0 + cc;
                if (p.stack_f32) static_cast<float*>(p.stack)[dst] = (float)x;
                else static_cast<double*>(p.stack)[dst] = x;
            }
        }
        wave_sync();                                 // sE is rewritten by the gate's first pass
        }
        // ---- K3 pass 1, lanes over the chunk's columns c of P_sub: E = D P_sub (block rows) - V (Z P_sub) ----
        if (lane < cw) {
            const int c = c0 + lane;
            const int colg = 15 + 6 * sSlot[c / 6] + (c % 6);
            double zp0 = 0, zp1 = 0, zp2 = 0;
            // the column of P_sub this lane walks is requested FOUR views ahead (an L2 read is ~0.6 us, a view's 30 FMAs ~0.1:
            // with one view of lookahead the pass was one load latency per view -- 30 of them per chunk of a 30-view track)
            double pb[4][6];
            auto fetchv = [&](double (&dst)[6], int vw) {
                const double* prow = p.P + (size_t)(15 + 6 * sSlot[vw]) * p.ldp + colg;
#pragma unroll
                for (int a = 0; a < 6; ++a) dst[a] = prow[(size_t)a * p.ldp];
            };
            auto usev = [&](const double (&pv)[6], int vw) {
                double e0 = 0, e1 = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    const int cc = 6 * vw + a;
                    zp0 += sZ[cc] * pv[a];
                    zp1 += sZ[C6 + cc] * pv[a];
                    zp2 += sZ[2 * C6 + cc] * pv[a];
                    e0 += sA[(2 * vw) * 6 + a] * pv[a];
                    e1 += sA[(2 * vw + 1) * 6 + a] * pv[a];
                }
                sE[(2 * vw) * ldE + lane] = e0;
                sE[(2 * vw + 1) * ldE + lane] = e1;
            };
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < M) fetchv(pb[i], i);
            for (int vw = 0; vw < M; vw += 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (vw + i < M) {
                        usev(pb[i], vw + i);
                        if (vw + i + 4 < M) fetchv(pb[i], vw + i + 4);
                    }
                }
            }
            // E -= V ZP  (same column, all rows)
            for (int L = 0; L < R2; ++L) {
                sE[L * ldE + lane] -= sV[L * 3 + 0] * zp0 + sV[L * 3 + 1] * zp1 + sV[L * 3 + 2] * zp2;
            }
        }
        wave_sync();
        // ---- K3 pass 2, lanes over rows L: this chunk's part of S[L][L2] = E[L,:] H_o[L2,:]^T ----
        if (lane < R2) {
            const double* erow = sE + lane * ldE;
            for (int cc = 0; cc < cw; ++cc) {
                const double e = erow[cc];
                ez0 += e * sZ[c0 + cc]; ez1 += e * sZ[C6 + c0 + cc]; ez2 += e * sZ[2 * C6 + c0 + cc];
            }
#pragma unroll
            for (int j = 0; j < RMAX - 1; ++j) {
                const int vw = j >> 1;
                if (j < R2 && vw >= vc0 && vw < vc0 + nv) {       // the D part of H_o row j lies in this chunk
                    double sacc = srow[j];
#pragma unroll
                    for (int a = 0; a < 6; ++a) sacc += erow[6 * (vw - vc0) + a] * sA[j * 6 + a];
                    srow[j] = sacc;
                }
            }
        }
        wave_sync();                                 // the next chunk restages sE
    }
    if constexpr (NWV > 1) {
        // the wavefronts' partial sums of srow / E Z^T, added up in wavefront 0's tile (free now) in turn; then wavefront 0 alone
        double* sAcc = sE0;
        const int lda = R2 + 3;
        __syncthreads();
#pragma unroll
        for (int w = 1; w < NWV; ++w) {
            if (wv == w && lane < R2 && CV * w < M) {        // (a wavefront without a chunk has nothing to add)
                double* arow = sAcc + lane * lda;
                const bool firstw = w == 1;
#pragma unroll
                for (int j = 0; j < RMAX; ++j)
                    if (j < R2) arow[j] = (firstw ? 0.0 : arow[j]) + srow[j];
                arow[R2] = (firstw ? 0.0 : arow[R2]) + ez0;
                arow[R2 + 1] = (firstw ? 0.0 : arow[R2 + 1]) + ez1;
                arow[R2 + 2] = (firstw ? 0.0 : arow[R2 + 2]) + ez2;
            }
            __syncthreads();
        }
        if (wv != 0) return;
        if (lane < R2 && CV < M) {
            const double* arow = sAcc + lane * lda;
#pragma unroll
            for (int j = 0; j < RMAX; ++j)
                if (j < R2) srow[j] += arow[j];
            ez0 += arow[R2]; ez1 += arow[R2 + 1]; ez2 += arow[R2 + 2];
        }
        wave_sync();
    }
    if (p.stamps) tq[3] = tq[4] = tq[5] = wall_clock64();
    // S row of this lane: the -V Z part, sigma^2 on the diagonal, the rhs column r_o; lane R2 holds the extra row r_o^T
#pragma unroll
    for (int j = 0; j < RMAX; ++j) {
        double x = 0.0;
        if (lane < R2) {
            if (j < R2) {
                x = srow[j] - (ez0 * sV[j * 3 + 0] + ez1 * sV[j * 3 + 1] + ez2 * sV[j * 3 + 2]);
                if (j == lane) x += p.sigma2;
            } else if (j == R2) {
                x = ro;
            }
        } else if (lane == R2) {
            if (j < R2) x = sRo[j];
        }
        srow[j] = x;
    }
    }
    // elimination over rows/cols rank..R2-1 with one matrix row per lane in REGISTERS (pivot row
    // entries travel by v_readlane, no LDS round trip per step); the extra row R2 ends with
    // -gamma in the corner
    int bad = 0;
    double gam = 0.0;
    if constexpr (RMAX <= 32) {
        // All 64 lanes: lane (g, l) = (lane >> 4, lane & 15) holds rows l (ea) and l + 16 (eb) of the columns j = 4 jj + g.
        // A pivot step is then: pivot by v_readlane, the multipliers (column k, which group k & 3 holds) by one
        // ds_bpermute pair per row slot, and the update with the pivot row folded in (v_fmac_f64_dpp row_newbcast:k & 15,
        // the pivot row of MY columns sits in MY 16-lane row) -- two DP instructions per (pivot, column quad) instead of
        // two v_readlane and an FMA per (pivot, column).  Columns left of the pivot are never read again, so they are
        // not masked out; rows at or above it are (their multiplier is zero).
        double* sT = sZ + 3 * C6;                       // the staging area of K4 / the gate is free now
        const int ldT = R2 + 3;
        wave_sync();
        if constexpr (CHUNKED) {                       // (the all-columns form wrote S there itself)
            if (lane <= R2) {
#pragma unroll
                for (int j = 0; j < RMAX; ++j)
                    if (j <= R2) sT[lane * ldT + j] = srow[j];
            }
        }
        wave_sync();
        const int g = lane >> 4, l15 = lane & 15;
        constexpr int NQ = (RMAX + 3) / 4;               // column quads that can hold a column j <= R2 (k_feature<24>: 6, <32>: 8)
        double ea[NQ], eb[NQ];
#pragma unroll
        for (int jj = 0; jj < NQ; ++jj) {
            const int j = 4 * jj + g;
            ea[jj] = (j <= R2 && l15 <= R2) ? sT[l15 * ldT + j] : 0.0;
            eb[jj] = (j <= R2 && l15 + 16 <= R2) ? sT[(l15 + 16) * ldT + j] : 0.0;
        }
        auto bperm_d = [&](double x, int src4) {
            const int lo = __builtin_amdgcn_ds_bpermute(src4, __double2loint(x));
            const int hi = __builtin_amdgcn_ds_bpermute(src4, __double2hiint(x));
            return __hiloint2double(hi, lo);
        };
        auto pivot_step = [&](auto tagk) {
            constexpr int k = decltype(tagk)::value;
            constexpr int gk = k & 3, jk = k >> 2, lk = k & 15;
            if (k >= rank && k < R2 && !bad) {
                const double piv = (k < 16) ? readlane_d(ea[jk], 16 * gk + lk) : readlane_d(eb[jk], 16 * gk + lk);
                if (!(piv > 0.0)) {
                    bad = 1;
                } else {
                    const double rp = fast_rcp(piv);
                    const int src4 = (16 * gk + l15) * 4;                              // the lane of group gk with my rows
                    const double mb = bperm_d(eb[jk], src4);
                    const double wB = (l15 + 16 > k) ? -(mb * rp) : 0.0;               // rows 16 .. 31
                    if constexpr (k < 16) {
                        const double ma = bperm_d(ea[jk], src4);
                        const double wA = (l15 > k) ? -(ma * rp) : 0.0;                // rows 0 .. 15; the pivot row is lane lk's ea
                        // column quads jk .. NQ - 1, up to four per statement
                        constexpr int NJ = NQ - jk, N1 = NJ < 4 ? NJ : 4, N2 = NJ - N1;
                        if constexpr (NJ > 0) {
                            double bq[N1], aq_[N1];
#pragma unroll
                            for (int i = 0; i < N1; ++i) { bq[i] = eb[jk + i]; aq_[i] = ea[jk + i]; }
                            elim_pairs<lk>(bq, aq_, wB, wA);
#pragma unroll
                            for (int i = 0; i < N1; ++i) { eb[jk + i] = bq[i]; ea[jk + i] = aq_[i]; }
                        }
                        if constexpr (N2 > 0) {
                            double bq[N2], aq_[N2];
#pragma unroll
                            for (int i = 0; i < N2; ++i) { bq[i] = eb[jk + 4 + i]; aq_[i] = ea[jk + 4 + i]; }
                            elim_pairs<lk>(bq, aq_, wB, wA);
#pragma unroll
                            for (int i = 0; i < N2; ++i) { eb[jk + 4 + i] = bq[i]; ea[jk + 4 + i] = aq_[i]; }
                        }
                    } else {                                                             // rows 0 .. 15 are done; the pivot row is lane lk's eb
                        constexpr int NJ = NQ - jk;                                      // (jk >= 4: at most four quads)
                        if constexpr (NJ > 0) {
                            double bq[NJ];
#pragma unroll
                            for (int i = 0; i < NJ; ++i) bq[i] = eb[jk + i];
                            elim_self<lk>(bq, wB);
#pragma unroll
                            for (int i = 0; i < NJ; ++i) eb[jk + i] = bq[i];
                        }
                    }
                }
            }
        };
        pivot_step(FTag<0>{}); pivot_step(FTag<1>{}); pivot_step(FTag<2>{}); pivot_step(FTag<3>{}); pivot_step(FTag<4>{});
        pivot_step(FTag<5>{}); pivot_step(FTag<6>{}); pivot_step(FTag<7>{}); pivot_step(FTag<8>{}); pivot_step(FTag<9>{});
        pivot_step(FTag<10>{}); pivot_step(FTag<11>{}); pivot_step(FTag<12>{}); pivot_step(FTag<13>{}); pivot_step(FTag<14>{});
        pivot_step(FTag<15>{}); pivot_step(FTag<16>{}); pivot_step(FTag<17>{}); pivot_step(FTag<18>{}); pivot_step(FTag<19>{});
        pivot_step(FTag<20>{}); pivot_step(FTag<21>{}); pivot_step(FTag<22>{});
        if constexpr (RMAX > 24) {
            pivot_step(FTag<23>{}); pivot_step(FTag<24>{}); pivot_step(FTag<25>{}); pivot_step(FTag<26>{}); pivot_step(FTag<27>{});
            pivot_step(FTag<28>{}); pivot_step(FTag<29>{}); pivot_step(FTag<30>{});
        }
#pragma unroll
        for (int j = 0; j < RMAX; ++j)
            if (j == R2) gam = -((j < 16) ? readlane_d(ea[j >> 2], 16 * (j & 3) + (j & 15)) : readlane_d(eb[j >> 2], 16 * (j & 3) + (j & 15)));
    } else {
        // RMAX = 64 (tracks of 16 - 31 views): the same all-lane form with FOUR row slots -- lane (g, l) holds rows l + 16 s,
        // s = 0 .. 3, of the columns j = 4 jj + g.  (Rounds 1-4 kept one matrix row per lane and moved every pivot-row entry
        // by v_readlane: 64 of the 190 us of a 30-view track.)
        double* sT = sZ + 3 * C6;
        const int ldT = R2 + 3;
        wave_sync();
        if (lane <= R2) {
#pragma unroll
            for (int j = 0; j < RMAX; ++j)
                if (j <= R2) sT[lane * ldT + j] = srow[j];
        }
        wave_sync();
        const int g = lane >> 4, l15 = lane & 15;
        double e[4][16];
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int j = 4 * jj + g, row = l15 + 16 * sl;
                e[sl][jj] = (j <= R2 && row <= R2) ? sT[row * ldT + j] : 0.0;
            }
        }
        auto bperm_d = [&](double x, int src4) {
            const int lo = __builtin_amdgcn_ds_bpermute(src4, __double2loint(x));
            const int hi = __builtin_amdgcn_ds_bpermute(src4, __double2hiint(x));
            return __hiloint2double(hi, lo);
        };
        auto pivot4 = [&](auto tagk) {
            constexpr int k = decltype(tagk)::value;
            constexpr int sk = k >> 4, lk = k & 15, gk = k & 3, jk = k >> 2;
            if (k >= rank && k < R2 && !bad) {
                const double piv = readlane_d(e[sk][jk], 16 * gk + lk);
                if (!(piv > 0.0)) {
                    bad = 1;
                } else {
                    const double rp = fast_rcp(piv);
                    const int src4 = (16 * gk + l15) * 4;                  // the lane of group gk with my rows
                    double w[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int sl = sk; sl < 4; ++sl) {
                        const double m = bperm_d(e[sl][jk], src4);
                        w[sl] = (l15 + 16 * sl > k) ? -(m * rp) : 0.0;
                    }
                    // the slots below the pivot's first (their source, the pivot row in slot sk, is still untouched), then slot sk itself
                    if constexpr (sk < 3) elim_from16<lk, jk, 16 - jk>(e[3], e[sk], w[3]);
                    if constexpr (sk < 2) elim_from16<lk, jk, 16 - jk>(e[2], e[sk], w[2]);
                    if constexpr (sk < 1) elim_from16<lk, jk, 16 - jk>(e[1], e[sk], w[1]);
                    elim_self16<lk, jk, 16 - jk>(e[sk], w[sk]);
                }
            }
        };
        feature_static_for<0, RMAX - 2>(pivot4);
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
#pragma unroll
            for (int jj = 0; jj < 16; ++jj)
                if (sl == (R2 >> 4) && jj == (R2 >> 2)) gam = -readlane_d(e[sl][jj], 16 * (R2 & 3) + (R2 & 15));
        }
    }
    if (p.stamps) tq[6] = wall_clock64();
    bool ok = (q >= 1) && (bad == 0) && (q < p.n_chi2);
    if (ok) ok = (gam <= p.chi2[q]);
    if (lane == 0) {
        p.rank[f] = rank;
        p.gamma[f] = gam;
        // 1 accepted, 0 rejected by the gate, 2 rejected because the gate matrix was not SPD.
        // (No global counters: 2000 workgroups adding to one word serialise at ~13 ns each and
        // were 50 us of this kernel; the host sums the per-feature results instead.)
        p.accepted[f] = ok ? 1 : (bad ? 2 : 0);
        if (p.acc_h) { p.rank_h[f] = rank; p.acc_h[f] = ok ? 1 : (bad ? 2 : 0); }
        if constexpr (SPLIT) {
            // the blocks inherit the track's gate result; their `rank` is what K5's row count 2 M - rank needs
            const SplitRec& sr = p.split[blockIdx.x];
            const unsigned char code = ok ? 1 : (bad ? 2 : 0);
            for (int g = 0; g < sr.ng; ++g) if (sr.child[g] >= 0) { p.rank[sr.child[g]] = 3; p.accepted[sr.child[g]] = code; }
            p.rank[sr.wide] = R2 - n_wide_rows; p.accepted[sr.wide] = code;
        }
        if (p.stamps) { tq[7] = wall_clock64(); for (int i = 0; i < 8; ++i) p.stamps[8 * f + i] = tq[i]; }
    }
}

// The remainder blocks of a batch with few enough long tracks (up to 16 GS_MAX_NB2 rows: a merge tree over them would take longer) as ONE dense
// row-major matrix [rows][6N + 1] = [H | r], zero where a track has no view, the rows of the accepted tracks only, zero rows up to the
// next multiple of 16: the sequential block update (k_gstream.h) takes them 16 at a time as its second source of rows, in front
// of the band root's -- inside the root sweep's launch, while the sweep is on its first columns.
//   reference MSCKF.py:581-588 (rows of the stack), :604-614 (the update they enter)
struct RemScatterArgs {
    const SplitRec* split;       // [n_tracks]
    int n_tracks;                // long tracks; workgroup n_tracks zeroes the padding rows and publishes the row count
    int rows_cap;                // sum of 3 ng: what `out` holds at most
    int dc, N;
    const int* view_ptr; const int* obs_slot;
    const long long* blk_off; const void* stack; int stack_f32;
    const int* rank; const unsigned char* accepted;
    double* out;                 // [<= rows_cap rounded up to 16][dc + 1]
    int* nrows;                  // [0] row blocks of 16 (K6-K7 reads it at the head of its launch), [1] rows
};
// Only the rows that exist are laid down, in track order: a track's first row is the number of rows of the accepted tracks in front
// of it (every workgroup sums them for itself: a few hundred records from L2 -- no scan kernel of its own).
__global__ __launch_bounds__(256) void k_rem_scatter(RemScatterArgs p) {
    __shared__ int s_view[256];                       // clone slot -> view of the track (-1: none)
    __shared__ int s_red[256];
    const int t = threadIdx.x, ld = p.dc + 1, me = (int)blockIdx.x;
    int part = 0;
    for (int i = t; i < min(me, p.n_tracks); i += 256) {
        const int f = p.split[i].wide;
        if (p.accepted[f] == 1) part += 2 * (p.view_ptr[f + 1] - p.view_ptr[f]) - p.rank[f];
    }
    s_red[t] = part;
    __syncthreads();
    for (int k = 128; k >= 1; k >>= 1) { if (t < k) s_red[t] += s_red[t + k]; __syncthreads(); }
    const int row0 = s_red[0];
    if (me == p.n_tracks) {                           // row0 = all rows: zero up to the next multiple of 16, publish
        const int pad = ((row0 + 15) & ~15) - row0;
        for (int e = t; e < pad * ld; e += 256) p.out[(size_t)row0 * ld + e] = 0.0;
        if (t == 0) { p.nrows[0] = (row0 + 15) >> 4; p.nrows[1] = row0; }
        return;
    }
    const SplitRec sr = p.split[me];
    const int f = sr.wide, v0 = p.view_ptr[f], M = p.view_ptr[f + 1] - v0;
    const int q = (p.accepted[f] == 1) ? 2 * M - p.rank[f] : 0;
    if (q == 0) return;
    for (int i = t; i < p.N; i += 256) s_view[i] = -1;
    __syncthreads();
    if (t < M) s_view[p.obs_slot[v0 + t]] = t;
    __syncthreads();
    const long long boff = p.blk_off[f];
    const int ldb = 6 * M + 1;
    for (int e = t; e < q * ld; e += 256) {
        const int row = e / ld, col = e - row * ld;
        double x = 0.0;
        int src = -1;
        if (col == p.dc) src = 6 * M;
        else { const int sl = col / 6, v = s_view[sl]; if (v >= 0) src = 6 * v + (col - 6 * sl); }
        if (src >= 0) {
            const long long at = boff + (long long)row * ldb + src;
            x = p.stack_f32 ? (double)static_cast<const float*>(p.stack)[at] : static_cast<const double*>(p.stack)[at];
        }
        p.out[(size_t)(row0 + row) * ld + col] = x;
    }
}

}  // namespace msckf
