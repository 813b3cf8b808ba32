// K6-K7 as a SEQUENTIAL BLOCK update that follows the root sweep of K5 row block by row block.
//   reference MSCKF.py:604-607 : S = T P T^T + R_n ; K = P T^T S^-1 ; dx = K r_n
//   reference MSCKF.py:612-614 : P+ = (I-KT) P (I-KT)^T + K R_n K^T ; P+ <- (P+ + P+^T)/2
// With R_n = sigma^2 I (MSCKF.py:598: Q^T (sigma^2 I) Q) the rows of the compressed system [T | r_n] are
// measurements with independent unit-variance-times-sigma^2 noise, so the batch update above equals processing
// them in blocks of 16 rows, each against the covariance the previous blocks left (the block Cholesky of S in
// disguise: with S = L L^T, X = P T^T L^-T one has K = X L^-1, P+ = P - X X^T, dx = X L^-1 r_n, and column block
// I of X is  X_I = (P^(I) T_I^T) L_II^-T  with  P^(I) = P - sum_{J<I} X_J X_J^T,  L_II L_II^T = T_I P^(I) T_I^T + sigma^2 I):
//
//     Y_I = P^(I)[:, clone columns of T_I] T_I^T          d x 16      (T_I: 16 rows of T, a band of `ncb` 16-column blocks)
//     A_I = T_I Y_I[clone rows] + sigma^2 I               16 x 16, SPD
//     X_I = Y_I L_II^-T,  L_II = chol(A_I)
//     P^(I+1) = P^(I) - X_I X_I^T
//
// Row block I of T is final as soon as the root sweep has passed it, i.e. this whole stage runs BESIDE the sweep, in
// the same launch (k_root_gain / k_root_gain_w at the end of this file: workgroup 0 sweeps and publishes a progress
// word, the other workgroups are the strips below), and only the last block trails it: K6-K7 used to be 102 us of
// launches behind K5 (two 180-deep products, a 180-column Cholesky, two triangular sweeps, the Joseph products),
// every one of them waiting for the complete T.  Checked against the reference's own outputs in
// tests (1e-15 on P+ in NumPy on all golden fixtures, recipe B included: each A_I is a 16 x 16 matrix of modest
// condition, where the reference inverts the whole S).
//
// dx needs no code of its own: the state is augmented by one row that starts as zero and is treated like a state
// row whose "Y" entry gets r_I added; after the last block it holds -dx (row 15 of strip 0, which has 15 real rows).
//
// Work split: workgroup r owns COLUMN strip r of P (strip 0 = the 15 IMU rows + the dx row, strip s >= 1 = clone
// columns 16 (s-1) ..): tiles P(s, r) live in matrix-core accumulator registers of wavefront s (lane (g, c) holds
// rows {g + 4 i} of column c) for the whole kernel; by symmetry the same registers ARE the A operand of P(r, s).
// Per row block: the wavefronts whose strip meets T_I's columns form partials of Y_I[r] (4 MFMAs each), one
// wavefront publishes the sum (2 KB, write-through stores + flag), every wavefront s fetches Y_I[s] from strip s's
// workgroup -- ONE all-to-all exchange per row block --, A_I and its elimination are computed redundantly in every
// workgroup (one wavefront, the in-wave elimination of k_chol16; the others follow its multipliers through LDS
// exactly like k_chol16's panel owners) and the rank-16 update of the strip is 4 MFMAs per tile.
// Inter-workgroup visibility: every exchanged byte is stored sc1 (write-through), drained (s_waitcnt vmcnt(0))
// before its flag is stored sc1 by one lane, and loaded sc1 by the wave that polled the flag
// (MI355X_MICROARCH.md, inter-workgroup visibility: valid forms, first table row).  Flags carry the launch's epoch:
// nothing has to be cleared between launches.  Every poll is bounded (status 2 = timeout).
#pragma once
#include <hip/hip_runtime.h>
#include "wave_ops.h"
#include "k_gain.h"
#include "k_sweep.h"
#include "k_wsweep.h"

namespace msckf {

typedef __attribute__((address_space(1))) unsigned long long gs_gu64;
__device__ __forceinline__ unsigned long long gs_ld(const void* p) {
    return __hip_atomic_load((const gs_gu64*)(const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double gs_ldd(const double* p) { return __longlong_as_double((long long)gs_ld(p)); }
__device__ __forceinline__ void gs_st(void* p, unsigned long long v) {
    __hip_atomic_store((gs_gu64*)(unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gs_std(double* p, double v) { gs_st(p, (unsigned long long)__double_as_longlong(v)); }

constexpr int GS_WAVES = 16;                 // wavefronts per workgroup: tile wavefronts 0 .., wavefront 15 also eliminates
constexpr int GS_PUB_WAVE = 14;              // publishes the strip's Y tile
constexpr int GS_MAX_NS = 32;                // strips (two tiles per wavefront)
constexpr int GS_MAX_NB2 = 1024;             // row blocks of the second source (dense remainder rows of split long tracks)
constexpr int GS_XS = 17;                    // X tiles in LDS: [column][row] with columns 17 doubles apart (a stride of 16 put a
constexpr int GS_XT = 16 * GS_XS;            //  wavefront's store on four bank pairs: 1.2 us per row block for thirteen tiles)
constexpr long long GS_TIMEOUT_TICKS = 50000000LL;   // 0.5 s of the 100 MHz wall clock

struct GStreamArgs {
    const double* P; int ldp;                // prior covariance, d x d
    const double* dx0;                       // null, or the dx a previous update on OTHER rows of the same batch left: the dx row starts
                                             // from it (P is then that update's P+; split long tracks, launch_gain_chain)
    const double* T; int ldt;                // root block [T | r_n]: dc rows x (dc + 1), row-major, zero below the diagonal
    const unsigned long long* progress;      // (epoch << 32) | rows of T that are final; null: all rows are (standalone)
    unsigned epoch;
    double* ex;                              // exchange tiles [nb][ns][256]
    unsigned long long* exflag;              // [row blocks of both sources][ns]: epoch << 32 | row block + 1
    double* dx; double* Pout; int ldo;
    double* dx_h; double* Pout_h; int* status_h;     // optional mirrors in pinned host memory (same layout; the one-shot call)
    int* status;                             // [0]: 0 ok, 1 a pivot was not a positive normal number, 2 timeout
    double sigma2;
    int d, dc, nb, ns, ncb;                  // nb row blocks of T, ns = nb + 1 strips, ncb column blocks per row block
    // A second, COMPLETE source of rows, taken FIRST (while the sweep that publishes T is still on its first columns): the
    // remainder blocks of split long tracks as a dense row-major matrix [16 nb2][ldt2] (k_rem_scatter; complete when the
    // launch starts -- its producer is ordered in front of the launch by the stream or an event).  nb2 = 0: none.
    const double* T2; int ldt2; int nb2;
    const int* nb2_dev;                      // null, or where the producer of T2 left the number of its row blocks (<= nb2, the capacity)
    int nb1;                                 // row blocks of T to take (nb, or 0 when there is no band root at all)
    int f32_update;                          // msckf_config.dtype = f32: the rank-16 products X_I[s] X_I[r]^T of the P-update on the
                                             // f32 matrix cores (fp32 operands and sums of 16 terms; P itself stays fp64)
    long long* stamps;                       // -DGS_STAMPS builds: wall-clock stamps of workgroup 0, 8 per row block
    long long* tstamp;                       // optional: [2] wall clock (10 ns ticks) when strip 0 has stored its results
};
#ifdef GS_STAMPS
#define GS_STAMP(w, i) do { if (p.stamps && r == 0 && wv == (w) && lane == 0) p.stamps[I * 8 + (i)] = wall_clock64(); } while (0)
#else
#define GS_STAMP(w, i) do { } while (0)
#endif

// Row blocks of T: when 6 N is no multiple of 16 the SHORT block is the first one (rows [0, rem)), every other block has
// 16 rows -- so the last block, the one that trails the sweep, is not preceded by a block that became final moments
// before it (with the short block last, two blocks' worth of work trailed the sweep).
// 16-column strips a row block meets when its rows are `band` columns wide (any alignment of the block against the strips)
__host__ __device__ inline int gstream_ncb(int dc, int band) {
    const int nb = (dc + 15) / 16, rem = dc - 16 * (nb - 1);
    const int bw = band < dc ? band : dc;
    const int n = ((rem & 15) + 15 + bw - 1) / 16 + 1;
    return n < nb ? (n < 1 ? 1 : n) : nb;
}
__host__ __device__ inline size_t gstream_lds_doubles(int ns, int ncb) {
    // multipliers [2][16][17] | 1 / l_cc [2][16] | control words | Y partials [ncb][256] | A partials [ncb][256] | X [ns][GS_XT]
    return 544 + 32 + 16 + (size_t)2 * ncb * 256 + (size_t)ns * GS_XT;
}

// In-wave elimination of a 16 x 16 SPD block in the accumulator layout (k_chol16's D phase, see k_gain.h): per pivot
// the row of multipliers m_c = -a_cp / a_pp and the progress word go to sw_half ([16][17] doubles), 1 / l_cc to sri.
// Returns true when a pivot is not a positive normal number.
// NPIV: pivots to run (the last row block of T may hold fewer than 16 rows: the rest of its A_I is sigma^2 on the diagonal).
template <int NPIV>
__device__ __forceinline__ bool gs_eliminate(double (&d)[4], unsigned sw_half, double* sri, int lane) {
    const int g = lane >> 4, cc = lane & 15, cc4 = cc * 4;
    const unsigned w_out = sw_half + lane * 8;
    int rlo, rhi;
    double piv;
    asm volatile("s_nop 1\n\tds_bpermute_b32 %0, %2, %3\n\tds_bpermute_b32 %1, %2, %4"
                 : "=&v"(rlo), "=&v"(rhi) : "v"(cc4), "v"(__double2loint(d[0])), "v"(__double2hiint(d[0])) : "memory");
    piv = readlane_d(d[0], 0);
    auto pivot = [&](auto tagp) {
        constexpr int P = decltype(tagp)::value;
        constexpr int GN = (P + 1) & 3, IN = ((P + 1) >> 2) & 3;     // where row / pivot P + 1 live
        int ccl = cc;
        asm volatile("" : "+v"(ccl));
        double r = __builtin_amdgcn_rcp(piv);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rlo), "+v"(rhi), "+v"(r)::"memory");
        const double lc = __hiloint2double(rhi, rlo);
        const double lcm = (ccl > P) ? -lc : 0.0;
        const double e = fma(-piv, r, 1.0);
        const double w = lcm * r;
        const double w2 = fma(w, e, w);
        if constexpr (P < NPIV - 1) {
            asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(d[IN]) : "v"(w2), "i"(P));
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
#ifndef GS_ELIM_PRIO
#define GS_ELIM_PRIO 3                       // (the elimination is the block's dependent chain: 1.6 us at priority 3, 2.0 at 0)
#endif
    if (GS_ELIM_PRIO) __builtin_amdgcn_s_setprio(GS_ELIM_PRIO);
    pivot(CTag<0>{}); pivot(CTag<1>{}); pivot(CTag<2>{}); pivot(CTag<3>{});
    if constexpr (NPIV > 4) { pivot(CTag<4>{}); pivot(CTag<5>{}); pivot(CTag<6>{}); pivot(CTag<7>{}); }
    if constexpr (NPIV > 8) { pivot(CTag<8>{}); pivot(CTag<9>{}); pivot(CTag<10>{}); pivot(CTag<11>{}); }
    if constexpr (NPIV > 12) { pivot(CTag<12>{}); pivot(CTag<13>{}); pivot(CTag<14>{}); pivot(CTag<15>{}); }
    // the pivots are the diagonal as it stands now: lane (g, c) picks register c >> 2 and takes it from row group c & 3
    double pivs;
    {
        // (scalars behind an empty asm statement: the compiler turned the selects over d[] into an indexed load of a copy of
        //  the array in scratch memory -- a store / load round trip at the end of every elimination)
        double d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];
        asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        const double sel = (cc < 8) ? ((cc < 4) ? d0 : d1) : ((cc < 12) ? d2 : d3);
        const int src4 = (16 * (cc & 3) + cc) * 4;
        const int plo = __builtin_amdgcn_ds_bpermute(src4, __double2loint(sel)), phi = __builtin_amdgcn_ds_bpermute(src4, __double2hiint(sel));
        pivs = __hiloint2double(phi, plo);
    }
    const bool bad = __ballot(!(pivs > 1e-200 && pivs < 1e200)) != 0ull;
    const double ri = bad ? 1.0 : fast_rsqrt(pivs);
    asm volatile("" ::: "memory");
    if (g == 0) sri[cc] = ri;
    if (GS_ELIM_PRIO) __builtin_amdgcn_s_setprio(0);
    return bad;
}

// A tile (any 16 rows x the block's 16 columns, accumulator layout) follows the pivots of gs_eliminate:
// a <- a L^-T (k_chol16's panel owners), four pivots per poll of the progress words.
// NT tiles of one wavefront take the same multipliers side by side (independent chains).
template <int NPIV, int NT>
__device__ __forceinline__ void gs_follow(double (&a0)[NT][4], unsigned w_in, const double* sri, int lane) {
    constexpr int LASTM = NPIV - 2;                                  // the last pivot with multipliers
    const int cc = lane & 15;
    const unsigned w_cc = w_in + cc * 8;
    auto follow4 = [&](auto tagp) {
        constexpr int P0 = decltype(tagp)::value;
        constexpr int PL = (P0 + 3 < LASTM) ? P0 + 3 : LASTM;
        double (*ap)[4] = a0;
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
        if constexpr (P0 + 3 <= LASTM) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(m3) : "v"(wcc_l), "i"((P0 + 3) * 17 * 8) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3)::"memory");
#define GS_FOLLOW1(PP, M)                                                                                                \
    _Pragma("unroll") for (int tq = 0; tq < NT; ++tq)                                                                   \
    asm volatile("s_nop 1\n\t"                                                                                          \
                 "v_fmac_f64_dpp %0, %0, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"                             \
                 "v_fmac_f64_dpp %1, %1, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"                             \
                 "v_fmac_f64_dpp %2, %2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"                             \
                 "v_fmac_f64_dpp %3, %3, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"                                 \
                 : "+v"(ap[tq][0]), "+v"(ap[tq][1]), "+v"(ap[tq][2]), "+v"(ap[tq][3]) : "v"(M), "i"(PP))
        GS_FOLLOW1(P0, m0);
        GS_FOLLOW1(P0 + 1, m1);
        GS_FOLLOW1(P0 + 2, m2);
        if constexpr (P0 + 3 <= LASTM) GS_FOLLOW1(P0 + 3, m3);
#undef GS_FOLLOW1
    };
    follow4(CTag<0>{});
    if constexpr (NPIV > 4) follow4(CTag<4>{});
    if constexpr (NPIV > 8) follow4(CTag<8>{});
    if constexpr (NPIV > 12) follow4(CTag<12>{});
    double ri = 0.0;
    {
        const unsigned ri_addr = lds_addr(sri + cc);
        int spins = 0;
        do {
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(ri) : "v"(ri_addr) : "memory");
            if (ri != 0.0) break;
            __builtin_amdgcn_s_sleep(1);
        } while (++spins < (1 << 22));
    }
#pragma unroll
    for (int tq = 0; tq < NT; ++tq)
#pragma unroll
        for (int i = 0; i < 4; ++i) a0[tq][i] *= ri;
}

// acc += the n partial tiles at src (256 doubles apart; this lane's four entries 64 apart), in order, four tiles' reads in flight
// at a time (a loop of one tile per trip, each waiting for its own four reads, took 0.9 us for twelve tiles -- twice per row block,
// on the block's dependent chain)
__device__ __forceinline__ void gs_sum_tiles(double (&acc)[4], const double* src, int n) {
    for (int s0 = 0; s0 < n; s0 += 4) {
        double v[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double* q = src + (size_t)min(s0 + c, n - 1) * 256;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[c][i] = q[64 * i];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (s0 + c < n) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] += v[c][i];
            }
    }
}

// WV wavefronts; wavefront wv < TW holds the tiles s = wv + TW q, q < TPW; wavefront WV - 1 eliminates (with TW < WV it has
// no tile of its own to follow afterwards), wavefront WV - 2 also publishes.
template <int WV, int TW, int TPW>
__device__ __forceinline__ void gain_stream_body(const GStreamArgs& p, const int r) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = lane >> 4, cc = lane & 15;
    const int d = p.d, dc = p.dc, nb = p.nb, ns = p.ns, ncb = p.ncb;
    double* sW = smem;                                   // [2][16][17]
    double* sRi = smem + 544;                            // [2][16]
    // ([0] bad pivot, [1] timeout; an LDS pointer by type: a volatile generic one compiles to flat accesses that wait for vmcnt)
    typedef __attribute__((address_space(3))) int gs_lds_int;
    volatile gs_lds_int* sCtl = (volatile gs_lds_int*)(gs_lds_int*)(int*)(smem + 576);
    // (a block of the dense second source meets all nb strips, one of T only ncb: the partial tiles are laid out for the larger
    //  count -- round 4 laid them out for ncb alone, and the second source's partials past ncb ran into sPartA / sX while other
    //  wavefronts were using those: the "rare nondeterminism" of batches with wide tracks)
    const int ncbl = (p.nb2 > 0 && nb > ncb) ? nb : ncb;
    double* sPartY = smem + 592;                         // [ncbl][256]
    double* sPartA = sPartY + (size_t)ncbl * 256;        // [ncbl][256]
    double* sX = sPartA + (size_t)ncbl * 256;            // [ns][GS_XT], operand order [column][row], column stride GS_XS
    const unsigned sw_addr = lds_addr(sW);
    auto g0 = [&](int s) { return s == 0 ? 0 : 15 + 16 * (s - 1); };
    auto nrows = [&](int s) { return s == 0 ? 15 : min(16, d - (15 + 16 * (s - 1))); };
    const int gr = g0(r), nr = nrows(r);
    const int rem0 = dc - 16 * (nb - 1);                 // rows of the first block of T (1 .. 16)
    // flag values: epoch << 32 | row block + 1; never 0 (the flags are zeroed once), never the value a previous launch left
    // in the same place (the epoch differs)

    // tiles P(s, r), s = wv + 16 q, symmetrised as they are loaded (the reference symmetrises its result, MSCKF.py:614)
    double Pt[TPW][4];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int s = (wv < TW) ? wv + TW * q : GS_MAX_NS + 1;
        const int gsr = g0(min(s, ns - 1)), ms = (s < ns) ? nrows(s) : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = g + 4 * i;
            const bool ok = (m < ms) && (cc < nr);
            const size_t row = ok ? gsr + m : 0, col = ok ? gr + cc : 0;
            const double x1 = p.P[row * p.ldp + col], x2 = p.P[col * p.ldp + row];
            Pt[q][i] = ok ? 0.5 * (x1 + x2) : 0.0;
            if (p.dx0) {
                // the dx row (row 15 of strip 0) holds -dx; its mirror image, column 15 of the tiles P(s, 0), is what strip 0's
                // workgroup forms the row's share of Y from
                if (s == 0 && m == 15 && cc < nr) Pt[q][i] = -p.dx0[gr + cc];
                if (r == 0 && cc == 15 && s < ns && m < ms) Pt[q][i] = -p.dx0[gsr + m];
            }
        }
    }
    if (t < 32) { sRi[t] = 0.0; sW[(t >> 4) * 272 + (t & 15) * 17 + 16] = __longlong_as_double(CHOL16_UNSET); }
    if (t < 4) sCtl[t] = 0;
    __syncthreads();

    const long long t_start = wall_clock64();
    bool failed = false;
    const int nb2 = p.nb2_dev ? min(p.nb2, *p.nb2_dev) : p.nb2;      // (written by a kernel in front of this launch: uniform)
    const int nb1 = p.nb1;
    for (int J = 0; J < nb2 + nb1; ++J) {
        const bool src2 = J < nb2;
        const int I = src2 ? J : J - nb2;
        const double* Tsrc = src2 ? p.T2 : p.T;
        const int ldts = src2 ? p.ldt2 : p.ldt;
        const int ncbs = src2 ? nb : ncb;
        // rows [row0, row0 + nrw) of the source: T's first block is the short one; the second source is a DENSE row matrix in
        // blocks of 16 (the remainder blocks of split long tracks, k_rem_scatter: its rows meet every strip)
        const int row0 = src2 ? 16 * I : ((I == 0) ? 0 : rem0 + 16 * (I - 1)), nrw = src2 ? 16 : ((I == 0) ? rem0 : 16);
        const int need = row0 + nrw;
        const unsigned long long tag = ((unsigned long long)p.epoch << 32) | (unsigned)(J + 1);
        // ---- A: row block I of T is final -------------------------------------------------------------------
        if (!src2 && p.progress && wv == WV - 1) {
            for (;;) {
                const unsigned long long v = gs_ld(p.progress);
                if ((v >> 32) == p.epoch && (int)(v & 0xffffffffu) >= need) break;
                if (wall_clock64() - t_start > GS_TIMEOUT_TICKS) { if (lane == 0) sCtl[1] = 1; break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        GS_STAMP(WV - 1, 0);                                      // rows of T seen
        if (p.tstamp && !src2 && r == 0 && wv == WV - 1 && lane == 0 && I < 15) p.tstamp[3 + I] = wall_clock64();
        // (the barrier hands the wait's outcome to the other wavefronts; a block that waits for nothing needs none: what it writes
        //  in LDS -- Y partials, then A partials, X -- was last read two barriers ago)
        if (!src2 && p.progress) {
            __syncthreads();
            if (sCtl[1]) { failed = true; break; }
        }
        const int s_lo = src2 ? 1 : 1 + (row0 >> 4), s_hi = min(s_lo + ncbs - 1, ns - 1);   // strips that hold T_I's columns
        // ---- B: partials of Y_I[r] = sum_s P(r, s) T_{I,s}^T ------------------------------------------------
        double Tt[TPW][4];
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int s = (wv < TW) ? wv + TW * q : GS_MAX_NS + 1;
            if (s >= s_lo && s <= s_hi) {
                const int trow = row0 + cc;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int col = 16 * (s - 1) + 4 * u + g;
                    const bool ok = cc < nrw && col < dc;
                    const double x = gs_ldd(Tsrc + (ok ? (size_t)trow * ldts + col : 0));
                    Tt[q][u] = ok ? x : 0.0;
                }
                v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pt[q][u], Tt[q][u], acc, 0, 0, 0);
                double* dst = sPartY + (size_t)(s - s_lo) * 256 + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) dst[64 * i] = acc[i];
            }
        }
        double rhs = 0.0;
        if (wv == WV - 2 && r == 0 && g == 3) {                   // r_n of this block: the dx row's share of Y
            const int trow = row0 + cc;
            const double x = gs_ldd(Tsrc + (cc < nrw ? (size_t)trow * ldts + dc : 0));
            rhs = cc < nrw ? x : 0.0;
        }
        GS_STAMP(WV - 1, 1);                                      // partials of Y written
        __syncthreads();
        // ---- C: publish Y_I[r] ---------------------------------------------------------------------------------
        if (wv == WV - 2) {
            double y[4] = {0.0, 0.0, 0.0, 0.0};
            gs_sum_tiles(y, sPartY + lane, s_hi - s_lo + 1);
            y[3] += rhs;                                               // (row 15 of strip 0; zero elsewhere)
            // payload write-through, drained, then the flag (MI355X_MICROARCH.md, inter-workgroup visibility, first table row).
            // Tried and dropped: tagged 8-byte granules polled by their readers instead of a flag (no drain, no flag store:
            // one round trip less on paper).  Readers that re-poll a line back to back keep being served the copy of their
            // first miss (every launch but the first timed out; a sleep between polls or a buffer_inv sc1 cured it), and
            // with the sleep the 4 KB sweeps of 13 x 13 wavefronts were slower than this: 108 against 90 us at N = 30.
            // Round 5 (tools/ubench/gstream_test.hip, 40 dense blocks at N = 30: 7.6 us per block): tagged granules behind an
            // UNDRAINED flag as a hint (one sweep per reader, sleep + second sweep on a stale tag) -- correct, but the drain is
            // only ~0.4 us of a block and the 4 KB reads cost more than that: 8.0 us; the T tiles of complete blocks requested one
            // block ahead: 8.05 us, requested behind the previous block's last barrier: 7.6 against 7.1 us.
            double* dst = p.ex + ((size_t)J * ns + r) * 256 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) gs_std(dst + 64 * i, y[i]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) gs_st(p.exflag + (size_t)J * ns + r, tag);
            GS_STAMP(WV - 2, 2);                                   // published
        }
        // ---- D: fetch Y_I[s]; partials of A_I = sum_s T_{I,s} Y_I[s] -------------------------------------------
        double Yt[TPW][4];
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int s = (wv < TW) ? wv + TW * q : GS_MAX_NS + 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) Yt[q][i] = 0.0;
            if (s < ns) {
                const unsigned long long* fl = p.exflag + (size_t)J * ns + s;
                bool ok = true;
                for (;;) {
                    if (gs_ld(fl) == tag) break;
                    if (wall_clock64() - t_start > GS_TIMEOUT_TICKS) { if (lane == 0) sCtl[1] = 1; ok = false; break; }
                    __builtin_amdgcn_s_sleep(2);                            // (never poll a line back to back: see above)
                }
                if (ok) {
                    const double* src = p.ex + ((size_t)J * ns + s) * 256 + lane;
#pragma unroll
                    for (int i = 0; i < 4; ++i) Yt[q][i] = gs_ldd(src + 64 * i);
                }
                if (s >= s_lo && s <= s_hi) {
                    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Tt[q][u], Yt[q][u], acc, 0, 0, 0);
                    double* dst = sPartA + (size_t)(s - s_lo) * 256 + lane;
#pragma unroll
                    for (int i = 0; i < 4; ++i) dst[64 * i] = acc[i];
                }
            }
        }
        GS_STAMP(WV - 1, 3);                                      // all Y tiles of this wavefront fetched
        __syncthreads();
        GS_STAMP(WV - 1, 4);
        // ---- E: eliminate A_I (one wavefront), every tile follows: X_I[s] = Y_I[s] L_II^-T ----------------------
        const int npiv = (nrw + 3) & ~3;                                   // pivots of this block (16 but for a short first one)
        const int half = J & 1;
        const unsigned sw_half = sw_addr + half * (16 * 17 * 8);
        double* sri = sRi + half * 16;
        if (wv == WV - 1) {
            double a[4] = {0.0, 0.0, 0.0, 0.0};
            gs_sum_tiles(a, sPartA + lane, s_hi - s_lo + 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) if (g + 4 * i == cc) a[i] += p.sigma2;
            // the other half's words are reset for the next block (its followers have passed this block's first barrier)
            if (lane < 16) { sW[(half ^ 1) * 272 + lane * 17 + 16] = __longlong_as_double(CHOL16_UNSET); sRi[(half ^ 1) * 16 + lane] = 0.0; }
            bool bad;
            if (npiv == 16) bad = gs_eliminate<16>(a, sw_half, sri, lane);
            else if (npiv == 12) bad = gs_eliminate<12>(a, sw_half, sri, lane);
            else if (npiv == 8) bad = gs_eliminate<8>(a, sw_half, sri, lane);
            else bad = gs_eliminate<4>(a, sw_half, sri, lane);
            if (bad && lane == 0) sCtl[0] = 1;
            GS_STAMP(WV - 1, 5);                                  // eliminated
        }
        if (wv < TW && wv < ns) {                                           // (tiles past the last strip are zeros: they ride along)
            if (npiv == 16) gs_follow<16, TPW>(Yt, sw_half, sri, lane);
            else if (npiv == 12) gs_follow<12, TPW>(Yt, sw_half, sri, lane);
            else if (npiv == 8) gs_follow<8, TPW>(Yt, sw_half, sri, lane);
            else gs_follow<4, TPW>(Yt, sw_half, sri, lane);
            GS_STAMP(0, 7);                                             // wavefront 0 has followed
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                const int s = (wv < TW) ? wv + TW * q : GS_MAX_NS + 1;
                if (s < ns) {
                    double* dst = sX + (size_t)s * GS_XT + cc * GS_XS + g;
#pragma unroll
                    for (int i = 0; i < 4; ++i) dst[4 * i] = Yt[q][i];  // [column][row]
                }
            }
        }
#ifdef GS_STAMPS
        if (p.stamps && r == 0 && I == 5 && lane == 0) p.stamps[64 * 8 + wv] = wall_clock64();
#endif
        __syncthreads();
        GS_STAMP(WV - 1, 6);                                      // everybody has followed
        if (sCtl[0] | sCtl[1]) { failed = true; break; }
        // ---- F: P(s, r) -= X_I[s] X_I[r]^T ----------------------------------------------------------------------
        {
            const double* xr = sX + (size_t)r * GS_XT + g * GS_XS + cc;
            double bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) bv[u] = xr[4 * GS_XS * u];
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                const int s = (wv < TW) ? wv + TW * q : GS_MAX_NS + 1;
                if (s < ns && p.f32_update) {
                    // v_mfma_f32_16x16x4_f32 leaves row 4 (lane >> 4) + reg of the product in a lane, the fp64 tile holds row
                    // (lane >> 4) + 4 reg: the A operand takes the rows of X_I[s] in the order pi(m) = (m >> 2) + 4 (m & 3), which
                    // puts product row (lane >> 4) + 4 reg exactly where the fp64 accumulator keeps it
                    const double* xs = sX + (size_t)s * GS_XT + g * GS_XS + ((cc >> 2) + 4 * (cc & 3));
                    v4f accf = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        accf = __builtin_amdgcn_mfma_f32_16x16x4f32((float)(-xs[4 * GS_XS * u]), (float)bv[u], accf, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) Pt[q][i] += (double)accf[i];
                } else if (s < ns) {
                    const double* xs = sX + (size_t)s * GS_XT + g * GS_XS + cc;
                    double av[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) av[u] = -xs[4 * GS_XS * u];
                    v4d acc = {Pt[q][0], Pt[q][1], Pt[q][2], Pt[q][3]};
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) Pt[q][i] = acc[i];
                }
            }
        }
    }
    if (failed) {
        if (r == 0 && t == 0) { p.status[0] = sCtl[1] ? 2 : 1; if (p.status_h) p.status_h[0] = sCtl[1] ? 2 : 1; }
        return;
    }
    if (r == 0 && t == 0) { p.status[0] = 0; if (p.status_h) p.status_h[0] = 0; }
    // ---- P+ tiles and dx (= minus the augmented row) -----------------------------------------------------------
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int s = (wv < TW) ? wv + TW * q : GS_MAX_NS + 1;
        if (s < ns) {
            const int gsr = g0(s), ms = nrows(s);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = g + 4 * i;
                if (m < ms && cc < nr) {
                    p.Pout[(size_t)(gsr + m) * p.ldo + gr + cc] = Pt[q][i];
                    if (p.Pout_h) p.Pout_h[(size_t)(gsr + m) * p.ldo + gr + cc] = Pt[q][i];
                }
            }
            if (s == 0 && g == 3 && cc < nr) { p.dx[gr + cc] = -Pt[q][3]; if (p.dx_h) p.dx_h[gr + cc] = -Pt[q][3]; }
        }
    }
    if (p.tstamp && r == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) p.tstamp[2] = wall_clock64();
    }
}

// alone (T complete, or a sweep kernel of another launch publishing it): one workgroup of 16 wavefronts per strip
template <int TPW>
__global__ __launch_bounds__(64 * GS_WAVES) void k_gain_stream(GStreamArgs p) {
    gain_stream_body<GS_WAVES, GS_WAVES, TPW>(p, blockIdx.x);
}

// ONE launch for the root sweep and the update that follows it: workgroup 0 is the sweep (NF fold wavefronts + the
// flusher), workgroups 1 .. ns the strips of the update (NF + 1 wavefronts each, TPW tiles per wavefront).  The strips
// wait for rows workgroup 0 publishes: workgroup 0 is dispatched first, and every workgroup has the chip to itself (a
// grid of at most 33 workgroups that ask for more than half of a CU's LDS), so all of them are resident at once.
// (As two launches on two streams the pair cost an event record in front of the sweep and a cross-stream wait behind
//  the update: ~17 us per update around ~225 us of kernels.)
// Workgroups behind the strips (mp.nodes != nullptr): the LAST level of group merges, in the same launch.  Each publishes the
// rows of its triangle as they become final (blocks of 8, its own progress word) and the root's folds take them from there
// (SweepFold::prod) instead of waiting for the level to end: the root runs ~27 macro steps behind the merges, where a launch
// boundary put it 68 steps + a launch gap behind (k_sweep.h).
template <int NF, int TPW>
__global__ __launch_bounds__(64 * (NF + 1)) void k_root_gain(SweepArgs sp, GStreamArgs gp, SweepArgs mp) {
    const int b = (int)blockIdx.x;
    if (b >= 1 && b <= gp.ns) gain_stream_body<NF + 1, NF, TPW>(gp, b - 1);
    else sweep_body<NF, 1, false, true>(b == 0 ? sp : mp, b == 0 ? 0 : b - 1 - gp.ns);
}

// ... with merge nodes of NFM > NFR fold slots (groups of 9 - 12 leaf triangles: one round on eleven slots): every workgroup
// has NFM + 1 wavefronts -- twelve, three per SIMD, which is what the sweep's 146 registers allow --, the root sweeps with its
// NFR slots and lets the spare ones go, the strips spread their tiles over NFM wavefronts.
template <int NFR, int NFM, int TPW>
__global__ __launch_bounds__(64 * (NFM + 1)) void k_root_gain_m(SweepArgs sp, GStreamArgs gp, SweepArgs mp) {
    const int b = (int)blockIdx.x;
    if (b == 0) sweep_body<NFR, 1, false, true, NFM - NFR>(sp, 0);
    else if (b <= gp.ns) gain_stream_body<NFM + 1, NFM, TPW>(gp, b - 1);
    else sweep_body<NFM, 1, false, true>(mp, b - 1 - gp.ns);
}

// The same pairing for the ring-buffered sweeps (N > 37 clones or tracks of 11 - 15 slots): k_wsweep's fold wavefronts store
// the final rows themselves (write-through) and wavefront 0 publishes the count; NF wavefronts per workgroup, the strips'
// tiles on NF - 1 of them.
template <int NF, int CS, int TPW>
__global__ __launch_bounds__(64 * NF) void k_root_gain_w(WSweepArgs sp, GStreamArgs gp) {
    if (blockIdx.x == 0) wsweep_body<NF, CS, true>(sp);
    else gain_stream_body<NF, NF - 1, TPW>(gp, (int)blockIdx.x - 1);
}

}  // namespace msckf
