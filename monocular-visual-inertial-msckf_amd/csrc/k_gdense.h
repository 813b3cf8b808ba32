// K6-K7 on a DENSE source of rows alone (the remainder rows of split long tracks: launch_gain_t2_early, launch_gain_chain_dense),
// two row blocks per exchange.
//   reference MSCKF.py:604-607, :612-614 (k_gstream.h restates them as the sequential block update this kernel runs)
// A row block of k_gstream.h costs ~7 us, half of it the ONE all-to-all exchange of its Y tiles.  Here blocks a = 2 p and
// b = 2 p + 1 share an exchange: both Y_a = P^(a) T_a^T and Y^_b = P^(a) T_b^T go out together (Y^_b is short of block a's update),
// and with W = T_b X_a (16 x 16: V = T_b Y_a follows block a's pivots like any tile, in the publishing wavefront, which is idle then)
//     Y_b = Y^_b - X_a W^T,     A_b = T_b Y^_b - W W^T + sigma^2 I
// are what the sequential update would have formed after block a; block a's rank-16 update of P runs while block b is eliminated.
// Per pair: one exchange, four workgroup barriers (two blocks of k_gstream.h: two and six), 32 matrix-core instructions per tile
// wavefront instead of 24.  Measured (tools/ubench/gstream_test.hip N band us_per_row reps nb2 1): 7.1 -> 6.1 us per block at N = 30,
// 7.7 -> 6.9 at N = 33; results to 1e-14 of the dense evaluation of MSCKF.py:604-614, odd block counts included.
// One tile per wavefront, the publishing and the eliminating wavefront without one (ns <= 14 strips): windows of up to 33 clones.
#pragma once
#include "k_gstream.h"

namespace msckf {

__host__ __device__ inline size_t gdense_lds_doubles(int ns, int nb) {
    // multipliers | 1 / l_cc | control | three partial-tile areas [nb][256] | X_a, X_b [ns][GS_XT] | W in operand order | T_b Y^_b
    return 592 + (size_t)3 * nb * 256 + (size_t)2 * ns * GS_XT + GS_XT + 256;
}

__global__ __launch_bounds__(64 * GS_WAVES) void k_gain_dense(GStreamArgs p) {
    constexpr int WV = GS_WAVES;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int r = (int)blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = lane >> 4, cc = lane & 15;
    const int d = p.d, dc = p.dc, nb = p.nb, ns = p.ns;
    double* sW = smem;                                   // [2][16][17]
    double* sRi = smem + 544;                            // [2][16]
    typedef __attribute__((address_space(3))) int gs_lds_int;
    volatile gs_lds_int* sCtl = (volatile gs_lds_int*)(gs_lds_int*)(int*)(smem + 576);     // [0] bad pivot, [1] timeout, [2] pairs whose Y partials the publisher has read
    double* sPA = smem + 592;                            // [nb][256]  partials of Y_a, then of A_a = T_a Y_a
    double* sPB = sPA + (size_t)nb * 256;                // [nb][256]  partials of Y^_b, then of V = T_b Y_a
    double* sPC = sPB + (size_t)nb * 256;                // [nb][256]  partials of T_b Y^_b
    double* sXa = sPC + (size_t)nb * 256;                // [ns][GS_XT]
    double* sXb = sXa + (size_t)ns * GS_XT;              // [ns][GS_XT]
    double* sWt = sXb + (size_t)ns * GS_XT;              // [GS_XT]    W, [column][row] like an X tile
    double* sAb = sWt + GS_XT;                           // [256]      T_b Y^_b + sigma^2 I, summed
    const unsigned sw_addr = lds_addr(sW);
    auto g0 = [&](int s) { return s == 0 ? 0 : 15 + 16 * (s - 1); };
    auto nrows = [&](int s) { return s == 0 ? 15 : min(16, d - (15 + 16 * (s - 1))); };
    const int gr = g0(r), nr = nrows(r);
    const int s = wv;                                    // this wavefront's tile P(s, r); wavefronts ns .. 15 have none
    const bool tile = s < ns;

    double Pt[4];
    {
        const int gsr = g0(min(s, ns - 1)), ms = tile ? nrows(s) : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = g + 4 * i;
            const bool ok = (m < ms) && (cc < nr);
            const size_t row = ok ? gsr + m : 0, col = ok ? gr + cc : 0;
            const double x1 = p.P[row * p.ldp + col], x2 = p.P[col * p.ldp + row];
            Pt[i] = ok ? 0.5 * (x1 + x2) : 0.0;
            if (p.dx0) {
                if (s == 0 && m == 15 && cc < nr) Pt[i] = -p.dx0[gr + cc];
                if (r == 0 && cc == 15 && tile && m < ms) Pt[i] = -p.dx0[gsr + m];
            }
        }
    }
    if (t < 32) { sRi[t] = 0.0; sW[(t >> 4) * 272 + (t & 15) * 17 + 16] = __longlong_as_double(CHOL16_UNSET); }
    if (t < 4) sCtl[t] = 0;
    __syncthreads();

    const long long t_start = wall_clock64();
    bool failed = false;
    const int nb2 = p.nb2_dev ? min(p.nb2, *p.nb2_dev) : p.nb2;
    const int npairs = (nb2 + 1) / 2;
    const bool clone_tile = tile && s >= 1;              // strips 1 .. hold T's columns
    for (int pr = 0; pr < npairs; ++pr) {
        const bool hasb = 2 * pr + 1 < nb2;
        const int rowa = 32 * pr, rowb = rowa + 16;
        const unsigned long long tag = ((unsigned long long)p.epoch << 32) | (unsigned)(pr + 1);
        // ---- B: partials of Y_a[r], Y^_b[r] ------------------------------------------------------------------
        double Ta[4], Tb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { Ta[u] = 0.0; Tb[u] = 0.0; }
        if (clone_tile) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int col = 16 * (s - 1) + 4 * u + g;
                const bool ok = col < dc;
                const double xa = gs_ldd(p.T2 + (ok ? (size_t)(rowa + cc) * p.ldt2 + col : 0));
                const double xb = gs_ldd(p.T2 + ((ok && hasb) ? (size_t)(rowb + cc) * p.ldt2 + col : 0));
                Ta[u] = ok ? xa : 0.0;
                Tb[u] = (ok && hasb) ? xb : 0.0;
            }
            v4d aa = {0.0, 0.0, 0.0, 0.0}, ab = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 4; ++u) aa = __builtin_amdgcn_mfma_f64_16x16x4f64(Pt[u], Ta[u], aa, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) ab = __builtin_amdgcn_mfma_f64_16x16x4f64(Pt[u], Tb[u], ab, 0, 0, 0);
            double* da = sPA + (size_t)(s - 1) * 256 + lane;
            double* db = sPB + (size_t)(s - 1) * 256 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) { da[64 * i] = aa[i]; db[64 * i] = ab[i]; }
        }
        double rhsa = 0.0, rhsb = 0.0;
        if (wv == WV - 2 && r == 0 && g == 3) {          // r_n of the two blocks: the dx row's share of Y
            const double xa = gs_ldd(p.T2 + (size_t)(rowa + cc) * p.ldt2 + dc);
            const double xb = gs_ldd(p.T2 + (hasb ? (size_t)(rowb + cc) * p.ldt2 + dc : 0));
            rhsa = xa; rhsb = hasb ? xb : 0.0;
        }
        __syncthreads();
        // ---- C: publish Y_a[r], Y^_b[r] (payload write-through, drained, then ONE flag: k_gstream.h) ------------
        if (wv == WV - 2) {
            double ya[4] = {0.0, 0.0, 0.0, 0.0}, yb[4] = {0.0, 0.0, 0.0, 0.0};
            gs_sum_tiles(ya, sPA + lane, nb);
            gs_sum_tiles(yb, sPB + lane, nb);
            ya[3] += rhsa; yb[3] += rhsb;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) sCtl[2] = pr + 1;             // (the two areas are free for the partials of A_a and V)
            double* da = p.ex + ((size_t)(2 * pr) * ns + r) * 256 + lane;
            double* db = p.ex + ((size_t)(2 * pr + 1) * ns + r) * 256 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) { gs_std(da + 64 * i, ya[i]); gs_std(db + 64 * i, yb[i]); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) gs_st(p.exflag + (size_t)pr * ns + r, tag);
        }
        // ---- D: fetch Y_a[s], Y^_b[s]; partials of A_a, V = T_b Y_a, T_b Y^_b ---------------------------------------
        double Ya[1][4], Yb[1][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { Ya[0][i] = 0.0; Yb[0][i] = 0.0; }
        if (tile) {
            const unsigned long long* fl = p.exflag + (size_t)pr * ns + s;
            bool ok = true;
            for (;;) {
                if (gs_ld(fl) == tag) break;
                if (wall_clock64() - t_start > GS_TIMEOUT_TICKS) { if (lane == 0) sCtl[1] = 1; ok = false; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (ok) {
                const double* sa = p.ex + ((size_t)(2 * pr) * ns + s) * 256 + lane;
                const double* sb = p.ex + ((size_t)(2 * pr + 1) * ns + s) * 256 + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) { Ya[0][i] = gs_ldd(sa + 64 * i); Yb[0][i] = gs_ldd(sb + 64 * i); }
            }
            if (clone_tile) {
                for (int spins = 0; sCtl[2] != pr + 1 && spins < (1 << 22); ++spins) __builtin_amdgcn_s_sleep(1);
                v4d aa = {0.0, 0.0, 0.0, 0.0}, vv = {0.0, 0.0, 0.0, 0.0}, ab = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int u = 0; u < 4; ++u) aa = __builtin_amdgcn_mfma_f64_16x16x4f64(Ta[u], Ya[0][u], aa, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) vv = __builtin_amdgcn_mfma_f64_16x16x4f64(Tb[u], Ya[0][u], vv, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) ab = __builtin_amdgcn_mfma_f64_16x16x4f64(Tb[u], Yb[0][u], ab, 0, 0, 0);
                double* da = sPA + (size_t)(s - 1) * 256 + lane;
                double* dv = sPB + (size_t)(s - 1) * 256 + lane;
                double* db = sPC + (size_t)(s - 1) * 256 + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) { da[64 * i] = aa[i]; dv[64 * i] = vv[i]; db[64 * i] = ab[i]; }
            }
        }
        __syncthreads();
        // ---- E1: eliminate A_a; X_a[s] = Y_a[s] L_a^-T; W = V L_a^-T -------------------------------------------------
        const unsigned swa = sw_addr, swb = sw_addr + 16 * 17 * 8;
        double* sria = sRi;
        double* srib = sRi + 16;
        if (wv == WV - 1) {
            double a[4] = {0.0, 0.0, 0.0, 0.0};
            gs_sum_tiles(a, sPA + lane, nb);
#pragma unroll
            for (int i = 0; i < 4; ++i) if (g + 4 * i == cc) a[i] += p.sigma2;
            if (lane < 16) { sW[272 + lane * 17 + 16] = __longlong_as_double(CHOL16_UNSET); sRi[16 + lane] = 0.0; }     // block b's words
            const bool bad = gs_eliminate<16>(a, swa, sria, lane);
            if (bad && lane == 0) sCtl[0] = 1;
        }
        if (wv == WV - 2) {
            // the publisher has no tile (ns <= 14): it sums V and T_b Y^_b beside the elimination and lets V follow the pivots
            double w1[1][4] = {{0.0, 0.0, 0.0, 0.0}};
            double abk[4] = {0.0, 0.0, 0.0, 0.0};
            gs_sum_tiles(w1[0], sPB + lane, nb);
            gs_sum_tiles(abk, sPC + lane, nb);
#pragma unroll
            for (int i = 0; i < 4; ++i) { if (g + 4 * i == cc) abk[i] += p.sigma2; sAb[64 * i + lane] = abk[i]; }
            gs_follow<16, 1>(w1, swa, sria, lane);
            double* dst = sWt + cc * GS_XS + g;
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[4 * i] = w1[0][i];
        }
        if (tile) {
            gs_follow<16, 1>(Ya, swa, sria, lane);
            double* dst = sXa + (size_t)s * GS_XT + cc * GS_XS + g;
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[4 * i] = Ya[0][i];
        }
        __syncthreads();
        // ---- E2: Y_b = Y^_b - X_a W^T; A_b = T_b Y^_b - W W^T + sigma^2 I; eliminate; X_b ----------------------------
        {
            const double* wr = sWt + g * GS_XS + cc;
            double wop[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) wop[u] = wr[4 * GS_XS * u];
            if (wv == WV - 1) {
                v4d acc = {sAb[lane], sAb[64 + lane], sAb[128 + lane], sAb[192 + lane]};
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-wop[u], wop[u], acc, 0, 0, 0);
                double a[4] = {acc[0], acc[1], acc[2], acc[3]};
                if (lane < 16) { sW[lane * 17 + 16] = __longlong_as_double(CHOL16_UNSET); sRi[lane] = 0.0; }          // the next pair's block a
                const bool bad = gs_eliminate<16>(a, swb, srib, lane);
                if (bad && lane == 0) sCtl[0] = 1;
            }
            if (tile) {
                const double* xs = sXa + (size_t)s * GS_XT + g * GS_XS + cc;
                v4d acc = {Yb[0][0], Yb[0][1], Yb[0][2], Yb[0][3]};
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xs[4 * GS_XS * u], wop[u], acc, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) Yb[0][i] = acc[i];
                {   // block a's rank-16 update while block b is eliminated
                    const double* xra = sXa + (size_t)r * GS_XT + g * GS_XS + cc;
                    v4d pa = {Pt[0], Pt[1], Pt[2], Pt[3]};
#pragma unroll
                    for (int u = 0; u < 4; ++u) pa = __builtin_amdgcn_mfma_f64_16x16x4f64(-xs[4 * GS_XS * u], xra[4 * GS_XS * u], pa, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) Pt[i] = pa[i];
                }
                gs_follow<16, 1>(Yb, swb, srib, lane);
                double* dst = sXb + (size_t)s * GS_XT + cc * GS_XS + g;
#pragma unroll
                for (int i = 0; i < 4; ++i) dst[4 * i] = Yb[0][i];
            }
        }
        __syncthreads();
        if (sCtl[0] | sCtl[1]) { failed = true; break; }
        // ---- F: P(s, r) -= X_b[s] X_b[r]^T  (block a's part went in beside block b's elimination) -----------------------
        if (tile) {
            const double* xrb = sXb + (size_t)r * GS_XT + g * GS_XS + cc;
            const double* xsb = sXb + (size_t)s * GS_XT + g * GS_XS + cc;
            v4d acc = {Pt[0], Pt[1], Pt[2], Pt[3]};
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xsb[4 * GS_XS * u], xrb[4 * GS_XS * u], acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) Pt[i] = acc[i];
        }
    }
    if (failed) {
        if (r == 0 && t == 0) { p.status[0] = sCtl[1] ? 2 : 1; if (p.status_h) p.status_h[0] = sCtl[1] ? 2 : 1; }
        return;
    }
    if (r == 0 && t == 0) { p.status[0] = 0; if (p.status_h) p.status_h[0] = 0; }
    if (tile) {
        const int gsr = g0(s), ms = nrows(s);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = g + 4 * i;
            if (m < ms && cc < nr) {
                p.Pout[(size_t)(gsr + m) * p.ldo + gr + cc] = Pt[i];
                if (p.Pout_h) p.Pout_h[(size_t)(gsr + m) * p.ldo + gr + cc] = Pt[i];
            }
        }
        if (s == 0 && g == 3 && cc < nr) { p.dx[gr + cc] = -Pt[3]; if (p.dx_h) p.dx_h[gr + cc] = -Pt[3]; }
    }
}

}  // namespace msckf
