// One column step of a fold (k_sweep.h, k_wsweep.h, k_lsweep.h): the Householder reflector that eliminates the
// tile's pivot column against its row of R, on a register tile laid out for the DPP row broadcast.
//
// Lane (rq, cq) = (lane >> 4, lane & 15) owns rows {rq + 4 rr} and local columns {cq + 16 k}, k < CS (CS = 4: 60
// columns + rhs, CS = 6: 90 + rhs; the rhs is the last local column).  The 16 column lanes of a row lane are one DPP
// row, so the pivot column -- column lane L of every row, slot K0 -- reaches the other 15 through the row_newbcast
// operand of v_fmac_f64 itself: the dots and the rank-1 update read it straight from its owner's registers and
// nothing of the tile goes through LDS (rounds 1-2 published the column in LDS and read it back: two LDS trips and
// NR / 2 wide reads on every step's dependent chain; headline k_sweep 95.5 -> 74.7 us per launch).  The broadcast
// lane is an immediate, hence one instantiation per column of the tile.  (A DPP broadcast reads garbage from a source lane
// that EXEC has switched off: the step runs with all 64 lanes active -- k_feature.h, which borrowed the trick, learnt it
// the hard way.)
//
// The four row lanes' partial dots are reduce-scattered with the gfx950 lane swaps: row r ends up with the dot of
// column slot r (and, CS = 6, rows r and r + 2 with that of slot 4 + r, r < 2), i.e. every lane looks after ONE or
// two entries of the pivot row of R: it reads them, forms tau, writes them back, and tau returns over the rows the
// same way.  The pivot column's own dot is v^T v: its squared norm costs no extra reduction.
#pragma once
#include <hip/hip_runtime.h>
#include "wave_ops.h"
#include "sweep_dpp_groups.h"

namespace msckf {

// 64-bit halves of the gfx950 lane swaps.  swap_rows32(x, y): x = [x rows 0,1 | y rows 0,1], y = [x rows 2,3 | y rows 2,3];
// swap_rows16(x, y): x = [x r0, y r0, x r2, y r2], y = [x r1, y r1, x r3, y r3]  (rows = the four 16-lane DPP rows).
__device__ __forceinline__ void swap_rows32(double& x, double& y) {
    const auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    x = __hiloint2double((int)h[0], (int)l[0]);
    y = __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ void swap_rows16(double& x, double& y) {
    const auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    x = __hiloint2double((int)h[0], (int)l[0]);
    y = __hiloint2double((int)h[1], (int)l[1]);
}
// a: the tile, NR live row slots; smem[rrow]: R(c, c); ra1 / wa1 (ra2 / wa2): where this lane's entry R(c, .) of
// slot rq (slot 4 + (rq & 1)) is read / written -- a zero word / the lane's dump word where it has none; on1 / on2:
// the entry is right of the pivot (or the rhs).  Rows 2, 3 mirror rows 0, 1 for the second entry (same reads, same
// tau; their wa2 is the dump word).
// (-DSWEEP_PROF: prof[1..4] collect the ticks of the dots, the reduction, the scalars + tau + R writes, the tau hand-back;
//  tools/sweep_prof.py prints them)
#ifdef SWEEP_PROF
#define SWEEP_STEP_TICK(slot) do { if (prof) { __builtin_amdgcn_sched_barrier(0); const long long tn_ = __builtin_readcyclecounter(); prof[slot] += tn_ - *tprev; *tprev = tn_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define SWEEP_STEP_TICK(slot) do { } while (0)
#endif
template <int CS, int RS, int NR, int K0, int L>
__device__ __forceinline__ void sweep_column_step(double (&a)[RS][CS], double* smem, int rrow, int ra1, int wa1, bool on1,
                                                  int ra2, int wa2, bool on2, int dump_i, int lane,
                                                  long long* prof = nullptr, long long* tprev = nullptr) {
    static_assert(CS == 4 || CS == 6, "60- or 90-column tiles");
    static_assert(K0 < CS && NR <= RS && L < 16, "pivot inside the tile");
    constexpr bool HAS2 = CS > 4;
    constexpr bool HAS1 = K0 <= 3;                       // slots 0..3 not all retired
    const double x0 = smem[rrow];
    double rck1 = 0.0, rck2 = 0.0;
    if constexpr (HAS1) rck1 = smem[ra1];
    if constexpr (HAS2) rck2 = smem[ra2];
    // ---- dots: one partial sum per live slot, the pivot column read through the row broadcast ----------
    double sp[CS];
#pragma unroll
    for (int k = 0; k < CS; ++k) sp[k] = 0.0;
    // (whole groups of row slots per assembly statement: sweep_dpp_groups.h says why)
    constexpr int NS = CS - K0;                          // live column slots
    {
        constexpr int G = DppGroupMax<NS>::dots;
#pragma unroll
        for (int rr = 0; rr + G <= NR; rr += G) DppDots<NS, G>::template dots<L, K0>(sp, a, rr);
        if constexpr (NR % G != 0) DppDots<NS, NR % G>::template dots<L, K0>(sp, a, NR - NR % G);
    }
    SWEEP_STEP_TICK(1);
    // ---- reduce-scatter over the four row lanes ---------------------------------------------------------
    double tot1 = 0.0, tot2 = 0.0;
    if constexpr (K0 <= 1) {
        swap_rows32(sp[0], sp[2]);
        swap_rows32(sp[1], sp[3]);
        double pA = sp[0] + sp[2], pB = sp[1] + sp[3];    // rows 0,1: slot 0 / 1;  rows 2,3: slot 2 / 3  (row pairs summed)
        swap_rows16(pA, pB);
        tot1 = pA + pB;
    } else if constexpr (HAS1) {
        // slots 0, 1 are retired: only rows 2, 3 need a total (rows 0, 1 get a copy nobody uses)
        swap_rows16(sp[2], sp[3]);
        double g = sp[2] + sp[3], h = g;                  // [s2 r0+r1, s3 r0+r1, s2 r2+r3, s3 r2+r3]
        swap_rows32(g, h);
        tot1 = g + h;
    }
    if constexpr (HAS2) {
        swap_rows16(sp[4], sp[5]);
        double g = sp[4] + sp[5], h = g;                  // [s4 r0+r1, s5 r0+r1, s4 r2+r3, s5 r2+r3]
        swap_rows32(g, h);
        tot2 = g + h;                                     // rows: [s4 s5 s4 s5]
    }
    // |pivot column|^2 = its dot with itself, on the lane that looks after it
    const double sg = HAS1 ? readlane_d(tot1, 16 * K0 + L) : readlane_d(tot2, 16 * (K0 - 4) + L);
    const bool live = sg > SWEEP_TINY;                    // wave-uniform; below: nothing to eliminate
    SWEEP_STEP_TICK(2);
    // ---- reflector scalars (every lane, uniform values) ---------------------------
    // sg > 1e-290 keeps ss = x0^2 + sg a normal number whose rsqrt / rcp seeds + one Newton step are good to a few
    // 1e-16 (the reflector stays orthogonal to that level).  A dead column (nothing to eliminate: the identity) may
    // send Inf / NaN down this chain; nothing of it is kept -- tau is selected to zero and the R writes are switched
    // off.  |x0| > 1e150 does not occur (R entries are bounded by the column norms of a normalised-coordinate
    // Jacobian stack).
    const double ss = fma(x0, x0, sg);
    const double y = fast_rsqrt(ss);
    const double nrm = ss * y;
    const double d = nrm + fabs(x0);
    const double beta = y * fast_rcp(d);                  // 1 / (nrm (nrm + |x0|))
    const double v0 = __builtin_copysign(d, x0);          // x0 - alpha with alpha = -sign(x0) nrm
    // ---- tau of this lane's column(s), its R entries, then the rank-1 update of every slot ----------
    double tau1 = 0.0, tau2 = 0.0;
    if constexpr (HAS1) {
        const bool act = on1 && live;
        tau1 = act ? beta * fma(v0, rck1, tot1) : 0.0;
        smem[act ? wa1 : dump_i] = fma(-tau1, v0, rck1);
    }
    if constexpr (HAS2) {
        const bool act = on2 && live;
        tau2 = act ? beta * fma(v0, rck2, tot2) : 0.0;
        smem[act ? wa2 : dump_i] = fma(-tau2, v0, rck2);
    }
    if (lane == 0 && live) smem[rrow] = x0 - v0;
    SWEEP_STEP_TICK(3);
    // tau of slot k on every row, by the lane swaps
    double nt[CS];
    if constexpr (K0 <= 1) {
        // [t0 t1 t2 t3] by rows -> [t0 t0 t2 t2], [t1 t1 t3 t3] -> four uniform-by-row copies
        double e = tau1, o = e;
        swap_rows16(e, o);
        nt[0] = e; nt[2] = e; nt[1] = o; nt[3] = o;
        swap_rows32(nt[0], nt[2]);
        swap_rows32(nt[1], nt[3]);
    } else if constexpr (HAS1) {
        // only slots 2, 3 are live: -> [t0 t1 t0 t1], [t2 t3 t2 t3] -> [t2 t2 t2 t2], [t3 t3 t3 t3]
        double e = tau1, o = e;
        swap_rows32(e, o);
        nt[0] = e; nt[1] = e; nt[2] = o; nt[3] = o;
        swap_rows16(nt[2], nt[3]);
    }
    if constexpr (HAS2) {
        nt[4] = tau2; nt[5] = tau2;                       // [t4 t5 t4 t5] by rows -> [t4 x4], [t5 x4]
        swap_rows16(nt[4], nt[5]);
    }
    SWEEP_STEP_TICK(6);
    // a[rr][k] -= (pivot column, row slot rr) * tau_k  (slot K0 last in its row: it rewrites the register the other
    // slots read through the broadcast; the owner's own column has tau = 0 and stays as it is -- it is retired)
#pragma unroll
    for (int k = 0; k < K0; ++k) nt[k] = 0.0;
    {
        constexpr int G = DppGroupMax<NS>::update;
#pragma unroll
        for (int rr = 0; rr + G <= NR; rr += G) DppUpdate<NS, G>::template update<L, K0>(a, nt, rr);
        if constexpr (NR % G != 0) DppUpdate<NS, NR % G>::template update<L, K0>(a, nt, NR - NR % G);
    }
}

}  // namespace msckf
