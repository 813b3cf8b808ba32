// K5, upper part, general form of k_sweep.h (reference MSCKF.py:594-598): the same systolic
// pipeline -- NF wavefronts fold NF source triangles into the band R at NF different columns,
// one workgroup barrier per macro step, arithmetic exactly that of folding the triangles one
// after the other -- for
//   * wider source triangles: a register tile of W = 16 CS columns (CS = 4: tracks of up to 10
//     clone slots, CS = 6: up to 15 slots; local column W-1 holds the rhs), lane (rq, cq) =
//     (lane >> 4, lane & 15) owns rows {rq + 4 rr}, rr < 4 CS, and local columns {cq + 16 k}, k < CS;
//   * any number of clones: the band R lives in a RING of RC rows x W doubles in LDS (entry (c, col)
//     at [c mod RC][col - c], rhs at [c mod RC][W-1]).  The schedule is static, so the host
//     (sweep_flush_table) knows for every macro step t which rows no present or future fold step
//     touches any more; at the head of step t the wavefronts copy rows [flo(t), fhi(t)) to the
//     output block and clear their ring slots for rows RC further down.
// The column step is sweep_step.h's: the pivot column reaches the tile's lanes through the DPP row broadcast of
// the FMAs, the row lanes' partial dots are reduce-scattered with the lane swaps and every lane looks after one
// or two entries of the pivot row of R.
#pragma once
#include <hip/hip_runtime.h>
#include "k_sweep.h"

namespace msckf {

struct WSweepArgs {
    const SweepNode* nodes;
    const SweepFold* folds;
    int node_base;
    double* rbuf;
    const double* zero;         // a double that reads 0.0 (tail of the workspace)
    const int* flush;           // per node: nsteps + 1 entries (lo | n << 16), at flush_off[node]
    const int* flush_off;       // [nodes] offset into `flush`
    int rc_log2;                // ring rows = 1 << rc_log2
    // k_wsweep body with PUB = true (the root sweep inside k_root_gain_w): rows leave with write-through stores and wavefront 0
    // publishes how many are final AND visible device-wide (k_gstream.h follows them block by block)
    unsigned long long* progress;
    unsigned epoch;
    long long* tstamp;          // optional: [0] start, [1] last row published (10 ns wall-clock ticks)
};

constexpr int WS_PUB_LAG = 4;             // steps between a row's store and the wait for it (the count goes out one step later)
template <int CS> struct WSweepGeom {
    static constexpr int W = 16 * CS;          // tile columns = LDS row stride
    static constexpr int MAX_W = W - 6;        // widest source / envelope
    static constexpr int RSL = 4 * CS;         // row slots of a lane
    static constexpr int NCH = 2 * CS;         // chunks of 8 columns
};

template <int CS>
__host__ __device__ inline size_t wsweep_lds_bytes(int rc, int nf, int nsteps) {
    // pad | ring | dump words | zero words | flush table (ints)
    return ((size_t)(rc + 1) * WSweepGeom<CS>::W + (size_t)nf * 64 + 2) * 8 + ((size_t)nsteps + 2) * 8;   // (flush + publish tables)
}

template <int NF, int CS, bool PUB>
__device__ __forceinline__ void wsweep_body(const WSweepArgs& p) {
    using G = WSweepGeom<CS>;
    constexpr int W = G::W, RSL = G::RSL;
    constexpr int CL = 16;
    constexpr bool HAS2 = CS > 4;          // lanes rq < 2 look after a second column slot (4 + rq)
    static_assert(CS >= 4 && CS <= 6, "column slots 4..6");
    static_assert((NF & (NF - 1)) == 0, "NF must be a power of two");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SweepNode nd = p.nodes[p.node_base + blockIdx.x];
    const int t = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lane = t & 63;
    const int rq = lane >> 4;
    const int cq = lane & 15;
    const int RC = 1 << p.rc_log2, RCM = RC - 1;
    // (one row of padding in front of the ring: a lane whose column has retired keeps walking its band
    //  offset below zero; with the pivot in ring row 0 that address must still be inside the allocation)
    double* Rb = smem + W;                                    // [RC][W]
    constexpr int RB0 = W;                                    // index of the ring in smem
    const int dump_i = (RC + 1) * W + wv * 64 + lane;
    const int zero_i = (RC + 1) * W + NF * 64;
    int* ftab = reinterpret_cast<int*>(smem + (size_t)(RC + 1) * W + NF * 64 + 2);
    const int nsteps = __builtin_amdgcn_readfirstlane(nd.nsteps);
    const int fold_end = __builtin_amdgcn_readfirstlane(nd.fold_end);
    const int wtot = __builtin_amdgcn_readfirstlane(nd.wtot);
    double* out = p.rbuf + nd.out_off;
    const int ldo = wtot + 1;

    if constexpr (PUB) { if (p.tstamp && t == 0) p.tstamp[0] = wall_clock64(); }
    for (int e = t; e < (RC + 1) * W; e += 64 * NF) smem[e] = 0.0;
    if (t < 2) smem[zero_i + t] = 0.0;
    {
        const int* src = p.flush + p.flush_off[p.node_base + blockIdx.x];
        for (int e = t; e < 2 * (nsteps + 1); e += 64 * NF) ftab[e] = src[e];     // flush entries, then the publish entries
    }
    // the node's first triangle (t0 == 0) is adopted: its rows ARE the first rows of R, nothing to eliminate
    const SweepFold f0 = p.folds[nd.fold_begin];
    const int adopt = (nd.fold_end > nd.fold_begin && f0.t0 == 0) ? 1 : 0;
    if (adopt) {
        __syncthreads();
        const double* src = p.rbuf + f0.src_off;
        const int ldw = f0.w + 1;
        for (int e = t; e < f0.w * ldw; e += 64 * NF) {
            const int r = e / ldw, lc = e - r * ldw;
            if (lc >= r) Rb[(size_t)((f0.off + r) & RCM) * W + (lc == f0.w ? W - 1 : lc - r)] = src[e];
        }
    }

    // The head (row slots 0, 1) of a wavefront's NEXT fold is fetched during the last chunks of the current one
    // only where the registers allow it (CS = 4); the 90-column tile loads it at the start of the fold.
    constexpr bool PREFETCH_HEAD = (CS == 4);
    double a[RSL][CS];
    double nxt[PREFETCH_HEAD ? 2 : 1][PREFETCH_HEAD ? CS : 1];
#pragma unroll
    for (int rr = 0; rr < RSL; ++rr)
#pragma unroll
        for (int k = 0; k < CS; ++k) a[rr][k] = 0.0;

    int tcur = 0;                       // macro steps (= barriers) this wavefront has done
    int f_off = 0, f_w = 0, f_ew = 0, f_t0 = 0;
    const double* f_src = p.rbuf;
    int n_off = 0, n_w = 0, n_ew = 0, n_t0 = 0;
    const double* n_src = p.rbuf;

    auto read_desc = [&](int fi, int& o_off, int& o_w, int& o_ew, int& o_t0, const double*& o_src) {
        const SweepFold f = p.folds[fi];
        o_off = __builtin_amdgcn_readfirstlane(f.off);
        o_w = __builtin_amdgcn_readfirstlane(f.w);
        o_ew = __builtin_amdgcn_readfirstlane(f.ew);
        o_t0 = __builtin_amdgcn_readfirstlane(f.t0);
        o_src = p.rbuf + f.src_off;
    };
    auto load_elem = [&](const double* src, int w, int rr, int k) -> double {
        const int r = rq + 4 * rr, lc = cq + CL * k;
        const bool isr = (k == CS - 1) && (cq == CL - 1);
        const bool ok = (r < w) && (isr || (lc >= r && lc < w));
        const int col = isr ? w : lc;
        const double* q = ok ? src + (r * (w + 1) + col) : p.zero;
        return *q;
    };
    auto fetch_next_head = [&]() {      // row slots 0, 1 of the next fold
        if constexpr (PREFETCH_HEAD) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int k = 0; k < CS; ++k) nxt[rr][k] = load_elem(n_src, n_w, rr, k);
        }
    };

    // PUB: a wavefront that stored rows at step ts waits for its stores PUB_LAG steps later (`pend`, one bit per step: by
    // then the round trip is over, and so are the prefetches of the fold issued before it), and one step after that --
    // behind the barrier that orders every wavefront's wait -- wavefront 0 publishes the rows that were final at step ts.
    constexpr int PUB_LAG = WS_PUB_LAG;
    unsigned pend = 0;
    int e_nx = 0, p_nx = 0;                                // table entries of the coming step (read one step ahead)
    const int* ptab = ftab + nsteps + 1;                   // rows to publish at the head of step ts (0: nothing), made by the host
    auto store_out = [&](double* q, double x) {
        if constexpr (PUB) __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)(unsigned long long*)q,
                                              (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *q = x;
    };
    auto publish = [&](int rows) {
        if (wv == 0 && lane == 0)
            __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)p.progress,
                               ((unsigned long long)p.epoch << 32) | (unsigned)rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // rows [lo, lo + n) are final at the head of macro step ts: out to HBM, ring slots cleared
    auto flush_rows = [&](int ts) {
        if constexpr (PUB) {
            if (pend & 1u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pend >>= 1;
        }
        // (the table entries of step ts were requested during step ts - 1: a dependent LDS read at the head of every step
        //  cost the 96-column root 0.12 us per step once the publish entry sat beside the flush entry)
        const int e = e_nx;
        const int pr_now = p_nx;
        if (ts < nsteps) {
            e_nx = ftab[ts + 1];
            if constexpr (PUB) p_nx = ptab[ts + 1];
        }
        if constexpr (PUB) {
            // (the rows that were final PUB_LAG + 1 steps ago, where that count passes a 16-row block boundary of the reader:
            //  the host's sweep_publish_table says so, one entry per step beside the flush entry)
            const int pr = __builtin_amdgcn_readfirstlane(pr_now);
            if (pr) publish(pr);
        }
        const int n = __builtin_amdgcn_readfirstlane(e >> 16);
        if (n == 0) return;
        const int lo = __builtin_amdgcn_readfirstlane(e & 0xFFFF);
        for (int r = (wv + ts) & (NF - 1); r < n; r += NF) {   // (rotating: a single row is not always wavefront 0's)
            const int c = lo + r;
            double* Rrow = Rb + (size_t)(c & RCM) * W;
            double* orow = out + (size_t)c * ldo;
#pragma unroll
            for (int h = 0; h < (W + 63) / 64; ++h) {
                const int dlt = lane + 64 * h;
                if (dlt < W) {
                    const double x = Rrow[dlt];
                    if (dlt == W - 1) store_out(orow + wtot, x);
                    else if (c + dlt < wtot) store_out(orow + c + dlt, x);
                    Rrow[dlt] = 0.0;
                }
            }
            if constexpr (PUB) pend |= 1u << (PUB_LAG - 1);   // waited for at step ts + PUB_LAG, published one barrier later
        }
    };
    auto barrier_step = [&]() {
        __syncthreads();
        ++tcur;
        flush_rows(tcur);
    };

    // The R entries this lane looks after: slot rq (local column lco1) and, on the 90-column tile, slot 4 + (rq & 1)
    // (rows 2, 3 mirror rows 0, 1: same reads, same tau, no write).
    int ra1 = 0, wa1 = 0, st1 = 0, wr1 = 0;
    int ra2 = 0, wa2 = 0, st2 = 0, wr2 = 0, sw2 = 0, ww2 = 0;
    const int lco1 = cq + CL * rq;
    const int lco2 = cq + CL * (4 + (rq & 1));
    const bool isr1 = (CS == 4) && (rq == 3) && (cq == CL - 1);
    const bool isr2 = HAS2 && (4 + (rq & 1) == CS - 1) && (cq == CL - 1);
    auto init_addr = [&]() {
        const int r0 = RB0 + (f_off & RCM) * W;
        {
            const bool valid = isr1 || lco1 < f_ew;
            ra1 = valid ? r0 + (isr1 ? W - 1 : lco1) : zero_i;
            wa1 = valid ? ra1 : dump_i;
            st1 = valid ? (isr1 ? W : W - 1) : 0;
            wr1 = valid ? RC * W : 0;
        }
        if constexpr (HAS2) {
            const bool valid = isr2 || lco2 < f_ew;
            const bool mine = valid && rq < 2;
            ra2 = valid ? r0 + (isr2 ? W - 1 : lco2) : zero_i;
            wa2 = mine ? ra2 : dump_i;
            st2 = valid ? (isr2 ? W : W - 1) : 0;
            wr2 = valid ? RC * W : 0;
            sw2 = mine ? st2 : 0;
            ww2 = mine ? wr2 : 0;
        }
    };

    auto step = [&](auto tagk, auto tagj) {
        constexpr int KK = decltype(tagk)::value;
        constexpr int I = 8 * KK + decltype(tagj)::value;                        // the fold's column
        constexpr int NR = (2 * KK + 2 < RSL) ? 2 * KK + 2 : RSL;                // live row slots
        constexpr int K0 = I / CL, L = I % CL;                                   // the pivot column's slot / column lane
        const int prow = (f_off + I) & RCM;
        const int rrow = RB0 + prow * W;                   // pivot row of R (uniform)
        int l1 = lco1, l2 = lco2;
        asm volatile("" : "+v"(l1), "+v"(l2));            // (keeps the lane masks of all columns from being hoisted and spilled)
        const bool on1 = (l1 > I) || isr1;
        const bool on2 = HAS2 && ((l2 > I) || isr2);
        sweep_column_step<CS, RSL, NR, K0, L>(a, smem, rrow, ra1, wa1, on1, ra2, wa2, on2, dump_i, lane);
        ra1 += st1; wa1 += st1;
        if constexpr (HAS2) { ra2 += st2; wa2 += sw2; }
        if (prow == RCM) {                                  // the next pivot row wraps around the ring
            ra1 -= wr1; wa1 -= wr1;
            if constexpr (HAS2) { ra2 -= wr2; wa2 -= ww2; }
        }
    };

    // one chunk of 8 columns: fetch the two row slots the NEXT chunk's first column needs, then the steps
    auto chunk = [&](auto tagk, bool have_next) {
        constexpr int KK = decltype(tagk)::value;
        if constexpr (KK < G::NCH) {
            if (8 * KK >= f_ew) return;
#pragma unroll
            for (int rr = 2 * KK + 2; rr <= 2 * KK + 3 && rr < RSL; ++rr) {
#pragma unroll
                for (int k = 0; k < CS; ++k) {
                    if (CL * k + CL - 1 >= 4 * rr) a[rr][k] = load_elem(f_src, f_w, rr, k);   // else structurally zero, never read
                }
            }
            if (have_next && KK == max((f_ew - 1) / 8 - 1, 0)) fetch_next_head();
            // (steps run over the ENVELOPE: the tile's rows fill in right of the source's last column wherever R
            //  already reaches further, and that fill has to be eliminated too)
            auto one = [&](auto tagj) {
                if (8 * KK + decltype(tagj)::value < f_ew) {
                    step(tagk, tagj);
                    barrier_step();
                }
            };
            one(STag<0>{}); one(STag<1>{}); one(STag<2>{}); one(STag<3>{});
            one(STag<4>{}); one(STag<5>{}); one(STag<6>{}); one(STag<7>{});
        }
    };

    __syncthreads();                                       // R zeroed / adopted, flush table in place
    e_nx = ftab[0];
    if constexpr (PUB) p_nx = ptab[0];
    flush_rows(0);
    int fi = nd.fold_begin + adopt + wv;
    bool have = fi < fold_end;
    if (have) {
        read_desc(fi, n_off, n_w, n_ew, n_t0, n_src);
        fetch_next_head();
    }
    while (have) {
        f_off = n_off; f_w = n_w; f_ew = n_ew; f_t0 = n_t0; f_src = n_src;
        while (tcur < f_t0) barrier_step();
        init_addr();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int k = 0; k < CS; ++k) {
                if constexpr (PREFETCH_HEAD) a[rr][k] = nxt[rr][k];
                else a[rr][k] = load_elem(f_src, f_w, rr, k);
            }
        // (the host schedules t0 >= 1 and one spare step per slot reuse)
        fi += NF;
        const bool have_next = fi < fold_end;
        if (have_next) read_desc(fi, n_off, n_w, n_ew, n_t0, n_src);
        chunk(STag<0>{}, have_next);
        chunk(STag<1>{}, have_next);
        chunk(STag<2>{}, have_next);
        chunk(STag<3>{}, have_next);
        chunk(STag<4>{}, have_next);
        chunk(STag<5>{}, have_next);
        chunk(STag<6>{}, have_next);
        chunk(STag<7>{}, have_next);
        chunk(STag<8>{}, have_next);
        chunk(STag<9>{}, have_next);
        chunk(STag<10>{}, have_next);
        chunk(STag<11>{}, have_next);
        have = have_next;
    }
    while (tcur < nsteps) barrier_step();
    // (the table's last entry, index nsteps, covers every row still in the ring)
    if constexpr (PUB) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        publish(wtot);
        if (p.tstamp && t == 0) p.tstamp[1] = wall_clock64();
    }
}

template <int NF, int CS>
__global__ __launch_bounds__(64 * NF) void k_wsweep(WSweepArgs p) {
    wsweep_body<NF, CS, false>(p);
}

}  // namespace msckf
