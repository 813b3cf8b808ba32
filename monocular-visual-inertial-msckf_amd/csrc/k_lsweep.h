// K5, leaves of the band plan (reference MSCKF.py:594-598, the Householder QR of the stacked rows): the
// systolic fold of k_wsweep.h applied to the K4 blocks themselves.  A leaf node owns a run of sorted features
// with the same first clone slot and a window of w columns (+ rhs); its workgroup keeps the leaf's R (w x W
// band storage) in LDS and folds "row blocks" into it: up to RB = 4 RSLOTS stacked rows [H_o | r_o] of a few
// consecutive accepted features, held in the REGISTERS of one wavefront (lane (rq, cq) = (lane >> 4, lane & 15)
// owns rows {rq + 4 rr} and window columns {cq + 16 k}; the column step is sweep_step.h's).  NF wavefronts fold NF row blocks at NF consecutive
// columns (block b runs column c at macro step t0(b) + c, t0(b) = 1 + (b / NF)(w + gap) + b % NF), one workgroup
// barrier per macro step; the arithmetic is that of folding the blocks one after the other.  Unlike a
// triangle's, all rows of a block are alive from the first column on; columns retire in chunks of 8 as in
// k_sweep.  While a wavefront runs the last chunk of a block it gathers its next block from the stack
// (row-major q x (6M+1) per feature, k_feature.h) straight into registers: per row the block offset and the
// 16-entry map "window slot -> view" of its feature (FeatInfo, written by the host at upload).
#pragma once
#include <hip/hip_runtime.h>
#include "k_fold.h"
#include "k_sweep.h"

namespace msckf {

struct __attribute__((aligned(16))) FeatInfo {
    long long blk_off;          // offset (scalars) of the feature's stack block
    int M;                      // views
    int pad;
    unsigned char col[16];      // col[j] = view whose clone slot is (first slot of the feature) + j, 0xFF: none
};

struct LSweepArgs {
    const FoldNode* nodes;      // leaf nodes (kind 0)
    int node_base;
    const FeatInfo* info;       // [F] sorted feature order
    const void* stack;
    int stack_f32;
    const int* rank;            // [F]
    const unsigned char* accepted;
    double* rbuf;
    int wide;                   // this launch takes the nodes with w + 1 > 64 (1) or <= 64 (0); the others exit at once
    long long zero_idx;         // index (scalars) of a zero word behind the last block: absent entries load it (no branches)
};

constexpr int LS_MAXF = 256;    // features per leaf node
constexpr int LS_MAXB = 128;    // row blocks per leaf node
constexpr int LS_FB = 12;       // features per row block
constexpr int LS_RS4 = 9;       // row slots of the 60-column leaf tile: 36 rows (two tracks of 10 views)
constexpr int LS_RS6 = 8;       // row slots of the 90-column leaf tile: 32 rows (one track of up to 16 views)
constexpr int LS_RS6T = 14;     // ... of its tall form: 56 rows (two tracks of 15 views), eight wavefronts (168 registers of tile)

template <int CS, int RSLOTS> struct LSweepGeom {
    static constexpr int W = 16 * CS;
    static constexpr int RB = 4 * RSLOTS;
};

template <int CS, int RSLOTS>
__host__ __device__ inline size_t lsweep_lds_bytes(int nf) {
    using G = LSweepGeom<CS, RSLOTS>;
    // pad row | R | dump | zero | qrow | blkfirst, nblk | per-wave row tables (base, M, cols)
    return ((size_t)(G::W + 1) * G::W + (size_t)nf * 64 + 2) * 8 + (size_t)(LS_MAXF + 2 + LS_MAXB + 4) * 4 +
           (size_t)nf * G::RB * (4 + 4 + 16);
}

template <int NF, int CS, int RSLOTS, bool PREF_ = (CS == 4)>
__global__ __launch_bounds__(64 * NF) void k_lsweep(LSweepArgs p) {
    using G = LSweepGeom<CS, RSLOTS>;
    constexpr int W = G::W, RB = G::RB;
    constexpr int CL = 16;
    constexpr bool HAS2 = CS > 4;
    constexpr int NCH = 2 * CS;
    // PREF: a wavefront gathers its next block into a second register tile during the last chunk of the current
    // one.  The 90-column tile has no room for that: there the rounds are aligned (every wavefront finishes its
    // block, 7 idle steps), then all gather at the same time straight into the tile -- one load latency per round.
    constexpr bool PREF = PREF_;
    constexpr int RGAP = PREF ? 1 : NF;             // macro steps between two rounds of a fold slot beyond the w of the fold itself
    constexpr int KGMIN = 1;                        // first chunk whose instance may hold the gathered next block beside the tile
    static_assert(CS >= 4 && CS <= 6, "column slots 4..6");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const FoldNode nd = p.nodes[p.node_base + blockIdx.x];
    if ((nd.w + 1 > 64 ? 1 : 0) != p.wide) return;
    const int t = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lane = t & 63;
    const int rq = lane >> 4, cq = lane & 15;
    const int w = __builtin_amdgcn_readfirstlane(nd.w);
    const int nfeat = min(nd.src_end - nd.src_begin, LS_MAXF);
    constexpr int RB0 = W;
    double* Rb = smem + W;                                     // [W][W] (rows 0..w-1 used)
    const int dump_i = (W + 1) * W + wv * 64 + lane;
    const int zero_i = (W + 1) * W + NF * 64;
    int* qrow = reinterpret_cast<int*>(smem + (size_t)(W + 1) * W + NF * 64 + 2);   // [LS_MAXF + 1] row prefix
    int* blkfirst = qrow + LS_MAXF + 2;                        // [LS_MAXB + 1]
    int* s_nblk = blkfirst + LS_MAXB + 2;
    int* rowbase = s_nblk + 2 + wv * RB;                       // per wave: [RB] block offset of the row (scalars, < 2^31)
    int* rowM = s_nblk + 2 + NF * RB + wv * RB;                // per wave: [RB] views of the row's feature (0: no row)
    unsigned char* rowcols = reinterpret_cast<unsigned char*>(s_nblk + 2 + 2 * NF * RB) + wv * RB * 16;   // per wave: [RB][16]
    double* out = p.rbuf + nd.out_off;
    const int ldo = w + 1;

    for (int e = t; e < (w + 1) * W; e += 64 * NF) smem[e] = 0.0;
    if (t < 2) smem[zero_i + t] = 0.0;
    for (int i = t; i < nfeat; i += 64 * NF) {
        const int f = nd.src_begin + i;
        qrow[i + 1] = (p.accepted[f] == 1) ? 2 * p.info[f].M - p.rank[f] : 0;
    }
    __syncthreads();
    if (t == 0) {
        // greedy packing of consecutive features into row blocks of at most RB rows / LS_FB features
        int acc = 0, b = 0, rows = 0, cnt = 0;
        qrow[0] = 0;
        blkfirst[0] = 0;
        for (int i = 0; i < nfeat; ++i) {
            const int q = qrow[i + 1];
            if (q > 0) {
                if ((rows + q > RB || cnt == LS_FB) && b + 1 < LS_MAXB) { ++b; blkfirst[b] = i; rows = 0; cnt = 0; }
                rows += q; ++cnt;
            }
            acc += q;
            qrow[i + 1] = acc;
        }
        const int nb = (rows > 0) ? b + 1 : b;
        blkfirst[nb] = nfeat;
        s_nblk[0] = nb;
    }
    __syncthreads();
    const int nblk = __builtin_amdgcn_readfirstlane(s_nblk[0]);
    const int nsteps = nblk > 0 ? 1 + ((nblk - 1) / NF) * (w + RGAP) + ((nblk - 1) % NF) + w : 0;
    const int KL = (w - 1) / 8;                                 // last chunk of a fold

    double a[RSLOTS][CS];
    double nxt[PREF ? RSLOTS : 1][PREF ? CS : 1];
    // ---- next block: row tables (two phases around the feature-record load), then the gather ------------
    int pa_M = 0, pa_L = 0;
    long long pa_off = 0;
    unsigned int pa_c[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    auto prep_A = [&](int b) __attribute__((always_inline)) {          // lane r < RB: which feature / local row is block row r; issue the record load
        pa_M = 0;
        if (lane < RB) {
            const int i0 = blkfirst[b], i1 = blkfirst[b + 1];
            const int g = qrow[i0] + lane;
            if (g < qrow[i1]) {
                int lo = i0, hi = i1 - 1;                       // last i with qrow[i] <= g
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (qrow[mid] <= g) lo = mid; else hi = mid - 1; }
                const FeatInfo* fi = p.info + (nd.src_begin + lo);
                pa_L = g - qrow[lo];
                pa_off = fi->blk_off;
                pa_M = fi->M;
                const unsigned int* cw = reinterpret_cast<const unsigned int*>(fi->col);
                pa_c[0] = cw[0]; pa_c[1] = cw[1]; pa_c[2] = cw[2]; pa_c[3] = cw[3];
            }
        }
    };
    auto prep_B = [&]() __attribute__((always_inline)) {               // the tables of the wavefront's next block
        if (lane < RB) {
            rowbase[lane] = (int)(pa_off + (long long)pa_L * (6 * pa_M + 1));
            rowM[lane] = pa_M;
            unsigned int* cw = reinterpret_cast<unsigned int*>(rowcols + lane * 16);
            cw[0] = pa_c[0]; cw[1] = pa_c[1]; cw[2] = pa_c[2]; cw[3] = pa_c[3];
        }
    };
    auto gather_t = [&](auto& dst, auto tagf) __attribute__((always_inline)) {
        constexpr bool F32 = decltype(tagf)::value != 0;
        const double* sd = static_cast<const double*>(p.stack);
        const float* sf = static_cast<const float*>(p.stack);
#pragma unroll
        for (int rr = 0; rr < RSLOTS; ++rr) {
            const int r = rq + 4 * rr;
            const int rb = rowbase[r], Mr = rowM[r];
            const unsigned char* rc = rowcols + r * 16;
#pragma unroll
            for (int k = 0; k < CS; ++k) {
                const int lc = cq + CL * k;
                const bool isr = (k == CS - 1) && (cq == CL - 1);
                const int slot = (lc * 43) >> 8;                 // lc / 6 for lc < 96
                const int aa = lc - 6 * slot;
                const int v = rc[slot];
                const bool ok = (Mr > 0) && (isr || v != 0xFF);
                const int cc = isr ? 6 * Mr : 6 * v + aa;
                const long long idx = ok ? (long long)(rb + cc) : p.zero_idx;
                if constexpr (F32) dst[rr][k] = (double)sf[idx];
                else dst[rr][k] = sd[idx];
            }
            // (keeps the address arithmetic of one row slot from being hoisted over the loads of all the others:
            //  without it the 90-column tile needs ~100 registers of addresses on top of the two blocks)
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto gather = [&](auto& dst) __attribute__((always_inline)) {
        if (p.stack_f32) gather_t(dst, STag<1>{});
        else gather_t(dst, STag<0>{});
    };

    int tcur = 0;
    // The R entries this lane looks after: slot rq (local column lco1) and, on the 90-column tile, slot 4 + (rq & 1)
    // (rows 2, 3 mirror rows 0, 1: same reads, same tau, no write).
    int ra1 = 0, wa1 = 0, st1 = 0, ra2 = 0, wa2 = 0, st2 = 0, sw2 = 0;
    const int lco1 = cq + CL * rq;
    const int lco2 = cq + CL * (4 + (rq & 1));
    const bool isr1 = (CS == 4) && (rq == 3) && (cq == CL - 1);
    const bool isr2 = HAS2 && (4 + (rq & 1) == CS - 1) && (cq == CL - 1);
    auto init_addr = [&]() {
        {
            const bool valid = isr1 || lco1 < w;
            ra1 = valid ? RB0 + (isr1 ? W - 1 : lco1) : zero_i;
            wa1 = valid ? ra1 : dump_i;
            st1 = valid ? (isr1 ? W : W - 1) : 0;
        }
        if constexpr (HAS2) {
            const bool valid = isr2 || lco2 < w;
            ra2 = valid ? RB0 + (isr2 ? W - 1 : lco2) : zero_i;
            wa2 = (valid && rq < 2) ? ra2 : dump_i;
            st2 = valid ? (isr2 ? W : W - 1) : 0;
            sw2 = (valid && rq < 2) ? st2 : 0;
        }
    };

    auto step = [&](auto tagk, auto tagj) {
        constexpr int I = 8 * decltype(tagk)::value + decltype(tagj)::value;   // the block's column
        constexpr int K0 = I / CL, L = I % CL;
        const int rrow = RB0 + I * W;
        int l1 = lco1, l2 = lco2;
        asm volatile("" : "+v"(l1), "+v"(l2));            // (keeps the lane masks of all columns from being hoisted and spilled)
        const bool on1 = (l1 > I) || isr1;
        const bool on2 = HAS2 && ((l2 > I) || isr2);
        sweep_column_step<CS, RSLOTS, RSLOTS, K0, L>(a, smem, rrow, ra1, wa1, on1, ra2, wa2, on2, dump_i, lane);
        ra1 += st1; wa1 += st1;
        if constexpr (HAS2) { ra2 += st2; wa2 += sw2; }
    };

    bool gathered = false;              // the next block sits in nxt
    auto chunk = [&](auto tagk, int bnext) {
        constexpr int KK = decltype(tagk)::value;
        if constexpr (KK < NCH) {
            if (8 * KK >= w) return;
            if (PREF && bnext >= 0) {
                if (KK == KL - 1) prep_A(bnext);
                if constexpr (PREF && KK >= KGMIN) {
                    if (KK == KL) {
                        if (KL == 0) prep_A(bnext);
                        prep_B();
                        gather(nxt);
                        gathered = true;
                    }
                }
            }
            auto one = [&](auto tagj) {
                if (8 * KK + decltype(tagj)::value < w) {
                    step(tagk, tagj);
                    __syncthreads();
                    ++tcur;
                }
            };
            one(STag<0>{}); one(STag<1>{}); one(STag<2>{}); one(STag<3>{});
            one(STag<4>{}); one(STag<5>{}); one(STag<6>{}); one(STag<7>{});
        }
    };

    __syncthreads();
    int b = wv;
    if constexpr (PREF) {
        if (b < nblk) {
            prep_A(b);
            prep_B();
            gather(nxt);
        }
    }
    while (b < nblk) {
        const int t0 = 1 + (b / NF) * (w + RGAP) + (b % NF);
        if constexpr (!PREF) {
            const int tr = (b / NF) * (w + RGAP);           // head of the round: every wavefront has finished its block
            while (tcur < tr) { __syncthreads(); ++tcur; }
            prep_A(b);
            prep_B();
            gather(a);
        }
        while (tcur < t0) { __syncthreads(); ++tcur; }
        init_addr();
        if constexpr (PREF) {
#pragma unroll
            for (int rr = 0; rr < RSLOTS; ++rr)
#pragma unroll
                for (int k = 0; k < CS; ++k) a[rr][k] = nxt[rr][k];
        }
        const int bn = b + NF;
        const int bnext = bn < nblk ? bn : -1;
        gathered = false;
        chunk(STag<0>{}, bnext);
        chunk(STag<1>{}, bnext);
        chunk(STag<2>{}, bnext);
        chunk(STag<3>{}, bnext);
        chunk(STag<4>{}, bnext);
        chunk(STag<5>{}, bnext);
        chunk(STag<6>{}, bnext);
        chunk(STag<7>{}, bnext);
        chunk(STag<8>{}, bnext);
        chunk(STag<9>{}, bnext);
        chunk(STag<10>{}, bnext);
        chunk(STag<11>{}, bnext);
        if constexpr (PREF) {
            if (bnext >= 0 && !gathered) {   // short windows: no chunk instance may hold a second block -- load it now
                prep_A(bnext);
                prep_B();
                gather(nxt);
            }
        }
        b = bn;
    }
    while (tcur < nsteps) { __syncthreads(); ++tcur; }

    // ---- flush R: row-major w x (w+1), entries at and right of the diagonal ----
    __syncthreads();
    for (int c = wv; c < w; c += NF) {
        const double* Rrow = Rb + (size_t)c * W;
        for (int col = c + lane; col < w; col += 64) out[(size_t)c * ldo + col] = Rrow[col - c];
        if (lane == 0) out[(size_t)c * ldo + w] = Rrow[W - 1];
    }
}

}  // namespace msckf
