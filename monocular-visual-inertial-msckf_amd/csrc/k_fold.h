// K5: QR compression of the stacked system (reference MSCKF.py:594-598) as a
// tree of "fold" nodes.  Only the 6N clone columns are factored: the first 15
// columns of every H_o row are identically zero (MSCKF.py:538-540 writes the
// clone block only), so T_H[:, :15] == 0.
//
// A node owns a window of clone slots [win_lo, win_lo + w/6) and produces the
// upper-triangular R (w x (w+1), last column = Q^T r) of all rows below it:
//   leaf : rows are the accepted features' compact blocks [H_o | r_o] (K4),
//          scattered into the window's local columns;
//   merge: rows are the children's R factors (the first child is adopted as
//          the accumulator in place, the others are folded into it).
// One workgroup per node.  The rows to fold are taken in batches of up to
// 16*RPT rows (sorted by leading column) that live in REGISTERS for the whole
// elimination: lane (rq, cq) of wave v owns rows {rq + 16 r} and columns
// {(4 v + cq) + NCG k} of the batch (cyclic in both directions, so work stays
// balanced as columns retire and rows come alive).  Per column j one Householder
// reflector is built from [R_jj ; batch(:, j)]:
//   - the 16 lanes that own column j publish it (v) through a double-buffered
//     LDS vector -> ONE workgroup barrier per column;
//   - every lane reads the v entries of its rows, forms its partial dots with
//     its columns and reduces them over the 16 row-lanes with DPP moves;
//   - R row j is streamed from/to HBM (prefetched one step ahead in registers).
// Householder only: the stack is exactly rank deficient (rank 6N-4) so no
// Gram / Cholesky-QR shortcut is admissible.
#pragma once
#include <hip/hip_runtime.h>
#include "wave_ops.h"

#ifndef FOLD_ABL
#define FOLD_ABL 0      // timing ablations (wrong results): 1 no rsqrt chain, 2 no DPP reduce, 4 no update, 8 no v read, 16 no dot
#endif

namespace msckf {

struct FoldNode {
    int kind;            // 0 = leaf (sources are sorted features), 1 = merge (sources are nodes)
    int src_begin;       // first source index
    int src_end;         // one past the last source
    int win_lo;          // first clone slot of the window
    int w;               // window width in columns (6 * slots)
    int pad;
    long long out_off;   // offset (doubles) of the node's R block in rbuf
};

struct FoldArgs {
    const FoldNode* nodes;
    int node_base;              // nodes [node_base, node_base + gridDim.x) run in this launch
    int lds_doubles;            // dynamic LDS size in doubles
    const int* view_ptr;        // sorted feature order
    const int* obs_slot;
    const int* fmin;            // [F] smallest slot of the feature
    const long long* blk_off;
    const void* stack;          // K4 blocks: row-major q x (6M+1) scalars per accepted feature (k_feature.h)
    int stack_f32;              // 0 = double, 1 = float
    const int* rank;
    const unsigned char* accepted;
    double* rbuf;
    long long* stamps;          // optional diagnostics: per-node cycle stamps (8 per node), may be null
};

// The triangles a CUT merge tree ends with (split long tracks' remainder rows, msckf_abi.hip plan_batch: the last levels of a
// dense tree halve a few hundred rows per ~250 us launch, K6-K7 takes 16 rows in ~7 us) laid down as ONE dense row-major matrix
// [rows_pad][dc + 1] -- window columns at their place, rhs last, zeros elsewhere and in the padding rows -- for K6-K7 to take as
// they are (k_gstream.h, second source).  reference MSCKF.py:594-598 (any orthogonal row compression of the stack serves :604-614)
constexpr int TRI_GATHER_MAX = 64;
struct TriGatherArgs {
    const double* rbuf; double* dst;
    int dc, n, rows, rows_pad;               // n triangles, sum of their widths, ... rounded up to whole blocks of 16
    long long off[TRI_GATHER_MAX];           // FoldNode::out_off
    int w[TRI_GATHER_MAX], col0[TRI_GATHER_MAX], row0[TRI_GATHER_MAX + 1];
};
__global__ __launch_bounds__(256) void k_tri_gather(TriGatherArgs p) {
    const int ld = p.dc + 1;
    const long long total = (long long)p.rows_pad * ld;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / ld), col = (int)(idx - (long long)row * ld);
        double v = 0.0;
        if (row < p.rows) {
            int k = 0;
            while (k + 1 < p.n && row >= p.row0[k + 1]) ++k;
            const int i = row - p.row0[k], w = p.w[k];
            const int j = (col == p.dc) ? w : col - p.col0[k];
            if (j >= 0 && (j < w || col == p.dc)) v = p.rbuf[p.off[k] + (long long)i * (w + 1) + j];
        }
        p.dst[idx] = v;
    }
}

constexpr int FOLD_MAX_SRC = 1024;   // sources per node the LDS bookkeeping can hold
constexpr int STAGE_U = 4;           // loads in flight per lane while a leaf stages one K4 block

// LDS carve-up of k_fold (offsets in doubles); shared by the kernel and the host planner.
struct FoldLayout {
    int ld;        // staging row stride (odd: conflict-free column walks)
    int vbuf;      // [2][bmax + 2] published column j (+ pivot)
    int ints;      // int region: nalive[w], rstate[w], srow0[FOLD_MAX_SRC+1], ctl[8], blead[bmax],
                   //             bsrc[bmax], slotmap[16][32]
    int panel;     // [prow][ld] staging panel
    int prow;      // staging rows that fit
};

__host__ __device__ inline FoldLayout fold_layout(int w, int bmax, int lds_doubles) {
    FoldLayout L;
    L.ld = (w + 1) | 1;
    L.vbuf = 0;
    L.ints = 2 * (bmax + 2);
    const int nints = 2 * w + (FOLD_MAX_SRC + 1) + 8 + 2 * bmax + 16 * 32;
    L.panel = L.ints + (nints + 1) / 2;
    L.prow = (lds_doubles - L.panel) / L.ld;
    return L;
}

// Fallback for windows whose packed R does not fit LDS (w > FOLD_RLDS_MAX_W): R rows are
// streamed from/to HBM every step.  T threads; batch of up to 16*RPT rows; up to CPT*(T/16)
// columns (incl. the rhs column).
template <int T, int RPT, int CPT>
__global__ __launch_bounds__(T) void k_fold_g(FoldArgs p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NCG = T / 16;          // column groups
    constexpr int BMAX = 16 * RPT;       // rows per register batch
    const FoldNode nd = p.nodes[p.node_base + blockIdx.x];
    const int t = threadIdx.x;
    const int rq = t & 15;               // row lane inside the DPP row
    const int cg = t >> 4;               // column group
    const int w = nd.w;
    const int nsrc = nd.src_end - nd.src_begin;
    const FoldLayout lay = fold_layout(w, BMAX, p.lds_doubles);
    const int ld = lay.ld;
    const int prow = lay.prow;
    double* vbuf = smem + lay.vbuf;
    int* nalive = reinterpret_cast<int*>(smem + lay.ints);   // [w]
    int* rstate = nalive + w;                                 // [w] 0 empty, 1 adopted child, 2 out block
    int* srow0 = rstate + w;                                  // [FOLD_MAX_SRC+1] first row of each source
    int* s_ctl = srow0 + (FOLD_MAX_SRC + 1);                  // [8]
    int* blead = s_ctl + 8;                                   // [BMAX] leading column of each batch row
    int* bsrc = blead + BMAX;                                 // [BMAX] merge: (child << 20) | child row
    int* slotmap = bsrc + BMAX;                               // [T/64][32] leaf: per-wave clone slots of a track
    double* panel = smem + lay.panel;                         // [prow][ld]

    double* out = p.rbuf + nd.out_off;
    const int ldo = w + 1;
    long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, tk4 = 0;
    if (p.stamps) tk0 = wall_clock64();

    // adopted child (merge only)
    int ad_off = 0, ad_w = 0;
    const double* ad_blk = nullptr;
    if (nd.kind == 1) {
        const FoldNode c0 = p.nodes[nd.src_begin];
        ad_off = 6 * (c0.win_lo - nd.win_lo);
        ad_w = c0.w;
        ad_blk = p.rbuf + c0.out_off;
    }
    for (int j = t; j < w; j += T) rstate[j] = (nd.kind == 1 && j >= ad_off && j < ad_off + ad_w) ? 1 : 0;

    // ---- per-source row counts (node-global prefix) -------------------------
    // leaf : source i = feature nd.src_begin + i, rows = accepted ? 2M - rank : 0
    // merge: source i = child node nd.src_begin + 1 + i (child 0 is adopted), rows = child w
    const int nfold = (nd.kind == 0) ? nsrc : nsrc - 1;
    for (int i = t; i < nfold; i += T) {
        int rows;
        if (nd.kind == 0) {
            const int f = nd.src_begin + i;
            rows = (p.accepted[f] == 1) ? 2 * (p.view_ptr[f + 1] - p.view_ptr[f]) - p.rank[f] : 0;
        } else {
            rows = p.nodes[nd.src_begin + 1 + i].w;
        }
        srow0[i + 1] = rows;
    }
    __syncthreads();
    if (t == 0) {
        int acc = 0;
        srow0[0] = 0;
        for (int i = 0; i < nfold; ++i) { acc += srow0[i + 1]; srow0[i + 1] = acc; }
        s_ctl[3] = acc;
    }
    __syncthreads();
    const int total_rows = s_ctl[3];

    // R row element (j, c) from wherever row j currently lives
    auto load_r = [&](int j, int c) -> double {
        const int st = rstate[j];
        if (st == 2) return out[(size_t)j * ldo + c];
        if (st == 1) {
            const int i = j - ad_off;
            if (c == w) return ad_blk[(size_t)i * (ad_w + 1) + ad_w];
            if (c < ad_off + ad_w) return ad_blk[(size_t)i * (ad_w + 1) + (c - ad_off)];
            return 0.0;
        }
        return 0.0;
    };

    if (p.stamps) tk1 = wall_clock64();
    for (int row_lo = 0; row_lo < total_rows; row_lo += BMAX) {
        const int nb = min(BMAX, total_rows - row_lo);
        long long ts0 = 0;
        if (p.stamps) ts0 = wall_clock64();
        double a[RPT][CPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r)
#pragma unroll
            for (int k = 0; k < CPT; ++k) a[r][k] = 0.0;

        if (nd.kind == 0) {
            // ---------- leaf: stage the compact K4 blocks through the LDS panel ------
            for (int sub_lo = row_lo; sub_lo < row_lo + nb; sub_lo += prow) {
                const int np = min(prow, row_lo + nb - sub_lo);
                __syncthreads();
                for (int e = t; e < np * ld; e += T) panel[e] = 0.0;
                __syncthreads();
                // one wavefront per source block (column-major (6M+1) x 2M)
                const int wv = t >> 6, ln = t & 63;
                int* smap = slotmap + wv * 32;
                for (int i = wv; i < nfold; i += T / 64) {
                    const int r0 = srow0[i], r1 = srow0[i + 1];
                    if (r1 <= sub_lo || r0 >= sub_lo + np || r1 == r0) continue;
                    const int f = nd.src_begin + i;
                    const int vbeg = p.view_ptr[f];
                    const int M = p.view_ptr[f + 1] - vbeg;
                    const int ldb = 6 * M + 1;                                  // block: (r1 - r0) rows x ldb, row-major
                    const float inv_ldb = 1.0f / (float)ldb;
                    const double* blk = static_cast<const double*>(p.stack) + p.blk_off[f];
                    const float* blkf = static_cast<const float*>(p.stack) + p.blk_off[f];
                    const int lead = 6 * (p.fmin[f] - nd.win_lo);
                    if (ln < M) smap[ln] = 6 * (p.obs_slot[vbeg + ln] - nd.win_lo);
                    const int nel = ldb * (r1 - r0);
                    // (all loads of a trip issued before the first use: the staging is bound by the HBM round trip)
                    for (int e0 = 0; e0 < nel; e0 += 64 * STAGE_U) {
                        double x[STAGE_U];
#pragma unroll
                        for (int u = 0; u < STAGE_U; ++u) {
                            const int e = e0 + 64 * u + ln;
                            x[u] = (e < nel) ? (p.stack_f32 ? (double)blkf[e] : blk[e]) : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < STAGE_U; ++u) {
                            const int e = e0 + 64 * u + ln;
                            if (e < nel) {
                                const int L = (int)(((float)e + 0.5f) * inv_ldb), c = e - L * ldb;   // e / ldb, exact for e < 2^20
                                const int gr = r0 + L;                  // node-global sorted row
                                if (gr >= sub_lo && gr < sub_lo + np) {
                                    const int col = (c == 6 * M) ? w : smap[c / 6] + (c % 6);
                                    panel[(gr - sub_lo) * ld + col] = x[u];
                                    if (c == 0) blead[gr - row_lo] = lead;
                                }
                            }
                        }
                    }
                }
                __syncthreads();
                // registers pick up their rows / columns of this panel
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    const int gr = row_lo + rq + 16 * r;
                    if (gr >= sub_lo && gr < sub_lo + np) {
#pragma unroll
                        for (int k = 0; k < CPT; ++k) {
                            const int c = cg + NCG * k;
                            if (c <= w) a[r][k] = panel[(gr - sub_lo) * ld + c];
                        }
                    }
                }
            }
        } else {
            // ---------- merge: batch rows come straight from the children's R blocks ---
            // sorted position of (child i, row ri) = ri + rows of the other children that sort first
            __syncthreads();
            for (int i = 0; i < nfold; ++i) {
                const FoldNode ch = p.nodes[nd.src_begin + 1 + i];
                const int off = 6 * (ch.win_lo - nd.win_lo);
                for (int ri = t; ri < ch.w; ri += T) {
                    const int lead = off + ri;
                    int pos = ri;
                    for (int k = 0; k < nfold; ++k) {
                        if (k == i) continue;
                        const FoldNode ok = p.nodes[nd.src_begin + 1 + k];
                        const int offk = 6 * (ok.win_lo - nd.win_lo);
                        int cnt = lead - offk + (k < i ? 1 : 0);
                        cnt = max(0, min(cnt, ok.w));
                        pos += cnt;
                    }
                    if (pos >= row_lo && pos < row_lo + nb) {
                        bsrc[pos - row_lo] = (i << 20) | ri;
                        blead[pos - row_lo] = lead;
                    }
                }
            }
            __syncthreads();
            // every lane loads its own elements (independent loads; entries below a child's
            // diagonal were never written and read back as the workspace's zeros)
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const int b = rq + 16 * r;
                if (b < nb) {
                    const int code = bsrc[b];
                    const FoldNode ch = p.nodes[nd.src_begin + 1 + (code >> 20)];
                    const int ri = code & 0xFFFFF;
                    const int off = 6 * (ch.win_lo - nd.win_lo);
                    const double* rowp = p.rbuf + ch.out_off + (size_t)ri * (ch.w + 1);
#pragma unroll
                    for (int k = 0; k < CPT; ++k) {
                        const int c = cg + NCG * k;
                        double x = 0.0;
                        if (c == w) x = rowp[ch.w];
                        else if (c >= off + ri && c < off + ch.w) x = rowp[c - off];
                        a[r][k] = x;
                    }
                }
            }
        }
        __syncthreads();
        // live-row count per column: rows are sorted by lead -> binary search
        for (int j = t; j < w; j += T) {
            int lo = 0, hi = nb;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (blead[mid] <= j) lo = mid + 1; else hi = mid; }
            nalive[j] = lo;
        }
        __syncthreads();
        const int jmin = blead[0];
        if (p.stamps) { const long long tn = wall_clock64(); tk2 += tn - ts0; ts0 = tn; }

        // R row jmin for this lane's columns (later rows are prefetched one step ahead)
        double rcur[CPT], rnxt[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int c = cg + NCG * k;
            rcur[k] = (c >= jmin && c <= w) ? load_r(jmin, c) : 0.0;
            rnxt[k] = 0.0;
        }

        for (int j = jmin; j < w; ++j) {
            const int na = nalive[j];
            const int ra = (na + 15) >> 4;                 // live register rows (uniform bound)
            // prefetch R row j+1
            if (j + 1 < w) {
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const int c = cg + NCG * k;
                    rnxt[k] = (c >= j + 1 && c <= w) ? load_r(j + 1, c) : 0.0;
                }
            }
            // the 16 lanes that own column j publish it, plus the pivot R_jj
            double* vb = vbuf + (j & 1) * (BMAX + 2);
            const int kj = j / NCG;
            if (cg == j - kj * NCG) {
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    if (k == kj) {
#pragma unroll
                        for (int r = 0; r < RPT; ++r)
                            if (r < ra) vb[rq + 16 * r] = a[r][k];
                        if (rq == 0) vb[BMAX] = rcur[k];
                    }
                }
            }
            __syncthreads();
            double v[RPT];
            double sg = 0.0;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                v[r] = (r < ra) ? vb[rq + 16 * r] : 0.0;
                sg += v[r] * v[r];
            }
            const double x0 = vb[BMAX];
            sg = row16_sum(sg);
            if (sg > 1e-290) {
                const double nrm = sqrt(x0 * x0 + sg);
                const double alpha = (x0 > 0.0) ? -nrm : nrm;
                const double v0 = x0 - alpha;
                const double beta = 1.0 / (nrm * (nrm + fabs(x0)));
                double dot[CPT];
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    double s = 0.0;
#pragma unroll
                    for (int r = 0; r < RPT; ++r) s += v[r] * a[r][k];
                    dot[k] = s;
                }
#pragma unroll
                for (int k = 0; k < CPT; ++k) dot[k] = row16_sum(dot[k]);
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const int c = cg + NCG * k;
                    if (c > j && c <= w) {
                        const double tau = beta * (dot[k] + v0 * rcur[k]);
                        if (rq == 0) out[(size_t)j * ldo + c] = rcur[k] - tau * v0;
#pragma unroll
                        for (int r = 0; r < RPT; ++r) a[r][k] -= tau * v[r];
                    } else if (c == j) {
                        if (rq == 0) out[(size_t)j * ldo + j] = alpha;
                    }
                }
            } else {
                // nothing to eliminate in this column: R row j passes through
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const int c = cg + NCG * k;
                    if (c >= j && c <= w && rq == 0) out[(size_t)j * ldo + c] = rcur[k];
                }
            }
            if (t == 0) rstate[j] = 2;
#pragma unroll
            for (int k = 0; k < CPT; ++k) rcur[k] = rnxt[k];
        }
        __syncthreads();
        if (p.stamps) { const long long tn = wall_clock64(); tk3 += tn - ts0; }
    }

    // ---- rows never touched by a fold step: copy (adopted) or zero-fill ------
    __syncthreads();
    for (int j = 0; j < w; ++j) {
        const int st = rstate[j];
        if (st == 2) continue;
        for (int c = j + t; c <= w; c += T) {
            double x = 0.0;
            if (st == 1) {
                const int i = j - ad_off;
                if (c == w) x = ad_blk[(size_t)i * (ad_w + 1) + ad_w];
                else if (c < ad_off + ad_w) x = ad_blk[(size_t)i * (ad_w + 1) + (c - ad_off)];
            }
            out[(size_t)j * ldo + c] = x;
        }
    }
    if (p.stamps && t == 0) {
        tk4 = wall_clock64();
        long long* o = p.stamps + 8 * (p.node_base + blockIdx.x);
        o[0] = tk1 - tk0; o[1] = tk2; o[2] = tk3; o[3] = tk4 - tk0; o[4] = w; o[5] = total_rows;
    }
}


// ---------------------------------------------------------------------------------
// Main variant: the node's R accumulator (packed upper triangle incl. the rhs column,
// (w+1)(w+2)/2 doubles) stays in LDS for the whole node; the step loop touches no global
// memory.  Row j of R sits at roff(j) = j (w+1) - j (j-1)/2 and holds columns j..w.
constexpr int FOLD_RLDS_MAX_W = 186;

struct FoldLayoutR {
    int ld, vbuf, ints, racc, panel, prow;
};
__host__ __device__ inline FoldLayoutR fold_layout_r(int w, int bmax, int lds_doubles) {
    FoldLayoutR L;
    L.ld = (w + 1) | 1;
    L.vbuf = 0;
    L.ints = 2 * (bmax + 4);            // double-buffered published column
    const int nints = (FOLD_MAX_SRC + 1) + 8 + 2 * bmax + 16 * 32;
    L.racc = L.ints + (nints + 1) / 2;
    L.panel = L.racc + (w + 1) * (w + 2) / 2;
    L.prow = (lds_doubles - L.panel) / L.ld;
    return L;
}

// T threads, RL (8 or 16) row lanes, RPT (multiple of 4) register rows, CPT register columns;
// TRI: rows come alive progressively (merge of triangles) -> skip dead row blocks.
template <int KK> struct KTag { static constexpr int value = KK; };

template <int T, int RL, int RPT, int CPT>
__global__ __launch_bounds__(T) void k_fold(FoldArgs p) {
    static_assert(RPT % 4 == 0, "RPT must be a multiple of 4");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NCG = T / RL;          // column groups
    constexpr int BMAX = RL * RPT;       // rows per register batch
    const FoldNode nd = p.nodes[p.node_base + blockIdx.x];
    const int t = threadIdx.x;
    const int rq = t & (RL - 1);         // row lane inside the half DPP row
    const int cg = t / RL;               // column group
    const int w = nd.w;
    const int nsrc = nd.src_end - nd.src_begin;
    const FoldLayoutR lay = fold_layout_r(w, BMAX, p.lds_doubles);
    const int ld = lay.ld;
    const int prow = lay.prow;
    double* vbuf = smem + lay.vbuf;
    int* srow0 = reinterpret_cast<int*>(smem + lay.ints);    // [FOLD_MAX_SRC+1] first row of each source
    int* s_ctl = srow0 + (FOLD_MAX_SRC + 1);                  // [8]
    int* blead = s_ctl + 8;                                   // [BMAX] leading column of each batch row
    int* bsrc = blead + BMAX;                                 // [BMAX] merge: (child << 20) | child row
    int* slotmap = bsrc + BMAX;                               // [T/64][32] leaf: per-wave clone slots of a track
    double* racc = smem + lay.racc;                           // packed R accumulator
    double* panel = smem + lay.panel;                         // [prow][ld] leaf staging panel
    const int nracc = (w + 1) * (w + 2) / 2;
    const int ldo = w + 1;
    double* out = p.rbuf + nd.out_off;
    long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0;
    if (p.stamps) tk0 = wall_clock64();

    // ---- R accumulator: zero, then adopt the first child (merge) --------------
    for (int e = t; e < nracc; e += T) racc[e] = 0.0;
    const int nfold = (nd.kind == 0) ? nsrc : nsrc - 1;
    for (int i = t; i < nfold; i += T) {
        int rows;
        if (nd.kind == 0) {
            const int f = nd.src_begin + i;
            rows = (p.accepted[f] == 1) ? 2 * (p.view_ptr[f + 1] - p.view_ptr[f]) - p.rank[f] : 0;
        } else {
            rows = p.nodes[nd.src_begin + 1 + i].w;
        }
        srow0[i + 1] = rows;
    }
    __syncthreads();
    if (nd.kind == 1) {
        // adopt the first child's R: one wavefront per row, lanes over the columns (coalesced, no div)
        const FoldNode c0 = p.nodes[nd.src_begin];
        const int off = 6 * (c0.win_lo - nd.win_lo);
        const int cw = c0.w, cld = c0.w + 1;
        const double* blk = p.rbuf + c0.out_off;
        for (int ri = t >> 6; ri < cw; ri += T / 64) {
            const int j = off + ri;
            double* dst = racc + j * (w + 1) - (j * (j - 1)) / 2;        // row j, entry (c - j)
            const double* srcrow = blk + (size_t)ri * cld;
            for (int cc = ri + (t & 63); cc <= cw; cc += 64) {
                const int c = (cc == cw) ? w : off + cc;
                dst[c - j] = srcrow[cc];
            }
        }
    }
    if (t == 0) {
        int acc = 0;
        srow0[0] = 0;
        for (int i = 0; i < nfold; ++i) { acc += srow0[i + 1]; srow0[i + 1] = acc; }
        s_ctl[3] = acc;
    }
    __syncthreads();
    const int total_rows = s_ctl[3];
    if (p.stamps) tk1 = wall_clock64();

    for (int row_lo = 0; row_lo < total_rows; row_lo += BMAX) {
        const int nb = min(BMAX, total_rows - row_lo);
        long long ts0 = 0;
        if (p.stamps) ts0 = wall_clock64();
        double a[RPT][CPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r)
#pragma unroll
            for (int k = 0; k < CPT; ++k) a[r][k] = 0.0;

        if (nd.kind == 0) {
            // ---------- leaf: stage the compact K4 blocks through the LDS panel ------
            for (int sub_lo = row_lo; sub_lo < row_lo + nb; sub_lo += prow) {
                const int np = min(prow, row_lo + nb - sub_lo);
                __syncthreads();
                for (int e = t; e < np * ld; e += T) panel[e] = 0.0;
                __syncthreads();
                const int wv = t >> 6, ln = t & 63;
                int* smap = slotmap + wv * 32;
                for (int i = wv; i < nfold; i += T / 64) {
                    const int r0 = srow0[i], r1 = srow0[i + 1];
                    if (r1 <= sub_lo || r0 >= sub_lo + np || r1 == r0) continue;
                    const int f = nd.src_begin + i;
                    const int vbeg = p.view_ptr[f];
                    const int M = p.view_ptr[f + 1] - vbeg;
                    const int ldb = 6 * M + 1;                                  // block: (r1 - r0) rows x ldb, row-major
                    const float inv_ldb = 1.0f / (float)ldb;
                    const double* blk = static_cast<const double*>(p.stack) + p.blk_off[f];
                    const float* blkf = static_cast<const float*>(p.stack) + p.blk_off[f];
                    const int lead = 6 * (p.fmin[f] - nd.win_lo);
                    if (ln < M) smap[ln] = 6 * (p.obs_slot[vbeg + ln] - nd.win_lo);
                    const int nel = ldb * (r1 - r0);
                    // (all loads of a trip issued before the first use: the staging is bound by the HBM round trip)
                    for (int e0 = 0; e0 < nel; e0 += 64 * STAGE_U) {
                        double x[STAGE_U];
#pragma unroll
                        for (int u = 0; u < STAGE_U; ++u) {
                            const int e = e0 + 64 * u + ln;
                            x[u] = (e < nel) ? (p.stack_f32 ? (double)blkf[e] : blk[e]) : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < STAGE_U; ++u) {
                            const int e = e0 + 64 * u + ln;
                            if (e < nel) {
                                const int L = (int)(((float)e + 0.5f) * inv_ldb), c = e - L * ldb;   // e / ldb, exact for e < 2^20
                                const int gr = r0 + L;                  // node-global sorted row
                                if (gr >= sub_lo && gr < sub_lo + np) {
                                    const int col = (c == 6 * M) ? w : smap[c / 6] + (c % 6);
                                    panel[(gr - sub_lo) * ld + col] = x[u];
                                    if (c == 0) blead[gr - row_lo] = lead;
                                }
                            }
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    const int gr = row_lo + rq + RL * r;
                    if (gr >= sub_lo && gr < sub_lo + np) {
#pragma unroll
                        for (int k = 0; k < CPT; ++k) {
                            const int c = cg + NCG * k;
                            if (c <= w) a[r][k] = panel[(gr - sub_lo) * ld + c];
                        }
                    }
                }
            }
        } else {
            // ---------- merge: batch rows come straight from the children's R blocks ---
            __syncthreads();
            for (int i = 0; i < nfold; ++i) {
                const FoldNode ch = p.nodes[nd.src_begin + 1 + i];
                const int off = 6 * (ch.win_lo - nd.win_lo);
                for (int ri = t; ri < ch.w; ri += T) {
                    const int lead = off + ri;
                    int pos = ri;
                    for (int k = 0; k < nfold; ++k) {
                        if (k == i) continue;
                        const FoldNode ok = p.nodes[nd.src_begin + 1 + k];
                        const int offk = 6 * (ok.win_lo - nd.win_lo);
                        int cnt = lead - offk + (k < i ? 1 : 0);
                        cnt = max(0, min(cnt, ok.w));
                        pos += cnt;
                    }
                    if (pos >= row_lo && pos < row_lo + nb) {
                        bsrc[pos - row_lo] = (i << 20) | ri;
                        blead[pos - row_lo] = lead;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const int b = rq + RL * r;
                if (b < nb) {
                    const int code = bsrc[b];
                    const FoldNode ch = p.nodes[nd.src_begin + 1 + (code >> 20)];
                    const int ri = code & 0xFFFFF;
                    const int off = 6 * (ch.win_lo - nd.win_lo);
                    const double* rowp = p.rbuf + ch.out_off + (size_t)ri * (ch.w + 1);
#pragma unroll
                    for (int k = 0; k < CPT; ++k) {
                        const int c = cg + NCG * k;
                        double x = 0.0;
                        if (c == w) x = rowp[ch.w];
                        else if (c >= off + ri && c < off + ch.w) x = rowp[c - off];
                        a[r][k] = x;
                    }
                }
            }
        }
        __syncthreads();
        __syncthreads();
        const int jmin = blead[0];
        if (p.stamps) { const long long tn = wall_clock64(); tk2 += tn - ts0; ts0 = tn; }

        // ---------- elimination: one reflector per column, no global memory ---------
        // Reflector j (its rows of v, v0, beta) is formed by the RL lanes that own column j and
        // published through a double-buffered LDS vector; ONE workgroup barrier per column.
        // Every step is laid out as  [all LDS reads] -> [math in registers] -> [all LDS writes]:
        // the R row, the published vector and the accumulator share one address space, so any
        // read placed after a write costs a full lgkmcnt drain (ablation: an "empty" step with the
        // reads and writes interleaved per column slot still took 0.7 us of the 1.0 us per column).
        int roff = jmin * (w + 1) - (jmin * (jmin - 1)) / 2;          // start of R row j in racc
        const int cgw = (t >> 6) * (64 / RL);                         // first column group of this wave
        constexpr int VB = BMAX + 4;
        // The pivot column j lives in register slot j / NCG, which changes only every NCG columns:
        // the column loop is cut into CPT chunks with the slot index a compile-time constant, so
        // publishing / retiring columns needs no per-slot selects or branches (an "empty" step was
        // 0.7 us of uniform-branch ladders before this split).
        auto publish = [&](auto tagk, int jn, double* vbn, int roff_n) {
            constexpr int KN = decltype(tagk)::value;
            if constexpr (KN < CPT) {
                double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
#pragma unroll
                for (int g = 0; g < RPT / 4; ++g) {
                    q0 = fma(a[4 * g + 0][KN], a[4 * g + 0][KN], q0);
                    q1 = fma(a[4 * g + 1][KN], a[4 * g + 1][KN], q1);
                    q2 = fma(a[4 * g + 2][KN], a[4 * g + 2][KN], q2);
                    q3 = fma(a[4 * g + 3][KN], a[4 * g + 3][KN], q3);
#pragma unroll
                    for (int u = 0; u < 4; ++u) vbn[rq + RL * (4 * g + u)] = a[4 * g + u][KN];
                }
                const double sgn = rowN_sum<RL>((q0 + q1) + (q2 + q3));
                if (rq == 0) {
                    vbn[BMAX] = sgn;                                   // |column jn|^2 over the batch rows
                    vbn[BMAX + 1] = racc[roff_n];                      // pivot R[jn][jn] (static until step jn;
                }                                                      //  thread 0 overwrites it with alpha then)
            }
        };
        auto run_chunk = [&](auto tagk) {
            constexpr int KK = decltype(tagk)::value;
            const int jlo = max(jmin, KK * NCG), jhi = min(w, (KK + 1) * NCG);
            for (int j = jlo; j < jhi; ++j) {
                const int roff1 = roff + (w + 1) - j;                 // start of R row j+1
                const double* vb = vbuf + (j & 1) * VB;
                double* vbn = vbuf + ((j + 1) & 1) * VB;
                const int jl = j - KK * NCG;                          // pivot's column group
                // ---- read phase -----------------------------------------------------------
                double v[RPT];
#pragma unroll
                for (int r = 0; r < RPT; ++r) v[r] = vb[rq + RL * r];
                const double sg = vb[BMAX];                           // |column j|^2 over the batch rows
                const double x0 = vb[BMAX + 1];                       // pivot R_jj (published: reading racc here
                                                                      // would race with thread 0 writing alpha)
                double rck[CPT];
                bool onk[CPT];
#pragma unroll
                for (int k = KK; k < CPT; ++k) {
                    const int c = cg + NCG * k;
                    onk[k] = (k == KK) ? ((cg > jl) && (c <= w)) : (c <= w);
                    rck[k] = onk[k] ? racc[roff + (c - j)] : 0.0;
                }
                // ---- math -------------------------------------------------------------------
                double alpha = x0;
                const bool live = sg > 1e-290;                        // uniform: same sigma everywhere (below: nrm^2 underflows)
                if (live) {
                    const double ss = fma(x0, x0, sg);
                    double nrm, beta;
                    if (ss > 1e-200 && ss < 1e200) {
                        const double y = fast_rsqrt(ss);
                        nrm = ss * y;                                  // a few 1e-16 relative: the reflector stays orthogonal to that level
                        beta = y * fast_rcp(nrm + fabs(x0));
                    } else {
                        nrm = sqrt(ss);
                        beta = 1.0 / (nrm * (nrm + fabs(x0)));
                    }
                    alpha = (x0 > 0.0) ? -nrm : nrm;
                    const double v0 = x0 - alpha;
#pragma unroll
                    for (int k = KK; k < CPT; ++k) {
                        // slot KK: skip when all column groups of this wavefront are retired;
                        // later slots: skip when they lie beyond the window
                        const bool any = (k == KK) ? (cgw + (64 / RL - 1) > jl) : true;
                        if (any && cgw + NCG * k <= w) {
                            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                            for (int g = 0; g < RPT / 4; ++g) {
                                s0 = fma(v[4 * g + 0], a[4 * g + 0][k], s0);
                                s1 = fma(v[4 * g + 1], a[4 * g + 1][k], s1);
                                s2 = fma(v[4 * g + 2], a[4 * g + 2][k], s2);
                                s3 = fma(v[4 * g + 3], a[4 * g + 3][k], s3);
                            }
                            const double sd = rowN_sum<RL>((s0 + s1) + (s2 + s3));
                            const double tau = onk[k] ? beta * fma(v0, rck[k], sd) : 0.0;
                            rck[k] = fma(-tau, v0, rck[k]);
#pragma unroll
                            for (int r = 0; r < RPT; ++r) a[r][k] = fma(-tau, v[r], a[r][k]);
                        }
                    }
                }
                // ---- write phase ----------------------------------------------------------
                if (live) {
#pragma unroll
                    for (int k = KK; k < CPT; ++k) {
                        const int c = cg + NCG * k;
                        if (onk[k] && rq == 0) racc[roff + (c - j)] = rck[k];
                    }
                    if (t == 0) racc[roff] = alpha;
                }
                // the owners of the next pivot column publish it with its squared norm
                const int jn = j + 1;
                if (jn < w) {
                    if (jl + 1 < NCG) { if (cg == jl + 1) publish(KTag<KK>{}, jn, vbn, roff1); }
                    else { if (cg == 0) publish(KTag<KK + 1>{}, jn, vbn, roff1); }
                }
                roff = roff1;
                __syncthreads();
            }
        };
        // prologue: the owners of column jmin publish it
        {
            const int kn = jmin / NCG;
            double* vbn = vbuf + (jmin & 1) * VB;
            if (cg == jmin - kn * NCG) {
                if (kn == 0) publish(KTag<0>{}, jmin, vbn, roff);
                if constexpr (CPT > 1) { if (kn == 1) publish(KTag<1>{}, jmin, vbn, roff); }
                if constexpr (CPT > 2) { if (kn == 2) publish(KTag<2>{}, jmin, vbn, roff); }
                if constexpr (CPT > 3) { if (kn == 3) publish(KTag<3>{}, jmin, vbn, roff); }
                if constexpr (CPT > 4) { if (kn == 4) publish(KTag<4>{}, jmin, vbn, roff); }
                if constexpr (CPT > 5) { if (kn == 5) publish(KTag<5>{}, jmin, vbn, roff); }
                if constexpr (CPT > 6) { if (kn == 6) publish(KTag<6>{}, jmin, vbn, roff); }
                if constexpr (CPT > 7) { if (kn == 7) publish(KTag<7>{}, jmin, vbn, roff); }
            }
        }
        __syncthreads();
        run_chunk(KTag<0>{});
        if constexpr (CPT > 1) run_chunk(KTag<1>{});
        if constexpr (CPT > 2) run_chunk(KTag<2>{});
        if constexpr (CPT > 3) run_chunk(KTag<3>{});
        if constexpr (CPT > 4) run_chunk(KTag<4>{});
        if constexpr (CPT > 5) run_chunk(KTag<5>{});
        if constexpr (CPT > 6) run_chunk(KTag<6>{});
        if constexpr (CPT > 7) run_chunk(KTag<7>{});
        static_assert(CPT <= 8, "add more chunks");
        if (p.stamps) { const long long tn = wall_clock64(); tk3 += tn - ts0; }
    }

    // ---- flush R to the node's block (row-major w x (w+1), entries at and right of the diagonal)
    __syncthreads();
    for (int j = t >> 6; j < w; j += T / 64) {
        const int ro = j * (w + 1) - (j * (j - 1)) / 2;
        for (int c = j + (t & 63); c <= w; c += 64) out[(size_t)j * ldo + c] = racc[ro + (c - j)];
    }
    if (p.stamps && t == 0) {
        const long long tk4 = wall_clock64();
        long long* o = p.stamps + 8 * (p.node_base + blockIdx.x);
        o[0] = tk1 - tk0; o[1] = tk2; o[2] = tk3; o[3] = tk4 - tk0; o[4] = w; o[5] = total_rows;
    }
}

}  // namespace msckf
