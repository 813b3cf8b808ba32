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
// One workgroup per node.  Rows are staged in LDS in batches sorted by their
// leading column; per column j one Householder reflector built from
// [R_jj ; batch(:, j)] is applied to R row j (streamed from/to HBM, prefetched
// one step ahead) and to the live batch rows (rows whose leading column <= j).
// Householder only: the stack is exactly rank deficient (rank 6N-4) so no
// Gram / Cholesky-QR shortcut is admissible.
#pragma once
#include <hip/hip_runtime.h>
#include "wave_ops.h"

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
    const double* stack;
    const int* rank;
    const unsigned char* accepted;
    double* rbuf;
};

constexpr int FOLD_MAX_SRC = 1024;   // sources per node the LDS bookkeeping can hold

// LDS carve-up of k_fold (offsets in doubles); shared by the kernel and the host planner.
struct FoldLayout {
    int ld;        // batch row stride (odd: conflict-free column walks)
    int rrow;      // [2][w+2] current / next R row
    int part;      // [T] partial dots
    int ints;      // int region: nalive[w], rstate[w], srow0[FOLD_MAX_SRC+1], ctl[8]
    int blead;     // [bcap] ints: leading column of each batch row
    int batch;     // [bcap][ld]
    int bcap;      // batch rows that fit
};

__host__ __device__ inline FoldLayout fold_layout(int w, int T, int lds_doubles) {
    FoldLayout L;
    L.ld = (w + 1) | 1;
    L.rrow = 0;
    L.part = 2 * (w + 2);
    L.ints = L.part + T;
    const int nints = 2 * w + (FOLD_MAX_SRC + 1) + 8;
    const int dyn = L.ints + (nints + 1) / 2;
    const int avail = lds_doubles - dyn;
    L.bcap = (2 * avail) / (2 * L.ld + 1) - 1;
    L.blead = dyn;
    L.batch = dyn + (L.bcap + 1) / 2;
    return L;
}

template <int T>
__global__ __launch_bounds__(T) void k_fold(FoldArgs p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const FoldNode nd = p.nodes[p.node_base + blockIdx.x];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = nd.w;
    const int nsrc = nd.src_end - nd.src_begin;
    const FoldLayout lay = fold_layout(w, T, p.lds_doubles);
    const int ld = lay.ld;
    const int Bcap = lay.bcap;
    double* rrow = smem + lay.rrow;
    double* part = smem + lay.part;
    int* nalive = reinterpret_cast<int*>(smem + lay.ints);   // [w]
    int* rstate = nalive + w;                                 // [w] 0 empty, 1 adopted child, 2 out block
    int* srow0 = rstate + w;                                  // [FOLD_MAX_SRC+1] first row of each source
    int* s_ctl = srow0 + (FOLD_MAX_SRC + 1);                  // [8]
    int* blead = reinterpret_cast<int*>(smem + lay.blead);    // [Bcap]
    double* batch = smem + lay.batch;                         // [Bcap][ld]

    double* out = p.rbuf + nd.out_off;
    const int ldo = w + 1;

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
    if (t == 0) {
        int acc = 0;
        for (int i = 0; i < nfold; ++i) {
            srow0[i] = acc;
            if (nd.kind == 0) {
                const int f = nd.src_begin + i;
                if (p.accepted[f]) acc += 2 * (p.view_ptr[f + 1] - p.view_ptr[f]) - p.rank[f];
            } else {
                acc += p.nodes[nd.src_begin + 1 + i].w;
            }
        }
        srow0[nfold] = acc;
        s_ctl[3] = acc;
    }
    __syncthreads();
    const int total_rows = s_ctl[3];

    // Row r of the node (0 <= r < total_rows) in lead-sorted order:
    //   leaf : sources are already sorted by lead, rows of a source share its lead
    //   merge: global sort position of (child c, row i) is computed in closed form
    for (int row_lo = 0; row_lo < total_rows; row_lo += Bcap) {
        const int nb = min(Bcap, total_rows - row_lo);
        // ---------- stage the batch: zero, then scatter ----------------------
        for (int e = t; e < nb * ld; e += T) batch[e] = 0.0;
        __syncthreads();
        if (nd.kind == 0) {
            // threads over (source, element): each source block is column-major (6M+1) x 2M
            for (int i = 0; i < nfold; ++i) {
                const int r0 = srow0[i], r1 = srow0[i + 1];
                if (r1 <= row_lo || r0 >= row_lo + nb || r1 == r0) continue;
                const int f = nd.src_begin + i;
                const int vbeg = p.view_ptr[f];
                const int M = p.view_ptr[f + 1] - vbeg;
                const int R2 = 2 * M, rk = p.rank[f];
                const double* blk = p.stack + p.blk_off[f];
                const int lead = 6 * (p.fmin[f] - nd.win_lo);
                const int nel = (6 * M + 1) * R2;
                for (int e = t; e < nel; e += T) {
                    const int c = e / R2, L = e - c * R2;
                    if (L < rk) continue;
                    const int br = r0 + (L - rk) - row_lo;
                    if (br < 0 || br >= nb) continue;
                    const int col = (c == 6 * M) ? w : 6 * (p.obs_slot[vbeg + c / 6] - nd.win_lo) + (c % 6);
                    batch[br * ld + col] = blk[e];
                    if (c == 0) blead[br] = lead;
                }
            }
        } else {
            for (int i = 0; i < nfold; ++i) {
                const FoldNode ch = p.nodes[nd.src_begin + 1 + i];
                const int off = 6 * (ch.win_lo - nd.win_lo);
                const double* blk = p.rbuf + ch.out_off;
                const int cw = ch.w, cld = ch.w + 1;
                // sorted position of (child i, row ri): rows of other children with smaller lead
                for (int e = t; e < cw * cld; e += T) {
                    const int ri = e / cld, cc = e - ri * cld;
                    if (cc < ri) continue;                       // below the diagonal: never written
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
                    const int br = pos - row_lo;
                    if (br < 0 || br >= nb) continue;
                    const int col = (cc == cw) ? w : off + cc;
                    batch[br * ld + col] = blk[ri * cld + cc];
                    if (cc == ri) blead[br] = lead;
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

        // ---------- fetch R row jmin synchronously -----------------------------
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
        for (int c = jmin + t; c <= w; c += T) rrow[(jmin & 1) * (w + 2) + c] = load_r(jmin, c);
        __syncthreads();

        for (int j = jmin; j < w; ++j) {
            const int na = nalive[j];
            double* cur = rrow + (j & 1) * (w + 2);
            double* nxtbuf = rrow + ((j + 1) & 1) * (w + 2);
            // prefetch R row j+1 (registers now, LDS at the end of the step)
            double nx0 = 0.0, nx1 = 0.0;
            const int pc0 = j + 1 + t, pc1 = j + 1 + t + T;
            if (j + 1 < w) {
                if (pc0 <= w) nx0 = load_r(j + 1, pc0);
                if (pc1 <= w) nx1 = load_r(j + 1, pc1);
            }
            // sigma = sum of squares of column j over the live rows (every wave redundantly)
            double sg = 0.0;
            for (int b = lane; b < na; b += 64) { const double x = batch[b * ld + j]; sg += x * x; }
            sg = wave_sum(sg);
            const double x0 = cur[j];
            if (sg > 0.0) {
                const double nrm = sqrt(x0 * x0 + sg);
                const double alpha = (x0 > 0.0) ? -nrm : nrm;
                const double v0 = x0 - alpha;
                const double beta = 1.0 / (nrm * (nrm + fabs(x0)));
                const int nc = w - j;                    // columns j+1 .. w
                int nchunk = (nc >= T) ? 1 : T / nc;
                if (nchunk > na) nchunk = na;
                const int rpc = (na + nchunk - 1) / nchunk;
                if (nchunk == 1) {
                    for (int c = j + 1 + t; c <= w; c += T) {
                        double dot = v0 * cur[c];
                        for (int b = 0; b < na; ++b) dot += batch[b * ld + j] * batch[b * ld + c];
                        const double tau = beta * dot;
                        out[(size_t)j * ldo + c] = cur[c] - tau * v0;
                        for (int b = 0; b < na; ++b) batch[b * ld + c] -= tau * batch[b * ld + j];
                    }
                } else {
                    const int ci = t % nc, ch = t / nc;
                    const int c = j + 1 + ci;
                    const bool on = ch < nchunk;
                    const int b0 = ch * rpc, b1 = min(na, b0 + rpc);
                    if (on) {
                        double dot = (ch == 0) ? v0 * cur[c] : 0.0;
                        for (int b = b0; b < b1; ++b) dot += batch[b * ld + j] * batch[b * ld + c];
                        part[ch * nc + ci] = dot;
                    }
                    __syncthreads();
                    if (on) {
                        double dot = 0.0;
                        for (int k = 0; k < nchunk; ++k) dot += part[k * nc + ci];
                        const double tau = beta * dot;
                        if (ch == 0) out[(size_t)j * ldo + c] = cur[c] - tau * v0;
                        for (int b = b0; b < b1; ++b) batch[b * ld + c] -= tau * batch[b * ld + j];
                    }
                }
                if (t == 0) out[(size_t)j * ldo + j] = alpha;
            } else {
                // nothing to eliminate in this column: R row j passes through
                for (int c = j + t; c <= w; c += T) out[(size_t)j * ldo + c] = cur[c];
            }
            if (j + 1 < w) {
                if (pc0 <= w) nxtbuf[pc0] = nx0;
                if (pc1 <= w) nxtbuf[pc1] = nx1;
            }
            __syncthreads();
            if (t == 0) rstate[j] = 2;
        }
        __syncthreads();
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
}

}  // namespace msckf
