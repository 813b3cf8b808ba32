// K5, upper part: merging upper-triangular blocks whose clone windows overlap
// (reference MSCKF.py:594-598, the same QR as k_fold.h, restricted to what is
// left once every group of equal-first-slot features has been reduced to
// triangles).  The stacked triangles form a BAND matrix (a track spans at most
// Mmax consecutive-ish clones), so the final R has band width <= 6 Mmax and a
// tree of ever wider dense merges (k_fold) wastes a chain of w column steps per
// level on it.  Here ONE workgroup keeps the band R in LDS and folds the source
// triangles into it as a systolic pipeline:
//
//   * a "fold" eliminates one source triangle (w rows, columns [off, off+w))
//     against R with one Householder reflector per column, rows held in the
//     REGISTERS of one wavefront (64 x 64 tile: lane (rq, cq) = (lane & 3, lane >> 2)
//     owns rows {rq + 4 rr} and local columns {cq + 16 k}; the rhs sits at local
//     column 63), so the dots reduce over a DPP quad only;
//   * step i of a fold touches R row off+i and nothing else of R, so fold g+1
//     may run its step for column c one macro step after fold g ran its own:
//     NW wavefronts work on NW different folds at NW different columns, one
//     workgroup barrier per macro step.  The arithmetic is exactly that of
//     folding the triangles one after the other.
//   * row r of a triangle comes alive at local column r and the columns left of
//     the pivot are retired: with the step index cut into chunks of 8 (KK = i / 8)
//     the live row slots (rr <= 2 KK + 1) and column slots (k >= KK / 2) are
//     compile-time loop bounds; each wavefront runs its own straight-line
//     sequence  [idle barriers] [chunk 0 .. chunk 7] [idle barriers] ...  and all
//     wavefronts execute the same number of barriers (nsteps).
//   * inside a step a lane forms one partial dot per live column slot (one running sum each: the step time
//     follows the FP64 instruction count, not the FMA chain); the quad's partials are reduce-scattered so
//     that row lane rq holds the dot of slot rq, i.e. every lane looks after ONE column of the pivot row:
//     it reads that R entry, forms tau, writes the entry back, and tau is broadcast over the quad for the
//     rank-1 update.
//   * the squared norm of the NEXT pivot column is taken by its owners right behind its update (one quad
//     reduction, handed on in scalar registers), so that the reflector scalars of a step -- a dependent chain of
//     ~370 cycles: rsq, rcp, two Newton steps -- no longer wait for the step's dots but run beside them
//     (round 3: the step was the sum of the two chains, now it is the longer of the two).
//   * the rows of the source triangle are fetched two row slots per chunk, one
//     chunk ahead of their first use; the first two row slots of the NEXT fold are
//     fetched during the last chunks of the current one.
//
// The host (build_plan_band) orders the folds by first column and assigns
//   t0[0] = 0 (adopted: its rows are copied into R),  t0[1] = 1,
//   t0[g] = max(t0[g-1] + off[g] - off[g-1] + 1,  t0[g-NF] + w[g-NF] + 1)
// (pipeline lag / wavefront reuse).  A fold's tile must cover the envelope of
// the R rows it meets (ew >= w: columns [off, off+ew) can fill in).
#pragma once
#include <hip/hip_runtime.h>
#include "wave_ops.h"

namespace msckf {

struct SweepFold {
    long long src_off;   // offset (doubles) of the source block in rbuf: row-major w x (w+1), upper triangular
    int off;             // first column of the source window, local to the node
    int w;               // rows / columns of the source triangle (<= SWEEP_MAX_W)
    int ew;              // tile width: columns [off, off+ew) may fill in (w <= ew <= SWEEP_MAX_W)
    int t0;              // macro step of the fold's first column
    int pad0, pad1;
};

struct SweepNode {
    int fold_begin, fold_end;   // folds [begin, end); a first fold with t0 == 0 is adopted (copied into R), the others
                                // run on fold slot (i - first scheduled) % NF
    int wtot;                   // columns of the node's R
    int nsteps;                 // macro steps
    long long out_off;          // output block in rbuf: row-major wtot x (wtot+1)
};

struct SweepArgs {
    const SweepNode* nodes;
    const SweepFold* folds;
    int node_base;
    double* rbuf;
    long long* stamps;          // optional: 8 ticks per node, may be null
    int stamp_base;
    const double* zero;         // a double that reads 0.0 (tail of the workspace)
};

constexpr int SWEEP_MAX_W = 60;        // widest source / envelope (local column 63 holds the rhs)
constexpr int SWEEP_RS = 64;           // doubles per R row in LDS: entry (c, col) at [c][col - c], rhs at [c][63]
constexpr double SWEEP_TINY = 1e-290;  // |column|^2 under this is treated as an exact zero column
constexpr int SWEEP_VB = 72;           // per-wavefront published column: [rq][16] rows + |column|^2 at [64]

__host__ __device__ inline size_t sweep_lds_bytes(int wtot, int nf, int wpf) {
    return ((size_t)wtot * SWEEP_RS + (size_t)nf * SWEEP_VB + (size_t)nf * wpf * 64 + 2 + (size_t)nf) * 8;   // R | published columns | dumps | zero | progress
}

// Sum over the 4 lanes of a DPP quad; every lane of the quad gets the sum.
template <int CTRL>
__device__ __forceinline__ double quad_move(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double x) {
    x += quad_move<0xB1>(x);    // quad_perm [1,0,3,2]
    x += quad_move<0x4E>(x);    // quad_perm [2,3,0,1]
    return x;
}

// Group exchange: the shard's accepted count (features with accepted == 1) as a double into its record and,
// with `mask`, the gate byte of every feature in INPUT order (perm: sorted position -> input index) behind it, so
// that the merging rank can hand the whole batch's gate results back with dx | P+ (reference counter MSCKF.py:578).
__global__ __launch_bounds__(256) void k_count_accepted(const unsigned char* accepted, int F, double* dst, const int* perm,
                                                        unsigned char* mask) {
    __shared__ int part[4];
    int n = 0;
    for (int f = threadIdx.x; f < F; f += 256) {
        const unsigned char a = accepted[f];
        n += a == 1;
        if (mask) mask[perm[f]] = a;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) *dst = (double)(part[0] + part[1] + part[2] + part[3]);
}

// Sum of the accepted counts the shards wrote into their group records (double N of every record) -> one int.
__global__ __launch_bounds__(64) void k_sum_record_counts(const double* recs, long long rec_stride, int N, int n_rec, int* dst) {
    double s = 0.0;
    for (int r = threadIdx.x; r < n_rec; r += 64) s += recs[(size_t)r * rec_stride + N];
    s = wave_sum(s);
    if (threadIdx.x == 0) *dst = (int)(s + 0.5);
}

// Merging rank: the gate bytes of every record (shard r: features [b[r], b[r+1]) of the whole batch, bytes at double
// `head` of its record) into the result range behind P_out; F_total into status word 3.
struct MaskBounds { int n; int b[65]; };
__global__ __launch_bounds__(256) void k_collect_masks(const double* recs, long long rec_stride, int head, MaskBounds mb,
                                                       unsigned char* dst, int* status) {
    const int r = blockIdx.x;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(recs + (size_t)r * rec_stride + head);
    const int lo = mb.b[r], n = mb.b[r + 1] - lo;
    for (int i = threadIdx.x; i < n; i += 256) dst[lo + i] = src[i];
    if (r == 0 && threadIdx.x == 0) status[3] = mb.b[mb.n];
}
__global__ void k_set_int(int* p, int v) { *p = v; }

template <int KK> struct STag { static constexpr int value = KK; };

#ifdef SWEEP_PROF
#define SWEEP_TICK(slot) do { const long long tn_ = __builtin_readcyclecounter(); prof[slot] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define SWEEP_TICK(slot) do { } while (0)
#endif

// NF concurrent folds, WPF wavefronts per fold (only 1 is enabled: splitting a fold's columns over two wavefronts
// -- 16 per workgroup, half the FMAs each -- measured the same 0.83 us per step: the step is a dependent chain
// barrier -> LDS -> reflector scalars / dots -> tau -> update -> column norm -> LDS, not an issue-rate limit).
// P2P: no workgroup barrier per macro step.  A fold's step for R row c only needs every EARLIER fold to be past
// row c; consecutive folds of the root are 7 rows apart, so there are 6 steps of slack between them.  Every fold
// slot publishes (fold index, rows done) in one LDS word after each step; before a step a wavefront makes sure
// the (up to NF - 1) folds in front of it have started and are past its row (a cached bound, re-read only when it
// is reached).  Heavy and light chunks then average out instead of every step costing the slowest wavefront's
// (the barrier wait was 660 of 1830 cycles per step).  Nodes whose folds share one first column (lag 1, no
// slack) keep the barrier form.
template <int NF, int WPF, bool P2P = false>
__global__ __launch_bounds__(64 * NF * WPF) void k_sweep(SweepArgs p) {
    static_assert(WPF == 1, "with two wavefronts per fold the pivot R(c,c) is rewritten by one while the other may still read it");
    constexpr int NW = NF * WPF;        // wavefronts
    constexpr int CL = 16 * WPF;        // column lanes of a fold
    constexpr int CS = 64 / CL;         // column slots of a lane
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SweepNode nd = p.nodes[p.node_base + blockIdx.x];
    const int t = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lane = t & 63;
    const int rq = lane & 3;            // row lane: the 4 lanes of a DPP quad
    const int fs = wv / WPF;            // fold slot
    const int cq = (lane >> 2) + 16 * (wv % WPF);   // column lane 0..CL-1
    double* Rb = smem;                                              // [wtot][SWEEP_RS]
    double* vb = smem + (size_t)nd.wtot * SWEEP_RS + fs * SWEEP_VB; // published column of this fold slot
    const int nsteps = __builtin_amdgcn_readfirstlane(nd.nsteps);
    const int fold_end = __builtin_amdgcn_readfirstlane(nd.fold_end);
    long long tk0 = 0;
    if (p.stamps) tk0 = wall_clock64();

    for (int e = t; e < nd.wtot * SWEEP_RS; e += 64 * NW) Rb[e] = 0.0;
    if (t < NF) reinterpret_cast<long long*>(smem + (size_t)nd.wtot * SWEEP_RS + NF * SWEEP_VB + NW * 64 + 2)[t] = -1LL << 32;   // fold index -1
    if (t < 2) smem[nd.wtot * SWEEP_RS + NF * SWEEP_VB + NW * 64 + t] = 0.0;
    // the node's first triangle (t0 == 0) is adopted: its rows ARE the first rows of R, nothing to eliminate
    const SweepFold f0 = p.folds[nd.fold_begin];
    const int adopt = (nd.fold_end > nd.fold_begin && f0.t0 == 0) ? 1 : 0;
    if (adopt) {
        __syncthreads();
        const double* src = p.rbuf + f0.src_off;
        const int ldw = f0.w + 1;
        for (int e = t; e < f0.w * ldw; e += 64 * NW) {
            const int r = e / ldw, lc = e - r * ldw;
            if (lc >= r) Rb[(size_t)(f0.off + r) * SWEEP_RS + (lc == f0.w ? 63 : lc - r)] = src[e];
        }
    }

    double a[16][CS];
    double nxt[2][CS];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr)
#pragma unroll
        for (int k = 0; k < CS; ++k) a[rr][k] = 0.0;

    int tcur = 0;                       // macro steps (= barriers) this wavefront has done
#ifdef SWEEP_PROF
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = __builtin_readcyclecounter();
#endif
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
    // element (row slot rr, column slot k) of a source triangle; unconditional load from a clamped address
    // (structural zeros are read from p.zero, a zero double of the workspace: no select after the load, so the
    //  wait for the data sits at its first use, chunks later)
    auto load_elem = [&](const double* src, int w, int rr, int k) -> double {
        const int r = rq + 4 * rr, lc = cq + CL * k;
        const bool isr = (k == CS - 1) && (cq == CL - 1);
        const bool ok = (r < w) && (isr || (lc >= r && lc < w));
        const int col = isr ? w : lc;
        // (64-bit addresses: a fold's source need not live in the workspace -- rank 0 folds gathered records where they lie)
        const double* q = ok ? src + (r * (w + 1) + col) : p.zero;
        return *q;
    };
    auto fetch_next_head = [&]() {      // row slots 0, 1 of the next fold
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int k = 0; k < CS; ++k) nxt[rr][k] = load_elem(n_src, n_w, rr, k);
    };

    // the owners of local column slot KN publish their column (row slots rr <= RP); its squared norm falls out
    // of the next step's dots (the owners' own dot is v^T v)
    auto publish = [&](auto tagk, auto tagr) {
        constexpr int KN = decltype(tagk)::value;
        constexpr int RP = decltype(tagr)::value < 15 ? decltype(tagr)::value : 15;
        if constexpr (KN < CS) {
            double* dst = vb + rq * 16;
#pragma unroll
            for (int rr = 0; rr <= RP; ++rr) dst[rr] = a[rr][KN];
        }
    };

    // Every lane looks after ONE entry of the pivot row of R: row lane rq of a quad takes column slot rq, i.e. local
    // column lco = cq + CL rq (the quad's four partial dots are reduce-scattered so that lane rq ends up with the
    // dot of slot rq; tau is formed once per column and broadcast back over the quad).  ra / wa: LDS indices
    // (doubles) where that entry is read / written; they advance by SWEEP_RS - 1 per step (next row, one column
    // less to the left), the rhs by SWEEP_RS.  Columns outside the tile read a zero word and write a dump word.
    static_assert(CS == 4, "one column slot per row lane of the quad");
    int ra = 0, wa = 0, rstep = 0;
    const int dump_i = nd.wtot * SWEEP_RS + NF * SWEEP_VB + wv * 64 + lane;
    const int zero_i = nd.wtot * SWEEP_RS + NF * SWEEP_VB + NW * 64;
    const int lco = cq + CL * rq;
    // ---- P2P progress words: [slot] = (fold index << 32) | rows done (first row not yet done by that fold) --------
    // (read and written with explicit ds instructions: a volatile pointer into LDS compiles to FLAT accesses with
    //  vmcnt waits, which is what made the first version of this variant slow)
    const unsigned prog_addr = (unsigned)(size_t)(smem + (size_t)nd.wtot * SWEEP_RS + NF * SWEEP_VB + NW * 64 + 2);
    auto prog_load = [&](int slot) -> long long {          // one lane asks, everybody gets the word
        double v;
        const unsigned a = prog_addr + (unsigned)slot * 8u;
        asm volatile("s_mov_b64 exec, 1\n\tds_read_b64 %0, %1\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
        return ((long long)hi << 32) | (unsigned int)lo;
    };
    auto prog_store = [&](int slot, long long w) {         // (called where the whole wavefront is active)
        const unsigned a = prog_addr + (unsigned)slot * 8u;
        const double v = __longlong_as_double(w);
        asm volatile("s_mov_b64 exec, 1\n\tds_write_b64 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(a), "v"(v) : "memory");
    };
    const int fbeg = nd.fold_begin + ((nd.fold_end > nd.fold_begin && f0.t0 == 0) ? 1 : 0);   // first fold that runs steps
    int cur_fi = 0;                      // fold this wavefront is running
    int safe_row = 0;                    // rows < safe_row are clear of every earlier fold (as of the last look)
    auto publish_prog = [&](int done) {
        if constexpr (P2P) {
            // rows < the published bound are clear of THIS fold and (transitively, through its own cached bound) of every
            // earlier one, so a follower normally looks at one word only.  LDS operations of a wavefront execute in
            // issue order: the word lands behind this step's R entries.
            asm volatile("" ::: "memory");
            const int bound = (done == 0x7fffffff) ? done : min(done, safe_row);
            prog_store(fs, ((long long)cur_fi << 32) | (unsigned int)bound);
        }
    };
    auto peek_prev = [&]() -> long long {        // the word of the fold right in front (issued early, used after the dots)
        if constexpr (P2P) {
            const int pf = cur_fi - 1;
            return (pf >= fbeg) ? prog_load((pf - fbeg) % NF) : (((long long)0x7fffffff) << 32);
        }
        return 0;
    };
    auto wait_row = [&](int c, long long w) {    // every earlier fold has left row c behind
        if constexpr (P2P) {
            int spins = 0;
            while (c >= safe_row) {
                const int pf = cur_fi - 1;
                int safe;
                const int idx1 = (int)(w >> 32), done1 = (int)(w & 0xffffffffLL);
                if (pf < fbeg) {
                    safe = 0x7fffffff;
                } else if (idx1 == pf) {
                    safe = done1;                                          // running: its bound is transitive
                } else {
                    // the fold in front has finished (or its slot has not started it yet): look at all of them
                    safe = 0x7fffffff;
#pragma unroll
                    for (int j = 1; j < NF; ++j) {
                        const int qf = cur_fi - j;
                        if (qf >= fbeg) {
                            const long long wj = prog_load((qf - fbeg) % NF);
                            const int idx = (int)(wj >> 32), done = (int)(wj & 0xffffffffLL);
                            const int lim = (idx > qf) ? 0x7fffffff : (idx == qf ? done : 0);   // finished / running / not started
                            safe = min(safe, lim);
                        }
                    }
                }
                safe_row = __builtin_amdgcn_readfirstlane(safe);
                if (c >= safe_row) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 22)) break;                        // (bounded: never hang the GPU)
                    w = peek_prev();
                }
            }
            asm volatile("" ::: "memory");
        }
    };
    auto init_addr = [&]() {
        const bool isr = (rq == CS - 1) && (cq == CL - 1);
        const bool valid = isr || lco < f_ew;
        ra = valid ? f_off * SWEEP_RS + (isr ? 63 : lco) : zero_i;
        wa = valid ? ra : dump_i;
        rstep = valid ? (isr ? SWEEP_RS : SWEEP_RS - 1) : 0;
    };

    // |next pivot column|^2 (column slot KN, rows rr <= RP) as a wave-uniform value: four partial sums in every lane over
    // ITS column of the slot (only the owners' sum is used), one quad reduction, v_readlane from the owners' quad
    double sg_cur = 0.0;                                  // |pivot column|^2 of the step about to run
    auto next_norm = [&](auto tagk, auto tagr, int col) {
        constexpr int KN = decltype(tagk)::value;
        constexpr int RP = decltype(tagr)::value < 15 ? decltype(tagr)::value : 15;
        if constexpr (KN < CS) {
            double p0 = a[0][KN] * a[0][KN], p1 = 0.0, p2 = 0.0, p3 = 0.0;
            if constexpr (RP >= 1) p1 = a[1][KN] * a[1][KN];
            if constexpr (RP >= 2) p2 = a[2][KN] * a[2][KN];
            if constexpr (RP >= 3) p3 = a[3][KN] * a[3][KN];
#pragma unroll
            for (int rr = 4; rr <= RP; ++rr) {
                if ((rr & 3) == 0) p0 = fma(a[rr][KN], a[rr][KN], p0);
                else if ((rr & 3) == 1) p1 = fma(a[rr][KN], a[rr][KN], p1);
                else if ((rr & 3) == 2) p2 = fma(a[rr][KN], a[rr][KN], p2);
                else p3 = fma(a[rr][KN], a[rr][KN], p3);
            }
            const double pn = quad_sum((p0 + p1) + (p2 + p3));
            sg_cur = readlane_d(pn, 4 * (col & (CL - 1)));
        }
    };

    auto step = [&](auto tagk, int i) {
        constexpr int KK = decltype(tagk)::value;
        constexpr int RMAX = 2 * KK + 1;                  // live row slots (rows <= 8 KK + 7)
        constexpr int K0 = (8 * KK) / CL;                 // first live column slot
        const int rrow = (f_off + i) * SWEEP_RS;          // pivot row of R (uniform)
        // ---- reads ------------------------------------------------------------------
        double v[RMAX + 1];
        {
            const double* src = vb + rq * 16;
#pragma unroll
            for (int rr = 0; rr <= RMAX; ++rr) v[rr] = src[rr];
        }
        long long wprev = 0;
        double x0 = 0.0, rck = 0.0;
        if constexpr (P2P) {
            wprev = peek_prev();                           // looked at behind the dots, which need no R
        } else {
            x0 = smem[rrow];
            rck = smem[ra];                                // R(c, off + lco)
        }
        const bool on = (lco > i) || (rq == CS - 1 && cq == CL - 1);   // left of / at the pivot: retired (the rhs never is)
        // |column i|^2 was taken when the column was last updated (next_norm)
        const double sg = sg_cur;
        const bool live = sg > SWEEP_TINY;                // wave-uniform; below: nothing to eliminate
        double ss = 1.0, y = 1.0, nrm = 1.0, beta = 0.0, alpha = 0.0, v0 = 0.0;
        auto scalars = [&]() {
            // ---- reflector scalars (every lane, uniform values) ---------------------------
            // Branch-free: sg > 1e-290 keeps ss = x0^2 + sg a normal number whose rsqrt / rcp seeds + one Newton step
            // are good to a few 1e-16 (the reflector stays orthogonal to that level); a column with nothing to
            // eliminate gets beta = 0, alpha = x0 (identity).  |x0| > 1e150 does not occur (R entries are bounded by
            // the column norms of a normalised-coordinate Jacobian stack).
            ss = live ? fma(x0, x0, sg) : 1.0;
            y = fast_rsqrt(ss);
            nrm = ss * y;
            beta = live ? y * fast_rcp(nrm + fabs(x0)) : 0.0;      // 1 / (nrm (nrm + |x0|))
            alpha = live ? ((x0 > 0.0) ? -nrm : nrm) : x0;
            v0 = x0 - alpha;
        };
        if constexpr (!P2P) scalars();                    // (independent of the dots: the two chains overlap)
        // ---- dots (one partial sum per live slot), reduce-scattered over the quad ---------------------
        double sp[CS];
#pragma unroll
        for (int k = 0; k < CS; ++k) {
            sp[k] = 0.0;
            if (k >= K0) {
                double s0 = v[0] * a[0][k];
#pragma unroll
                for (int rr = 1; rr <= RMAX; ++rr) s0 = fma(v[rr], a[rr][k], s0);
                sp[k] = s0;
            }
        }
        double tot;                                       // lane rq: the full dot of slot rq
        {
            const bool b0 = (rq & 1) != 0, b1 = (rq & 2) != 0;
            double pB = (b0 ? sp[3] : sp[2]) + quad_move<0xB1>(b0 ? sp[2] : sp[3]);
            if constexpr (K0 <= 1) {
                double pA = (b0 ? sp[1] : sp[0]) + quad_move<0xB1>(b0 ? sp[0] : sp[1]);
                tot = (b1 ? pB : pA) + quad_move<0x4E>(b1 ? pA : pB);
            } else {
                tot = pB + quad_move<0x4E>(pB);           // slots 0, 1 are retired: lanes 0, 1 hold a copy nobody uses
                (void)b1;
            }
        }
        if constexpr (P2P) {
            wait_row(f_off + i, wprev);                    // the earlier folds are past this row of R
            x0 = smem[rrow];
            rck = smem[ra];
            scalars();
        }
        // ---- tau of this lane's column, its R entry, then the rank-1 update of every slot ----------
        const double tau_own = (on ? beta : 0.0) * fma(v0, rck, tot);
        smem[on ? wa : dump_i] = fma(-tau_own, v0, rck);
        if (rq == 0 && cq == 0) smem[rrow] = alpha;
        ra += rstep; wa += rstep;
        auto slot = [&](auto tags) {
            constexpr int k = decltype(tags)::value;
            if constexpr (k < CS) {
                constexpr int CTRL = (k == 0) ? 0x00 : (k == 1) ? 0x55 : (k == 2) ? 0xAA : 0xFF;   // quad_perm [k,k,k,k]
                const double tau = quad_move<CTRL>(tau_own);
#pragma unroll
                for (int rr = 0; rr <= RMAX; ++rr) a[rr][k] = fma(-tau, v[rr], a[rr][k]);
            }
        };
        // the slot of the next pivot column first, then its owners publish it and take its norm while the other slots update
        const int in = i + 1;
        slot(STag<K0>{});
        if (in < f_ew && (in & 7) != 0) {
            if (cq == (in & (CL - 1))) publish(STag<K0>{}, STag<RMAX>{});         // same chunk: in / CL == K0
            next_norm(STag<K0>{}, STag<RMAX>{}, in);
        }
        slot(STag<K0 + 1>{});
        slot(STag<K0 + 2>{});
        slot(STag<K0 + 3>{});
        if (in < f_ew && (in & 7) == 0) {
            // first column of the next chunk: slot 8 (KK + 1) / CL, column lane 8 (KK + 1) % CL, two more row slots
            constexpr int KN = (8 * (KK + 1)) / CL;
            if (cq == (8 * (KK + 1)) % CL) publish(STag<KN>{}, STag<RMAX + 2>{});
            next_norm(STag<KN>{}, STag<RMAX + 2>{}, in);
        }
    };

    // one chunk of 8 columns: fetch the two row slots the chunk's LAST publish needs, then the steps
    auto chunk = [&](auto tagk, bool have_next) {
        constexpr int KK = decltype(tagk)::value;
        if (8 * KK >= f_ew) return;
#pragma unroll
        for (int rr = 2 * KK + 2; rr <= 2 * KK + 3 && rr < 16; ++rr) {
#pragma unroll
            for (int k = 0; k < CS; ++k) {
                if (CL * k + CL - 1 >= 4 * rr) a[rr][k] = load_elem(f_src, f_w, rr, k);   // else structurally zero, never read
            }
        }
        if (have_next && KK == max((f_ew - 1) / 8 - 1, 0)) fetch_next_head();
        const int ihi = min(8 * KK + 8, f_ew);        // (steps run over the ENVELOPE: the tile's rows fill in right of the
                                                                // source's last column wherever R already reaches further, and that fill has to be eliminated too)
        for (int i = 8 * KK; i < ihi; ++i) {
            SWEEP_TICK(0);                                 // left the barrier
            step(tagk, i);
#ifdef SWEEP_PROF
            __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the tick sees the LDS writes out
#endif
            SWEEP_TICK(4);                                 // updates, R writes, publish issued
            if constexpr (P2P) {
                publish_prog(i + 1 < f_ew ? f_off + i + 1 : 0x7fffffff);
            } else {
                __syncthreads();
            }
            SWEEP_TICK(5);                                 // barrier wait
            ++tcur;
        }
    };

    __syncthreads();                                       // R zeroed
    int fi = nd.fold_begin + adopt + fs;
    bool have = fi < fold_end;
    if (have) {
        read_desc(fi, n_off, n_w, n_ew, n_t0, n_src);
        fetch_next_head();
    }
    while (have) {
        f_off = n_off; f_w = n_w; f_ew = n_ew; f_t0 = n_t0; f_src = n_src;
        if constexpr (P2P) {
            cur_fi = fi;
            safe_row = 0;
            publish_prog(f_off);                           // started: nothing done yet
        } else {
            while (tcur < f_t0 - 1) { __syncthreads(); ++tcur; }
        }
        init_addr();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int k = 0; k < CS; ++k) a[rr][k] = nxt[rr][k];
        if (cq == 0) publish(STag<0>{}, STag<1>{});
        next_norm(STag<0>{}, STag<1>{}, 0);
        if constexpr (!P2P) {
            __syncthreads();                               // column 0 is visible to the fold's other wavefront
            ++tcur;                                        // (the host schedules t0 >= 1 and one spare step per slot reuse)
        }
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
        have = have_next;
    }
    if constexpr (!P2P) {
        while (tcur < nsteps) { __syncthreads(); ++tcur; }
    }

    // ---- flush R: row-major wtot x (wtot+1), entries at and right of the diagonal ----
    __syncthreads();
    double* out = p.rbuf + nd.out_off;
    const int ldo = nd.wtot + 1;
    for (int c = wv; c < nd.wtot; c += NW) {
        const double* Rrow = Rb + (size_t)c * SWEEP_RS;
        for (int col = c + lane; col < nd.wtot; col += 64) {
            const int dlt = col - c;
            out[(size_t)c * ldo + col] = (dlt < 63) ? Rrow[dlt] : 0.0;
        }
        if (lane == 0) out[(size_t)c * ldo + nd.wtot] = Rrow[63];
    }
    if (p.stamps && t == 0) {
        long long* o = p.stamps + 8 * (p.stamp_base + blockIdx.x);
        o[0] = 0; o[1] = 0; o[2] = 0; o[3] = wall_clock64() - tk0; o[4] = nd.wtot; o[5] = nd.nsteps;
    }
#ifdef SWEEP_PROF
    if (p.stamps && lane == 0 && blockIdx.x == 0) {
        long long* o = p.stamps + 16 * 8192 + 8 * wv;
        for (int q = 0; q < 8; ++q) o[q] = prof[q];
    }
#endif
}

}  // namespace msckf
