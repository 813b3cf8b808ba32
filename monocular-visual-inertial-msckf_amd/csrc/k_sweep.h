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
//     REGISTERS of one wavefront (64 x 64 tile: lane (rq, cq) = (lane >> 4, lane & 15)
//     owns rows {rq + 4 rr} and local columns {cq + 16 k}; the rhs sits at local
//     column 63).  The column step itself is sweep_step.h's: the pivot column reaches
//     the tile's other lanes through the DPP row broadcast of the FMAs, nothing of the
//     tile goes through LDS;
//   * step i of a fold touches R row off+i and nothing else of R, so fold g+1
//     may run its step for column c one macro step after fold g ran its own:
//     NW wavefronts work on NW different folds at NW different columns, one
//     workgroup barrier per macro step.  The arithmetic is exactly that of
//     folding the triangles one after the other.
//   * row r of a triangle comes alive at local column r and the columns left of
//     the pivot are retired: with the step index cut into chunks of 8 (KK = i / 8)
//     the live row slots (rr <= 2 KK + 1) and column slots (k >= i / 16) are
//     compile-time loop bounds -- and so is the broadcast lane i % 16: one instance
//     of the step per column.  Each wavefront runs its own straight-line sequence
//     [idle barriers] [chunk 0 .. chunk 7] [idle barriers] ...  and all wavefronts
//     execute the same number of barriers (nsteps).
//   * the rows of the source triangle are fetched two row slots per chunk, one
//     chunk ahead of their first use; the first two row slots of the NEXT fold are
//     fetched during the last chunks of the current one.
//   (Round 3 history: the step was measured to follow the number of instructions a wavefront issues, not its
//    dependent chains -- weaving the reflector scalars into the dots, +15 instructions, cost 94.6 -> 105 us per
//    launch; halving the FMAs saved 3.5 %; taking the LDS trip of the pivot column out, -30 instructions, 95.5 -> 74 us.)
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
    int prod;            // 0, or 1 + index of the progress word of the node that is still WRITING the source block in the same
                         // launch (k_root_gain's merge workgroups): the flusher lets a macro step start only when the rows its
                         // folds fetch in it are published (host table, sweep_gate_table)
    int ld;              // doubles per row of the source block (0: w + 1).  A streamed block has whole cache lines per row (64):
                         // a line fetched for a published row must not hold part of a row that is not final yet
};

struct SweepNode {
    int fold_begin, fold_end;   // folds [begin, end); a first fold with t0 == 0 is adopted (copied into R), the others
                                // run on fold slot (i - first scheduled) % NF
    int wtot;                   // columns of the node's R
    int nsteps;                 // macro steps
    long long out_off;          // output block in rbuf: row-major wtot x (wtot+1), or wtot x ldo
    int ldo;                    // doubles per row of the output block (0: wtot + 1)
    int pad;
};

struct SweepArgs {
    const SweepNode* nodes;
    const SweepFold* folds;
    int node_base;
    double* rbuf;
    long long* stamps;          // optional: 8 ticks per node, may be null
    int stamp_base;
    const double* zero;         // a double that reads 0.0 (tail of the workspace)
    // k_sweep<.., FL = true> (the root sweep of an update whose K6-K7 follows it row block by row block, k_gstream.h):
    const int* flush_tab;       // nsteps + 1 entries lo | n << 16: rows [lo, lo + n) of R are final at the head of macro step t
    unsigned long long* progress;   // (epoch << 32) | rows of the output block that are final AND visible device-wide
    unsigned epoch;
    long long* tstamp;          // optional: [0] wall clock (10 ns ticks) when the sweep starts, [1] when its last row is published
    const int* flush_off;       // optional: per node (index in the launch) the offset of its table in flush_tab
    int prog_stride;            // progress word of node i of the launch: progress[i * prog_stride] (0: one word, the root's)
    int pub_shift;              // rows are published in blocks of 1 << pub_shift (0 = 4: k_gstream.h's row blocks, aligned to
                                // the END of the block; 3: blocks of 8 from row 0, what a fold fetches per chunk)
    // the consumer of streamed sources (the root inside k_root_gain's launch):
    const unsigned long long* src_progress;   // progress words of the producing nodes (SweepFold::prod), null: nothing is streamed
    int n_prod;                               // ... how many (<= 64)
    // behind the nsteps + 1 flush entries of the node's table (host: sweep_gate_table): nsteps + 2 step entries (what must be
    // published before macro step t starts: up to two requirements prod << 6 | rows, 12 bits each) | n_gate requirements of step 0
    int n_gate;
};

constexpr int SWEEP_MAX_W = 60;        // widest source / envelope (local column 63 holds the rhs)
constexpr int SWEEP_RS = 64;           // doubles per R row in LDS: entry (c, col) at [c][col - c], rhs at [c][63]
constexpr double SWEEP_TINY = 1e-290;  // |column|^2 under this is treated as an exact zero column
constexpr long long SWEEP_TIMEOUT_TICKS = 50000000;   // 0.5 s of the 100 MHz wall clock: a streamed source that never arrives
}  // namespace msckf
#include "sweep_step.h"
namespace msckf {

__host__ __device__ inline size_t sweep_lds_bytes(int wtot, int nf, int wpf) {
    return ((size_t)wtot * SWEEP_RS + (size_t)nf * wpf * 64 + 2) * 8;   // R | dump words | zero words
}
// ... with the flusher's table (nsteps + 1 ints) behind them
__host__ __device__ inline size_t sweep_lds_bytes_fl(int wtot, int nf, int nsteps, int n_gate = -1) {
    return sweep_lds_bytes(wtot, nf, 1) + (((size_t)nsteps + 2 + (n_gate >= 0 ? nsteps + 2 + n_gate : 0)) * 4 + 15) / 16 * 16;
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
#define SWEEP_TICK(slot) do { __builtin_amdgcn_sched_barrier(0); const long long tn_ = __builtin_readcyclecounter(); prof[slot] += tn_ - tprev; tprev = tn_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SWEEP_TICK(slot) do { } while (0)
#endif

// NF concurrent folds, one wavefront each.  Lane (rq, cq) = (lane >> 4, lane & 15): the 16 column lanes of one row
// lane are one DPP row, so the pivot column -- held by column lane i & 15 of every row -- reaches the other 15 through
// the row_newbcast operand of the FMA itself: the dots and the rank-1 update read it straight from the owner's
// registers, nothing of the fold goes through LDS (rounds 1-2 published the column there and read it back: two LDS
// trips and (RMAX + 1) / 2 wide reads per step on the step's dependent chain).  The broadcast lane is an immediate,
// hence one instantiation of the step per column of a 16-column period (KK, J).  The four row lanes' partial dots
// are reduce-scattered with the lane swaps (row r ends up with the dot of column slot r: 6 swaps and 3 additions)
// and tau goes back the same way.
// XW: wavefronts of the workgroup between the fold slots and the flusher that have no part in the sweep (a launch whose other
// workgroups need more wavefronts than this node has fold slots, k_root_gain_m): they help with the prologue and leave.
template <int NF, int WPF, bool P2P, bool FL, int XW = 0>
__device__ __forceinline__ void sweep_body(const SweepArgs& p, const int bidx) {
    static_assert(WPF == 1 && !P2P, "one wavefront per fold, one barrier per macro step");
    static_assert(XW == 0 || FL, "spare wavefronts only in the fused launches");
    constexpr int NW = NF;              // fold wavefronts
    constexpr int NT = 64 * (NF + XW + (FL ? 1 : 0));   // threads: with FL one more wavefront, the flusher
    constexpr int CL = 16;              // column lanes of a fold
    constexpr int CS = 4;               // column slots of a lane
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SweepNode nd = p.nodes[p.node_base + bidx];
    const int t = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lane = t & 63;
    const int rq = lane >> 4;           // row lane: one DPP row of 16 lanes
    const int fs = wv;                  // fold slot
    const int cq = lane & 15;           // column lane
    double* Rb = smem;                                              // [wtot][SWEEP_RS]
    const int nsteps = __builtin_amdgcn_readfirstlane(nd.nsteps);
    const int fold_end = __builtin_amdgcn_readfirstlane(nd.fold_end);
    long long tk0 = 0;
    if (p.stamps) tk0 = wall_clock64();

    for (int e = t; e < nd.wtot * SWEEP_RS; e += NT) Rb[e] = 0.0;
    if (t < 2) smem[nd.wtot * SWEEP_RS + NW * 64 + t] = 0.0;
    int* ftab = reinterpret_cast<int*>(smem + (size_t)nd.wtot * SWEEP_RS + NW * 64 + 2);
    if constexpr (FL) {
        // (the flusher reads its table from LDS: a vector-memory load per step would make it wait for its own stores)
        const int* tab = p.flush_tab + (p.flush_off ? p.flush_off[bidx] : 0);
        const int ntab = nsteps + 1 + (p.src_progress ? nsteps + 2 + p.n_gate : 0);
        for (int e = t; e < ntab; e += NT) ftab[e] = tab[e];
    }
    // the node's first triangle (t0 == 0) is adopted: its rows ARE the first rows of R, nothing to eliminate
    const SweepFold f0 = p.folds[nd.fold_begin];
    const int adopt = (nd.fold_end > nd.fold_begin && f0.t0 == 0) ? 1 : 0;
    if (adopt) {
        __syncthreads();
        const double* src = p.rbuf + f0.src_off;
        const int ldw = f0.w + 1, lds_ = f0.ld ? f0.ld : ldw;
        for (int e = t; e < f0.w * ldw; e += NT) {
            const int r = e / ldw, lc = e - r * ldw;
            if (lc >= r) Rb[(size_t)(f0.off + r) * SWEEP_RS + (lc == f0.w ? 63 : lc - r)] = src[(size_t)r * lds_ + lc];
        }
    }

    if constexpr (FL) {
        // The flusher: at the head of every macro step it writes the rows of R that no present or future fold step touches
        // (the schedule is static: flush_tab, made by the host's sweep_flush_table) to the output block with write-through
        // stores and publishes how many rows are final.  A row's count goes out three steps after its stores, behind a
        // COUNTED wait (vector-memory instructions of one wavefront complete in issue order on gfx9-family parts, so
        // `vmcnt(M)` with M <= the instructions issued since leaves exactly the younger ones in flight): a `vmcnt(0)` per
        // step would hold the flusher -- and with it the step's barrier -- for a store's round trip.  It joins the
        // barriers bare (s_barrier without the waits of __syncthreads()).
        if constexpr (XW > 0) {
            if (wv >= NF && wv < NF + XW) { __syncthreads(); return; }        // (their share of the zeroed R is in place)
        }
        if (wv == NF + XW) {
            const int ldo = nd.ldo ? nd.ldo : nd.wtot + 1;
            double* out = p.rbuf + nd.out_off;
            const int wtot = __builtin_amdgcn_readfirstlane(nd.wtot);
            const unsigned long long ep = (unsigned long long)p.epoch << 32;
            int c1 = 0, c2 = 0;                     // vector-memory instructions issued one / two steps ago
            int l1 = 0, l2 = 0, l3 = 0;             // rows final one / two / three steps ago
            int published = 0;
            const int psh = p.pub_shift ? p.pub_shift : 4;
            const int boff = p.pub_shift ? 0 : (16 - (wtot & 15)) & 15;   // k_gstream.h's row blocks end at rows = wtot (mod 16)
            auto* progw = (__attribute__((address_space(1))) unsigned long long*)(p.progress + (size_t)bidx * p.prog_stride);
            // Streamed sources: `seen_rows` (lane l: rows of producer l published in this launch, as last read) is refreshed only
            // when a requirement is not covered by it -- while the producers are still running the sweep has to wait for them
            // anyway, and once they are through ONE refresh covers every later requirement.  Per step at most two requirements
            // (prod << 6 | rows, 12 bits each: the host moves a third to an earlier step), read with the step's flush entry.
            const bool gated = p.n_gate >= 0 && p.src_progress;
            const int* gtab = ftab + nsteps + 1;            // [nsteps + 2]: entry t = what must be published before macro step t starts
            const int* g0 = gtab + nsteps + 2;              // [n_gate]: ... before the first fetch (step 0)
            int seen_rows = 0;
            bool dead = false;
            auto need = [&](int r) {
                const int prod = r >> 6, rows = r & 63;
                long long tstart = 0;
                while (__builtin_amdgcn_readlane(seen_rows, prod) < rows) {
                    if (tstart == 0) tstart = wall_clock64();
                    else { __builtin_amdgcn_s_sleep(1); if (wall_clock64() - tstart > SWEEP_TIMEOUT_TICKS) { dead = true; break; } }
                    const unsigned long long v = (lane < p.n_prod) ? __hip_atomic_load((__attribute__((address_space(1))) unsigned long long*)(unsigned long long*)(p.src_progress + lane),
                                                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                    seen_rows = ((unsigned)(v >> 32) == p.epoch) ? (int)(unsigned)v : 0;
                }
            };
            if (p.tstamp && lane == 0) p.tstamp[0] = wall_clock64();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // R zeroed / adopted, table in place
            asm volatile("" ::: "memory");
            if (gated) for (int k = 0; k < p.n_gate && !dead; ++k) need(__builtin_amdgcn_readfirstlane(g0[k]));   // the rows the fold slots fetch before their first step
            __builtin_amdgcn_s_barrier();           // (the fold wavefronts' second barrier in front of their first fetch)
            asm volatile("" ::: "memory");
            for (int ts = 0; ts <= nsteps; ++ts) {
                const int e = __builtin_amdgcn_readfirstlane(ftab[ts]);
                const int ge = gated ? __builtin_amdgcn_readfirstlane(gtab[ts + 1]) : 0;     // (what the folds fetch at the head of the next step)
                const int lo = e & 0xFFFF, n = e >> 16;
                for (int c = lo; c < lo + n; ++c) {
                    const double x = Rb[(size_t)c * SWEEP_RS + lane];
                    const int col = (lane == 63) ? wtot : c + lane;
                    if (lane == 63 || col < wtot) {
                        __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)(unsigned long long*)(out + (size_t)c * ldo + col),
                                           (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                int issued = n;
                if (((l3 + boff) >> psh) > ((published + boff) >> psh) && !dead) {   // (the reader takes rows in blocks of 16, the short block first)
                    // the stores of rows < l3 were issued three steps ago or earlier: c2 + c1 + n instructions since
                    const int m = c2 + c1 + n;
                    switch (m < 7 ? m : 7) {
                        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                        default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                    }
                    if (lane == 0) __hip_atomic_store(progw, ep | (unsigned)l3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    published = l3;
                    ++issued;
                    if (p.tstamp && lane == 0) { const int kb = (l3 + boff) >> psh; if (kb < 14) p.tstamp[18 + kb] = wall_clock64(); }
                }
                l3 = l2; l2 = l1; l1 = lo + n;
                c2 = c1; c1 = issued;
                if (ts < nsteps) {
                    if (ge != 0 && !dead) { need(ge & 4095); if ((ge >> 12) != 0 && !dead) need(ge >> 12); }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0 && !dead) __hip_atomic_store(progw, ep | (unsigned)wtot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.tstamp && lane == 0) p.tstamp[1] = wall_clock64();
            return;
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
    int f_off = 0, f_w = 0, f_ew = 0, f_t0 = 0, f_ld = 1;
    const double* f_src = p.rbuf;
    int n_off = 0, n_w = 0, n_ew = 0, n_t0 = 0, n_ld = 1;
    const double* n_src = p.rbuf;

    auto read_desc = [&](int fi, int& o_off, int& o_w, int& o_ew, int& o_t0, int& o_ld, const double*& o_src) {
        const SweepFold f = p.folds[fi];
        o_off = __builtin_amdgcn_readfirstlane(f.off);
        o_w = __builtin_amdgcn_readfirstlane(f.w);
        o_ew = __builtin_amdgcn_readfirstlane(f.ew);
        o_t0 = __builtin_amdgcn_readfirstlane(f.t0);
        o_ld = __builtin_amdgcn_readfirstlane(f.ld ? f.ld : f.w + 1);
        o_src = p.rbuf + f.src_off;
    };
    // element (row slot rr, column slot k) of a source triangle; unconditional load from a clamped address
    // (structural zeros are read from p.zero, a zero double of the workspace: no select after the load, so the
    //  wait for the data sits at its first use, chunks later)
    auto load_elem = [&](const double* src, int w, int ld, int rr, int k) -> double {
        const int r = rq + 4 * rr, lc = cq + CL * k;
        const bool isr = (k == CS - 1) && (cq == CL - 1);
        const bool ok = (r < w) && (isr || (lc >= r && lc < w));
        const int col = isr ? w : lc;
        // (64-bit addresses: a fold's source need not live in the workspace -- rank 0 folds gathered records where they lie)
        const double* q = ok ? src + (r * ld + col) : p.zero;
        return *q;
    };
    auto fetch_next_head = [&]() {      // row slots 0, 1 of the next fold
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int k = 0; k < CS; ++k) nxt[rr][k] = load_elem(n_src, n_w, n_ld, rr, k);
    };

    // Every lane looks after ONE entry of the pivot row of R: row lane rq takes column slot rq, i.e. local column
    // lco = cq + CL rq (the four partial dots are reduce-scattered so that row rq ends up with the dot of slot rq; tau
    // is formed once per column and sent back over the rows).  ra / wa: LDS indices (doubles) where that entry is
    // read / written; they advance by SWEEP_RS - 1 per step (next row, one column less to the left), the rhs by
    // SWEEP_RS.  Columns outside the tile read a zero word and write a dump word.
    int ra = 0, wa = 0, rstep = 0;
    const int dump_i = nd.wtot * SWEEP_RS + wv * 64 + lane;
    const int zero_i = nd.wtot * SWEEP_RS + NW * 64;
    const int lco = cq + CL * rq;
    const bool isr_lane = (rq == CS - 1) && (cq == CL - 1);
    auto init_addr = [&]() {
        const bool valid = isr_lane || lco < f_ew;
        ra = valid ? f_off * SWEEP_RS + (isr_lane ? 63 : lco) : zero_i;
        wa = valid ? ra : dump_i;
        rstep = valid ? (isr_lane ? SWEEP_RS : SWEEP_RS - 1) : 0;
    };

    auto step = [&](auto tagk, auto tagj) {
        constexpr int KK = decltype(tagk)::value;
        constexpr int I = 8 * KK + decltype(tagj)::value;  // the fold's column
        constexpr int RMAX = 2 * KK + 1;                  // live row slots (rows <= 8 KK + 7)
        constexpr int K0 = I / CL;                        // first live column slot: the pivot column's
        constexpr int L = I % CL;                         // column lane that holds the pivot column
        const int rrow = (f_off + I) * SWEEP_RS;          // pivot row of R (uniform)
        int lcl = lco;
        asm volatile("" : "+v"(lcl));                     // (or the 60 lane masks lco > I are hoisted out of the fold loop and spilled)
        const bool on = (lcl > I) || isr_lane;            // left of / at the pivot: retired (the rhs never is)
#ifdef SWEEP_PROF
        sweep_column_step<CS, 16, RMAX + 1, K0, L>(a, smem, rrow, ra, wa, on, 0, 0, false, dump_i, lane, prof, &tprev);
#else
        sweep_column_step<CS, 16, RMAX + 1, K0, L>(a, smem, rrow, ra, wa, on, 0, 0, false, dump_i, lane);
#endif
        ra += rstep; wa += rstep;
    };

    // one chunk of 8 columns: fetch the two row slots the NEXT chunk's first column needs, then the steps
    auto chunk = [&](auto tagk, bool have_next) {
        constexpr int KK = decltype(tagk)::value;
        if (8 * KK >= f_ew) return;
#pragma unroll
        for (int rr = 2 * KK + 2; rr <= 2 * KK + 3 && rr < 16; ++rr) {
#pragma unroll
            for (int k = 0; k < CS; ++k) {
                if (CL * k + CL - 1 >= 4 * rr) a[rr][k] = load_elem(f_src, f_w, f_ld, rr, k);   // else structurally zero, never read
            }
        }
        if (have_next && KK == max((f_ew - 1) / 8 - 1, 0)) fetch_next_head();
        // (steps run over the ENVELOPE: the tile's rows fill in right of the source's last column wherever R already
        //  reaches further, and that fill has to be eliminated too)
        auto one = [&](auto tagj) {
            if (8 * KK + decltype(tagj)::value < f_ew) {
                SWEEP_TICK(0);                             // left the barrier
                step(tagk, tagj);
                SWEEP_TICK(4);                             // updates issued
                __syncthreads();
                SWEEP_TICK(5);                             // barrier wait
#ifdef SWEEP_PROF
                prof[7] += 1;
#endif
                ++tcur;
            }
        };
        one(STag<0>{}); one(STag<1>{}); one(STag<2>{}); one(STag<3>{});
        one(STag<4>{}); one(STag<5>{}); one(STag<6>{}); one(STag<7>{});
    };

    __syncthreads();                                       // R zeroed
    if constexpr (FL) __syncthreads();                     // (the flusher has seen to it that the first rows are there)
    int fi = nd.fold_begin + adopt + fs;
    bool have = fi < fold_end;
    if (have) {
        read_desc(fi, n_off, n_w, n_ew, n_t0, n_ld, n_src);
        fetch_next_head();
    }
    while (have) {
        f_off = n_off; f_w = n_w; f_ew = n_ew; f_t0 = n_t0; f_ld = n_ld; f_src = n_src;
        while (tcur < f_t0) { __syncthreads(); ++tcur; }   // (the host schedules t0 >= 1 and one spare step per slot reuse)
        init_addr();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int k = 0; k < CS; ++k) a[rr][k] = nxt[rr][k];
        fi += NF;
        const bool have_next = fi < fold_end;
        if (have_next) read_desc(fi, n_off, n_w, n_ew, n_t0, n_ld, n_src);
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
    while (tcur < nsteps) { __syncthreads(); ++tcur; }
    if constexpr (FL) return;                              // (the flusher writes the output block, row by row as it becomes final)

    // ---- flush R: row-major wtot x (wtot+1), entries at and right of the diagonal ----
    __syncthreads();
    double* out = p.rbuf + nd.out_off;
    const int ldo = nd.ldo ? nd.ldo : nd.wtot + 1;
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

template <int NF, int WPF, bool P2P = false, bool FL = false>
__global__ __launch_bounds__(64 * (NF + (FL ? 1 : 0)) * WPF) void k_sweep(SweepArgs p) {
    sweep_body<NF, WPF, P2P, FL>(p, (int)blockIdx.x);
}

}  // namespace msckf
