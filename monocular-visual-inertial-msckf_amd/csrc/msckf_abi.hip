// Host side of the C-ABI (include/msckf_mi355x.h): context, HBM buffers, the
// feature sort + QR-tree plan, and the launch sequence K1..K7 on one HIP stream.
// gfx950 only.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>      // types only: librccl is loaded with dlopen at the first msckf_comm_* call

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <pthread.h>
#include <sched.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/msckf_mi355x.h"
#include "k_feature.h"
#include "k_fold.h"
#include "k_gain.h"
#include "k_gstream.h"
#include "k_gdense.h"
#include "k_select.h"
#include "k_sweep.h"
#include "k_wsweep.h"
#include "k_lsweep.h"
#include "k_gather.h"
#include "k_state.h"
#include "k_assoc.h"

using namespace msckf;

namespace {

#ifndef MSCKF_FOLD_T
#define MSCKF_FOLD_T 512
#endif
#ifndef MSCKF_FOLD_RL
#define MSCKF_FOLD_RL 16
#endif
constexpr int FOLD_T = MSCKF_FOLD_T;             // main fold kernel threads
constexpr int FOLD_RL = MSCKF_FOLD_RL;           // row lanes (8 or 16)
constexpr int FOLD_NCG = FOLD_T / FOLD_RL;       // column groups of the fold kernel
// register tile variants: window classes W1/W2/W3 (w+1 <= 64 / 128 / 192 columns) x batch rows
constexpr int FOLD_CPT1 = (64 + FOLD_NCG - 1) / FOLD_NCG;
constexpr int FOLD_CPT2 = (128 + FOLD_NCG - 1) / FOLD_NCG;
constexpr int FOLD_CPT3 = (192 + FOLD_NCG - 1) / FOLD_NCG;
constexpr int FOLD_RPT_LEAF = ((160 / FOLD_RL + 3) / 4) * 4;   // <= 160-row leaves
constexpr int FOLD_RPT_BIG = 256 / FOLD_RL;                     // 256-row batches (W1, W2)
constexpr int FOLD_RPT_W3 = 192 / FOLD_RL;                      // 192-row batches (W3)
#ifndef MSCKF_CHOL_T
#define MSCKF_CHOL_T 768
#endif
constexpr int CHOL_T = MSCKF_CHOL_T;             // threads of the register-tiled Cholesky
#ifndef MSCKF_CHOL16
#define MSCKF_CHOL16 1                           // 1: k_chol16 (16 x 16 matrix-core blocks), 0: k_chol_tile (4 x 4 register tiles)
#endif
#ifndef MSCKF_SOLVE_WAVES
#define MSCKF_SOLVE_WAVES 8
#endif
constexpr int SOLVE_WAVES = MSCKF_SOLVE_WAVES;   // wavefronts per workgroup of k_solve_lds
#ifndef MSCKF_SOLVE_ROWS
#define MSCKF_SOLVE_ROWS 1
#endif
constexpr int SOLVE_ROWS = MSCKF_SOLVE_ROWS;     // rows of Y per wavefront
constexpr int FOLDG_T = 512;                     // fallback kernel (R streamed through HBM)
constexpr int LDS_MAX_BYTES = 160 * 1024;        // gfx950: 160 KiB per workgroup
#ifndef MSCKF_SWEEP_NW
#define MSCKF_SWEEP_NW 8
#endif
#ifndef MSCKF_SWEEP_WPF
#define MSCKF_SWEEP_WPF 1
#endif
constexpr int LS_BIG_BATCH = 4000;               // from this many features on the 60-column leaves run twelve row blocks at a time
constexpr bool SWEEP_P2P = false;                // (round 2 measured progress words instead of the barrier per macro step: 159 vs 148 us at the root)
constexpr int SWEEP_NW = MSCKF_SWEEP_NW;         // concurrent folds of k_sweep
constexpr int SWEEP_WPF = MSCKF_SWEEP_WPF;       // wavefronts per fold (1 or 2)
// Group exchange: a shard's record holds one triangle slot per first clone slot; the slot is as wide as the sweep tile of
// the batch's mode (60 columns for k_sweep / k_wsweep<4>, 90 for k_wsweep<6>), whatever the shard's own tracks look like.
inline int xchg_w(int mode) { return mode == 2 ? WSweepGeom<6>::MAX_W : SWEEP_MAX_W; }
inline size_t xchg_slot(int mode) { return (size_t)xchg_w(mode) * (xchg_w(mode) + 1); }
constexpr int FOLD_LDS_BYTES = 160 * 1024 - 512;

inline int fold_class(int w) { return (w + 1 <= 64) ? 1 : (w + 1 <= 128) ? 2 : 3; }
// rows one register batch of a node with window width w holds
inline int fold_bmax(int w) {
    if (w > FOLD_RLDS_MAX_W) return (w + 1 <= 6 * 32) ? 16 * 12 : 16 * 6;     // fallback kernel tiles
    return FOLD_RL * (fold_class(w) == 3 ? FOLD_RPT_W3 : FOLD_RPT_BIG);
}

struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
    bool view = false;      // points into an arena: never freed or re-allocated on its own
};

inline void set_view(Buf& b, void* base, size_t off, size_t bytes) {
    b.p = static_cast<char*>(base) + off; b.bytes = bytes; b.view = true;
}

double now_us() {
    using namespace std::chrono;
    return duration_cast<duration<double, std::micro>>(steady_clock::now().time_since_epoch()).count();
}

// Host-side helpers of the drop-in call: the copies of the caller's arrays into the pinned upload image and the validation of
// the tracks are loops over F features that split cleanly over feature ranges (rounds 2-3 also gathered the tracks into sorted
// order here -- 29 us of the call at the headline, 155 at 10000 features; k_gather.h does that on the device now).  A few
// persistent worker threads (created with the context, CPU work only -- they never make a HIP call, whose first use costs a
// new thread ~100 ms) take chunks of a parallel_for next to the calling thread.  After a run they keep polling for
// MSCKF_HOST_SPIN_US (default 1000) microseconds before they sleep on the condition variable, so a filter calling at frame
// rate finds them awake (a condition-variable wake-up alone costs what the parallel loop saves at the headline size).
// MSCKF_HOST_THREADS=n sets the worker count (default 3, 0 = everything on the calling thread; 7 measured no faster: the phase
// is bound by the pinned image's memory traffic), MSCKF_HOST_PAR_MIN the smallest batch that is split (default 1024 features).
class HostPool {
public:
    explicit HostPool(int workers, int spin_us = 1000) : spin_us_(spin_us) {
        // the workers sit on the allowed CPUs next to the creating thread's (the context's one host thread): neighbours
        // share its L3 / NUMA node -- workers scattered over a two-socket box made the gather SLOWER than one thread
        std::vector<int> cpus;
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        const int me = sched_getcpu();
        if (me >= 0 && sched_getaffinity(0, sizeof(allowed), &allowed) == 0) {
            for (int k = 1; k < CPU_SETSIZE && (int)cpus.size() < workers; ++k) {
                const int cpu = (me + k) % CPU_SETSIZE;
                if (CPU_ISSET(cpu, &allowed)) cpus.push_back(cpu);
            }
        }
        pinned_ = (int)cpus.size() >= workers;       // else: no busy polling on CPUs the workers share with the caller
        for (int i = 0; i < workers; ++i) {
            const int cpu = i < (int)cpus.size() ? cpus[i] : -1;
            th_.emplace_back([this, cpu] {
                if (cpu >= 0) {
                    cpu_set_t one;
                    CPU_ZERO(&one);
                    CPU_SET(cpu, &one);
                    (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
                }
                worker();
            });
        }
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    int workers() const { return (int)th_.size(); }
    // fn(chunk) for chunk = 0 .. n - 1, on the workers and the calling thread; returns when every chunk is done
    void run(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        if (th_.empty() || n == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
        {
            std::unique_lock<std::mutex> lk(m_);
            while (inside_.load(std::memory_order_acquire) != 0) { lk.unlock(); std::this_thread::yield(); lk.lock(); }   // late leavers of the last run
            fn_ = &fn; n_ = n;
            next_.store(0, std::memory_order_relaxed);
            done_.store(0, std::memory_order_relaxed);
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        int c;
        while ((c = next_.fetch_add(1, std::memory_order_relaxed)) < n) { fn(c); done_.fetch_add(1, std::memory_order_release); }
        // (the workers are pinned, the caller is not: when the scheduler has moved it onto a worker's CPU, a caller that only
        //  spins here keeps that worker -- and the chunk it holds -- off the CPU for a whole timeslice: 8 ms calls, measured on
        //  the 5th - 10th call of a fresh process.  After a short spin the wait gives the CPU away.)
        for (int spins = 0; done_.load(std::memory_order_acquire) < n; ++spins) {
            if (spins < 256) cpu_relax(); else sched_yield();
        }
    }

private:
    static void cpu_relax() { __builtin_ia32_pause(); }
    void worker() {
        unsigned long long seen = 0;
        for (;;) {
            // spin for a while, then sleep
            bool got = false;
            const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(pinned_ ? spin_us_ : 0);
            for (int spins = 0;; ++spins) {
                if (gen_.load(std::memory_order_acquire) != seen) { got = true; break; }
                cpu_relax();
                if ((spins & 63) == 63) {
                    if (std::chrono::steady_clock::now() > t_end) break;
                    sched_yield();                  // (a caller that has been moved onto this CPU must not wait for the poll to end)
                }
            }
            const std::function<void(int)>* fn = nullptr;
            int n = 0;
            {
                std::unique_lock<std::mutex> lk(m_);
                if (!got) cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
                seen = gen_.load(std::memory_order_acquire);
                if (stop_) return;
                fn = fn_; n = n_;
                inside_.fetch_add(1, std::memory_order_acq_rel);
            }
            int c;
            while ((c = next_.fetch_add(1, std::memory_order_relaxed)) < n) { (*fn)(c); done_.fetch_add(1, std::memory_order_release); }
            inside_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<unsigned long long> gen_{0};
    std::atomic<int> next_{0}, done_{0}, inside_{0};
    const std::function<void(int)>* fn_ = nullptr;
    int n_ = 0;
    bool stop_ = false;
    bool pinned_ = false;
    int spin_us_ = 1000;

public:
    // memcpy split over the pool (the 300 KB covariance in and out of the pinned staging buffers)
    void copy(void* dst, const void* src, size_t bytes) {
        const int parts = workers() + 1;
        if (parts == 1 || bytes < (size_t)128 * 1024) { std::memcpy(dst, src, bytes); return; }
        const size_t piece = ((bytes / parts) + 4095) & ~(size_t)4095;
        const int n = (int)((bytes + piece - 1) / piece);
        const std::function<void(int)> fn = [&](int ch) {
            const size_t o = (size_t)ch * piece;
            std::memcpy(static_cast<char*>(dst) + o, static_cast<const char*>(src) + o, std::min(piece, bytes - o));
        };
        run(n, fn);
    }
};

// batches from this many features on split their host loops over the pool
inline int host_par_min() {
    static const int v = [] { const char* e = std::getenv("MSCKF_HOST_PAR_MIN"); return e ? std::max(1, std::atoi(e)) : 1024; }();
    return v;
}

}  // namespace

struct msckf_ctx {
    msckf_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[8]{};
    std::string last_error;
    double hp[12] = {0};                  // MSCKF_HOSTPROF=1: accumulated host-side phase times of msckf_update (us)
    long hp_calls = 0;
    // capacities
    int maxN = 0, maxF = 0, maxM = 0;
    // current problem
    int N = 0, d = 0, dc = 0, F = 0, sumM = 0, Mmax = 0;
    bool have_state = false, have_features = false, ran = false, ran_gain = false;
    bool acc_from_dev = false;            // the accepted total of a merged update sits in status word 2 (summed on the device)
    ncclComm_t comm = nullptr;            // RCCL communicator of the sharded update (msckf_comm_init)
    int comm_rank = 0, comm_world = 1;
    Buf dCommBuf;
    bool gain_blocked = false;            // the last K6 ran the two-block factorisation (second status word in use)
    double sigma = 0.0;
    double g[3]{}, Kinv[9]{};
    int n_chi2 = 0;
    // device buffers
    Buf dP, dPout, dCamR, dCamT, dCamR0, dCamT0, dChi2;
    Buf dViewPtr, dObsUV, dObsSlot, dBase, dMvec, dRho, dFmin, dBlkOff, dStack, dRank, dAcc, dGamma, dKeep;
    Buf dNodes, dRbuf, dStamps, dSweepNodes, dSweepFolds;
    Buf dPlanArena;                       // the plan's tables in one allocation (upload_plan): dNodes, dSweepNodes, dSweepFolds, dRootFlush, dFlush, dFlushOff are views
    Buf dLineBase, dLineDir, dLineConf, dLostFor, dTrackedFor, dSelFlags, dWorld;   // f1 (k_select)
    Buf dY, dS, dL, dU, dInvd, dK, dB2, dD, dPn, dDx, dCholWork, dStatus;
    // host-side plan
    std::vector<int> perm;                // sorted position -> input index
    std::vector<FoldNode> nodes;
    std::vector<std::pair<int, int>> levels;   // (node_base, count) per fold launch
    int root = -1;
    // band plan: leaves per first-slot group (k_fold), group merges and the root as k_sweep pipelines
    bool band_plan = false;
    std::vector<SweepNode> snodes;        // [0, n_group_merges) group merges, last = root
    std::vector<SweepFold> sfolds;
    int n_group_merges = 0;
    std::vector<std::pair<int, int>> sweep_levels;   // (node_base, count) per group-merge launch
    std::vector<int> sweep_level_nf;                 // fold slots (wavefronts) of that launch: SWEEP_NW or SWEEP_NW_BIG
    int sweep_mode = 0;                   // 0 k_sweep (60-column tiles, whole band R in LDS), 1 k_wsweep<4> (ring), 2 k_wsweep<6> (90-column tiles, ring)
    std::vector<int> h_flush, h_flush_off;            // k_wsweep: per sweep node the rows final at the head of every macro step
    Buf dFlush, dFlushOff;
    Buf dAssocUV, dAssocRes;              // f4 (k_assoc)
    Buf dFeatInfo;                        // k_lsweep: per sorted feature the block offset and the window-slot -> view map
    // plan cache: the K5 plan is a function of (N, exchange mode, the sorted tracks' first slot / last slot / view count);
    // a batch with the same key reuses the plan, its device tables and the zero pattern of the R workspace
    bool plan_valid = false;
    bool plan_no_wide = false;            // the cached plan was made by msckf_run_compress's re-plan
    int plan_N = -1;
    bool plan_xchg = false;
    std::vector<int> plan_fmin, plan_fmax, plan_view;
    long long stack_elems = 0;            // scalars of the current batch's stack blocks (a zero word follows them)
    int leaf_nf = 8;
    bool leaf_tall = false;               // 90-column leaves with 56-row blocks (k_lsweep<8, 6, LS_RS6T>)                      // row blocks in flight per 60-column leaf workgroup (8 or 12)
    bool leaf_narrow = false, leaf_wide = false;      // band plan: leaf nodes with w + 1 <= 64 / > 64 exist
    size_t root_off = 0;                  // offset (doubles) of the root block [T | r_n] in rbuf
    size_t zero_off = 0;                  // 16 doubles of the workspace no kernel writes: they read 0.0
    // group exchange (sharded band pipeline): the group triangles live in one export record at the head of rbuf
    bool xchg = false;                    // requested by msckf_set_group_exchange
    int xchg_span = 0;                    // longest track of the WHOLE batch in clone slots (msckf_set_exchange_span); 0: not told
    bool xchg_planned = false;            // the current plan has the record layout
    std::vector<double> h_xflags;         // [N] 1.0 where this shard has tracks starting at the slot
    std::vector<double> x_key;            // flags of the last merged records (plan cache of msckf_run_merge_groups)
    int x_nrec = 0;
    bool x_plan_valid = false;
    std::vector<SweepNode> x_snodes;      // rank-0 merge plan: cross-rank group merges, last = root
    std::vector<SweepFold> x_sfolds;
    int x_n_merges = 0;
    size_t x_root_off = 0, x_zero_off = 0;
    long long x_rec_base = 0;             // where the records lay (offset from the workspace base) when the plan was made
    size_t rbuf_doubles = 0;              // used by the plan
    size_t gather_off = 0;                // region for gathered shard blocks
    int gather_cap = 0;
    int n_leaves = 0;
    int n_leaves0 = 0;                    // band plan: the leaves [0, n_leaves0) fold short tracks only (none of a split track's blocks)
    int acc_override = -1;                // total accepted over all shards (merge_gain path)
    float us_host_prep = 0, us_h2d = 0, us_d2h = 0;
    float us_stage[3] = {0, 0, 0};
    float us_total = 0;
    std::vector<int> h_view_sorted;
    std::vector<int> h_view_in;           // view_ptr as the caller gave it (set_tracks permutes with it)
    std::vector<int> h_fmin, h_fmax;      // first / last slot per sorted feature (msckf_replan)
    bool have_tracks = false, use_select = false;
    msckf_select_params sel_params{};     // of the last msckf_run_select
    std::vector<double> h_cam[4];         // host mirror of cam_R / cam_t / cam_R0 / cam_t0 (clone bookkeeping, f2)
    // arenas: what travels together lives together, so each direction is ONE copy through pinned memory
    // (a pageable hipMemcpyAsync costs ~9 us apiece; the 15 + 6 of them were 270 us of the host-inclusive call)
    Buf dPoseArena, dFeatArena, dRawArena, dResArena, dGateArena;     // dRawArena: the tracks in the caller's order (k_gather.h)
    void *hPose = nullptr, *hFeat = nullptr, *hRes = nullptr, *hGate = nullptr, *hP = nullptr;   // pinned staging
    size_t hFeatCap = 0, hGateCap = 0, res_dx_off = 0, res_p_off = 0;
    size_t res_mask_off = 0, res_mask_cap = 0, res_cap = 0;   // gate bytes of the whole (sharded) batch behind P_out; arena bytes
    // sharded update: the gate results ride with the exchange (msckf_set_exchange_mask)
    std::vector<int> x_bounds;            // [n_shards + 1] shard r holds features [b[r], b[r+1]) of the whole batch; empty = off
    int xmask_doubles = 0;                // doubles of the record head that carry the shard's gate bytes (input order)
    Buf dPerm;                            // sorted position -> input index, on the device (view into the feature arena)
    std::vector<double> chi2_cache;
    bool defer_state_sync = false;        // msckf_update: the feature upload's sync covers the state upload
    bool oneshot = false;                 // msckf_update: set_features uploads, launches K1-K4 and plans K5 meanwhile, no sync
    bool feature_launched = false;        // K1-K4 of the current batch is already in the stream (oneshot)
    HostPool* pool = nullptr;             // host worker threads for the pack loops (CPU work only)
    // K6-K7 beside the root sweep (k_gstream.h, k_root_gain): the sequential block update polls the rows the sweep's flusher publishes
    Buf dGsEx, dGsFlag, dGsProg;          // exchange tiles [nb][ns][256], their flags, the sweep's progress word
    Buf dRootFlush, dXRootFlush;          // flush tables of the root sweeps (rows final at the head of every macro step): local plan, merge plan
    std::vector<int> h_root_flush;        // ... of the local plan's root (band plan, k_sweep form)
    std::vector<int> x_root_flush;        // ... of the merge plan's root (msckf_run_merge_groups)
    unsigned gs_epoch = 0;                // tag of the current launch pair in the progress word and the exchange flags
    bool gs_enabled = true;               // MSCKF_GAIN_STREAM=0: the round-3 K6-K7 (separate launches behind the root sweep)
    bool gs_overlap = true;               // MSCKF_GAIN_OVERLAP=0: k_gain_stream as a launch of its own behind the root sweep
    int root_band = 0;                    // widest row of the root block in columns (the local plan's / the merge plan's)
    bool gs_stamp = false;                // msckf_run_timed: k_root_gain notes when its sweep ends and when its update ends
    bool gs_fused_last = false;           // the last pipeline ran k_root_gain (stage events cannot split it)
    // tracks that span more than WIDE_SPAN clone slots are sorted behind the others ([0, Fb) short, [Fb, F) long) and split
    int Fb = 0, Fw = 0, Fw1 = 0, Mmax_band = 0, Mmax_wide = 0, Mmax_w1 = 0;   // (Fw1 of the Fw wide tracks have <= 15 views)
    // Round 5: a long track (more than WIDE_SPAN clone slots) is SPLIT (k_feature.h, two-level nullspace basis): the sorted
    // arrays hold its blocks behind the F tracks -- [F, F + nNarrow) the narrow blocks of its view groups (ordinary <= 10-slot
    // tracks to every K5 kernel, sorted by (first slot, last slot) among themselves), [F + nNarrow, Fs) one remainder block per
    // long track (3 (groups - 1) rows that touch every slot of the track).  The long tracks themselves, [Fb, F), are entries
    // of K1-K3 only (gate, mask, counters): no K5 plan names them.
    int Fs = 0, nNarrow = 0;              // entries of the sorted arrays; narrow blocks
    int sumMs = 0;                        // views of all entries
    bool split_on = false;                // this batch's long tracks were split (Fw > 0: k_feature<64, true> writes their blocks)
    std::vector<SplitRec> h_split;        // per long track, in sorted order
    std::vector<int> h_parent;            // [Fs - F] sorted index of the long track a block belongs to
    Buf dSplit, dRem;
    int rem_cap = 0;                      // rows the remainder blocks may hold in all (3 per view group)
    // remainder rows (capacity: 3 per view group) up to which K6-K7 takes them as they are (msckf_debug_set_rem_direct_rows).  Measured
    // on tracks ~ U[2, 30] at N = 30, as they are against their own (cut) merge tree: 450 tracks 777 / 971 us, 600 (3100 rows)
    // 961 / 1056, 800 (4250 rows) 1221 / 1185, 1000 1486 / 1286
    static constexpr int REM_DIRECT_DEFAULT = 3840;
    int rem_leaf_rows = 0;                // rows of a leaf of the remainder rows' merge tree (MSCKF_REM_LEAF_ROWS; 0: one register batch of k_fold)
    int rem_direct_max = REM_DIRECT_DEFAULT, rem_direct_max_wide = 16 * GS_MAX_NB2;
    bool in_merge = false;                // a merge of gathered shard blocks is being launched: its K6-K7 has ONE source of rows, whatever the
                                          //   rank's own last batch looked like
    bool t2_early = false;                // this run: the dense remainder rows were applied by a launch of their own (launch_gain_t2_early); the update
                                          //   on the band root starts from that launch's P_out / dx
    bool retry_plain = false;             // a K6-K7 launch timed out once: the context runs without in-launch waits since (msckf_get_result)
    bool fake_timeout_done = false;       // MSCKF_DEBUG_FAKE_TIMEOUT
    bool rem_direct = false;              // ... few enough (3840; 16384 on windows of more than 31 clones): K6-K7 takes them as they are (k_rem_scatter), no QR of their own
    // ... the remainder blocks' own QR: a merge tree beside the band pipeline (second stream), its root the second source of
    // rows for K6-K7 -- applied by a k_gain_stream launch of its own behind the first one, ordered by an event
    std::vector<FoldNode> rnodes;
    std::vector<std::pair<int, int>> rlevels;
    int rroot = -1; size_t rroot_off = 0;
    // ... or, the tree cut where a further level would remove fewer rows than K6-K7 takes in the launch's time: the triangles it
    // ends with (k_tri_gather lays them down in dRem as dense rows for a k_gain_stream launch behind the first update)
    std::vector<int> rtops;
    int rtop_rows = 0;
    int rem_cut_rows = -1;                // MSCKF_REM_CUT_ROWS: rows a level must remove to be worth its launch (-1: 600 with R in LDS, else never; 0: never)
    Buf dRNodes;
    hipEvent_t ev_rem = nullptr;
    bool wide_active = false;             // the current plan keeps the wide tracks out of the band pipeline / tree
    bool no_wide = false;                 // the batch was re-planned with every track in one plan (msckf_run_compress: the
                                          // exported block must hold the wide tracks' rows too)
    hipStream_t stream2 = nullptr;        // the long tracks' kernels (k_feature<64, true>, k_rem_scatter, the remainder blocks' tree) run beside the band pipeline
    hipEvent_t ev_fork = nullptr, ev_wfeat = nullptr;   // uploads done -> stream2 may start; the wide tracks' K4 blocks are written
    bool wide_on_stream2 = false;         // this batch's wide k_feature went to stream2 (ev_wfeat pending)
    // The one-shot call's K5 plan (a memset of the workspace + four to six tables of a few KB, each a ~5 us blit kernel) goes up on
    // a stream of its own beside K1-K4; the main stream waits for ev_plan in front of K5.  (In the main stream the copies sat
    // between k_feature and k_lsweep: 25 us of the call at the headline, rocprofv3 --memory-copy-trace.)
    hipStream_t stream_up = nullptr;
    hipEvent_t ev_plan = nullptr;
    hipStream_t plan_stream = nullptr;    // where upload_plan puts its copies (stream, or stream_up in the one-shot call)
    void* hPlan = nullptr; size_t hPlanCap = 0;   // pinned staging image of the plan tables
    hipEvent_t ev_plan_up = nullptr; bool plan_staged = false;   // ... its copies are through
    // The one-shot call's P and poses go up on stream_up too, beside the tracks; the main stream waits for ev_state in front of K1.
    hipEvent_t ev_state = nullptr;
    bool state_pending = false;           // ev_state recorded, the main stream has not been told to wait for it yet
    // The one-shot call's results are written to the pinned host buffers by the kernels themselves (k_feature: rank / accepted,
    // the fused K6-K7 kernels: status, dx, P+) next to the HBM copies: the two copy commands behind K7 (and the ~10 us each of
    // them waits behind its predecessor) are gone, the host reads hGate / hRes when the stream has drained.
    bool want_direct = false;             // msckf_update is running (and MSCKF_DIRECT_RESULT is not 0)
    bool gate_direct = false;             // this batch's k_feature mirrored its results into hGate
    bool res_direct = false;              // this run's K6-K7 mirrored status | dx | P+ into hRes
    long direct_serial = -1;              // ... the run it did so for (run_serial)
    // ... and the gate results are on the host when K1-K4 have ended (ev_gate): msckf_update sums them and fills the caller's mask
    // while K5-K7 run, msckf_get_result finds the sums here
    hipEvent_t ev_gate = nullptr;
    bool gate_event = false;              // ev_gate sits behind this batch's k_feature
    long gate_serial = -1;                // run whose gate sums are in gate_cnt (and whose mask the caller already has)
    int gate_cnt[4] = {0, 0, 0, 0};
    const uint8_t* gate_mask_dst = nullptr;
    bool direct_enabled = true;
    // Resident calls that leave their work in the stream without waiting for it (msckf_set_features, msckf_set_poses,
    // msckf_commit_covariance): the next call that rewrites a pinned staging buffer, or that uses the side stream, waits first.
    // The last level of group merges inside k_root_gain's launch, the root taking their rows as they are published (k_gstream.h):
    bool root_streamed = false;           // this plan's root folds name their producers (SweepFold::prod), first fold not adopted
    int stream_level = -1;                // index of that level in sweep_levels
    std::vector<int> h_mflush;            // [n offsets | the nodes' flush tables]
    int mflush_at = 0;                    // where h_mflush sits in the uploaded h_root_flush
    bool x_streamed = false; int x_root_n_gate = -1, x_mflush_at = 0;   // the same for rank 0's merge plan (run_merge_groups)
    int root_n_gate = -1;                 // step-0 requirements behind the root's flush + gate tables (sweep_gate_table), -1: no gate table
    Buf dMFlush, dMProg;                  // ... on the device; the merge nodes' progress words (64)
    bool stream_enabled = true;           // MSCKF_ROOT_STREAM=0: the level keeps its own launch
    int n_cu = 256;                       // compute units of the device: workgroups that wait for each other inside one launch must all be resident
    bool feat_busy = false, pose_busy = false;   // hFeat / hPose may still be read by a copy or by k_gather
    bool main_busy = false;                      // the main stream holds work nobody has waited for
    bool run_pending = false;                    // ... a pipeline / merge among it (kernels that read the K5 plan and the workspace)
    bool wide_concurrent = true;          // MSCKF_WIDE_STREAM=0: everything on one stream
    long run_serial = 0;                  // bumped by every pipeline / merge launch
    long fetched_serial = -1;             // the run whose return code msckf_get_result derived last ...
    int fetched_rc = 0;                   // ... and that code: msckf_commit_covariance need not read the gate results again
};

namespace {

#define HIPCHK(ctx, call)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_);              \
            return MSCKF_ERR_HIP;                                                               \
        }                                                                                       \
    } while (0)

// ---- librccl, loaded on demand (the sharded update's one exchange; a single-GPU process never loads it) ----
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
RcclApi& rccl() {
    static RcclApi api;
    if (api.lib || !api.err.empty()) return api;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (api.lib) break; }
    if (!api.lib) { api.err = std::string("dlopen librccl: ") + dlerror(); return api; }
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(api.lib, n); if (!p) { ok = false; api.err = std::string("librccl has no ") + n; } return p; };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.Gather = reinterpret_cast<decltype(api.Gather)>(sym("ncclGather"));
    api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { dlclose(api.lib); api.lib = nullptr; }
    return api;
}
#define NCCLCHK(ctx, call)                                                                      \
    do {                                                                                        \
        ncclResult_t r_ = (call);                                                               \
        if (r_ != ncclSuccess) {                                                                \
            (ctx)->last_error = std::string(#call) + ": " + rccl().GetErrorString(r_);          \
            return MSCKF_ERR_COMM;                                                              \
        }                                                                                       \
    } while (0)



int ensure(msckf_ctx* c, Buf& b, size_t bytes, bool zero = false) {
    if (b.bytes >= bytes && b.p) return MSCKF_OK;
    if (b.view) { b.p = nullptr; b.bytes = 0; b.view = false; }       // (a view into an arena that is too small: a buffer of its own)
    if (b.p) HIPCHK(c, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    size_t want = std::max<size_t>(bytes, 256);
    HIPCHK(c, hipMalloc(&b.p, want));
    b.bytes = want;
    if (zero) HIPCHK(c, hipMemset(b.p, 0, want));
    return MSCKF_OK;
}

template <typename Tp>
Tp* ptr(const Buf& b) { return reinterpret_cast<Tp*>(b.p); }

// clone poses: host mirror -> pinned arena image -> HBM, one copy
int upload_poses(msckf_ctx* c, hipStream_t st = nullptr) {
    if (!st) st = c->stream;
    const size_t N = c->h_cam[1].size() / 3, mN = c->maxN;
    if (N == 0) return MSCKF_OK;
    if (c->pose_busy) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream_up)); c->pose_busy = false; }
    double* h = static_cast<double*>(c->hPose);
    std::memcpy(h, c->h_cam[0].data(), N * 72);
    std::memcpy(h + 9 * mN, c->h_cam[1].data(), N * 24);
    std::memcpy(h + 12 * mN, c->h_cam[2].data(), N * 72);
    std::memcpy(h + 21 * mN, c->h_cam[3].data(), N * 24);
    HIPCHK(c, hipMemcpyAsync(c->dPoseArena.p, h, 24 * mN * 8, hipMemcpyHostToDevice, st));
    c->pose_busy = true;
    return MSCKF_OK;
}

// status (64 B) | dx (d) | P_out (d x d) | gate bytes of the whole batch: ONE contiguous range for the CURRENT d, so
// that one collective / one copy moves a complete result (re-seated whenever d changes)
void seat_result_views(msckf_ctx* c) {
    const size_t d = c->d;
    c->res_dx_off = 64; c->res_p_off = 64 + d * 8; c->res_mask_off = c->res_p_off + d * d * 8;
    set_view(c->dStatus, c->dResArena.p, 0, 64);
    set_view(c->dDx, c->dResArena.p, c->res_dx_off, d * 8);
    set_view(c->dPout, c->dResArena.p, c->res_p_off, d * d * 8);
}

// doubles in front of the group triangles of an export record: N flags | accepted count | gate bytes
inline size_t rec_head(const msckf_ctx* c) { return (size_t)c->N + 1 + (size_t)c->xmask_doubles; }
int sweep_mode_for(const msckf_ctx* c, int N, int max_span);
// sweep mode of the group exchange at N clones: that of the whole batch's longest track when the caller told it
// (every rank then lays its record out alike), else the 60-column k_sweep form only (-1: no group exchange)
inline int xchg_mode(const msckf_ctx* c, int N) {
    if (c->xchg_span > 0) return sweep_mode_for(c, N, c->xchg_span);
    return sweep_mode_for(c, N, 1) == 0 ? 0 : -1;
}
inline size_t rec_slot(const msckf_ctx* c) { const int m = xchg_mode(c, c->N); return xchg_slot(m < 0 ? 0 : m); }

// the clone set changed: feature batch, plan and results of the old layout are void
void invalidate_batch(msckf_ctx* c) {
    c->have_features = false;
    c->have_tracks = false;
    c->use_select = false;
    c->ran = false;
    c->F = 0;
}

// Tracks that span more than WIDE_SPAN clone slots are SPLIT (k_feature.h: two-level nullspace basis) wherever a batch is
// planned for this context alone: their narrow blocks are ordinary tracks of the 60-column band pipeline, their remainder
// blocks a small dense QR beside it.  Any window size, both dtypes.  Not with the group exchange of a sharded update (its
// record layout is the band pipeline's; such a batch keeps one plan for every track) nor where a plan is forced.
constexpr int WIDE_SPAN = SPLIT_GSLOTS;
// the sequential block update (k_gstream.h) on DENSE rows at this window size: strips and LDS
bool gstream_ok_dc(const msckf_ctx* c, int dc) {
    if (!c->gs_enabled || dc < 1) return false;
    const int nb = (dc + 15) / 16, ns = nb + 1;
    return ns <= GS_MAX_NS && gstream_lds_doubles(ns, nb) * 8 <= (size_t)(LDS_MAX_BYTES - 1024);
}
bool split_ok(const msckf_ctx* c, int N) {
    if ((c->cfg.flags & (MSCKF_FLAG_TREE_PLAN | MSCKF_FLAG_BAND_ONLY)) || c->xchg) return false;
    static const bool off = [] { const char* e = std::getenv("MSCKF_SPLIT"); return e && std::atoi(e) == 0; }();
    return !off && N > WIDE_SPAN && gstream_ok_dc(c, 6 * N);
}
struct Run { int b, e; };                 // entries [b, e) of the sorted arrays

// ---- QR tree plan ---------------------------------------------------------
// Leaves: consecutive sorted features whose stacked-row bound stays under
// leaf_rows.  Merge levels: consecutive nodes, up to `arity` children as long
// as the rows to fold fit one LDS batch of the parent window, else two.
void build_tree(msckf_ctx* c, const std::vector<int>& fmin, const std::vector<int>& fmax,
                const std::vector<int>& view_sorted, const std::vector<unsigned char>* valid, const std::vector<Run>& runs,
                size_t off0, std::vector<FoldNode>& nodes, std::vector<std::pair<int, int>>& levels, int& root, size_t& root_off,
                size_t& off_end, int& n_leaves, int leaf_rows_default = 160, bool tall_merges = false, int cut_rows = 0,
                std::vector<int>* tops = nullptr) {
    const int N = c->N;
    const int leaf_rows = c->cfg.leaf_rows > 0 ? c->cfg.leaf_rows : leaf_rows_default;
    const int arity = c->cfg.merge_arity > 0 ? c->cfg.merge_arity : 6;
    const int rem0 = c->split_on ? c->F + c->nNarrow : (1 << 30);        // remainder blocks: entries [rem0, Fs)
    nodes.clear();
    levels.clear();
    size_t off = off0;
    auto push = [&](int kind, int b, int e, int lo, int hi) {
        FoldNode n{};
        n.kind = kind; n.src_begin = b; n.src_end = e; n.win_lo = lo; n.w = 6 * (hi - lo + 1); n.pad = 0;
        n.out_off = (long long)off;
        off += (size_t)n.w * (n.w + 1);
        nodes.push_back(n);
    };
    // leaves (with `valid`, the features k_select masked out carry no rows: they neither count
    // towards a leaf nor widen its window, and stretches without a valid feature get no leaf)
    auto live = [&](int i) { return !valid || ((*valid)[i] & 1); };
    // (a leaf's rows must come sorted by their first column and every run is sorted on its own: no leaf across a run
    //  boundary; the merge levels sort their rows themselves)
    for (const Run& run : runs) {
        int f = run.b;
        const int F = run.e;
        while (f < F) {
            while (f < F && !live(f)) ++f;
            if (f >= F) break;
            int lo = fmin[f], hi = fmax[f];
            int rows = 0, e = f, last = f;
            while (e < F && (e - f) < FOLD_MAX_SRC) {
                if (live(e)) {
                    // (rows of the entry: at most 2 per view; a remainder block holds 3 (groups - 1) of them unless the track's geometry
                    //  is degenerate -- an estimate is all a leaf's size needs, k_fold takes what it finds in register batches of 256)
                    const int r = e >= rem0 ? 3 * ((int)c->h_split[c->h_parent[e - c->F] - c->Fb].ng - 1) : 2 * (view_sorted[e + 1] - view_sorted[e]);
                    if (e > f && rows + r > leaf_rows) break;
                    rows += r;
                    lo = std::min(lo, fmin[e]);
                    hi = std::max(hi, fmax[e]);
                    last = e;
                }
                ++e;
            }
            push(0, f, last + 1, lo, hi);
            f = last + 1;
        }
    }
    n_leaves = (int)nodes.size();
    if (n_leaves > 0) levels.push_back({0, n_leaves});
    // merge levels
    int lvl_base = 0, lvl_cnt = n_leaves;
    while (lvl_cnt > 1 || (lvl_cnt == 1 && (nodes[lvl_base].win_lo != 0 || nodes[lvl_base].w != 6 * N))) {
        const int nb = (int)nodes.size();
        const int end = lvl_base + lvl_cnt;
        struct Grp { int b, e, lo, hi; };
        std::vector<Grp> grps;
        int i = lvl_base;
        while (i < end) {
            int lo = nodes[i].win_lo, hi = lo + nodes[i].w / 6 - 1;
            int e = i + 1;
            int fold_rows = 0;
            while (e < end && (e - i) < arity) {
                const int lo2 = std::min(lo, nodes[e].win_lo);
                const int hi2 = std::max(hi, nodes[e].win_lo + nodes[e].w / 6 - 1);
                const int cap = fold_bmax(6 * (hi2 - lo2 + 1));
                // keep one register batch per node -- unless the tree is a chain of dense levels anyway (the remainder blocks'
                // tree: a level costs ~270 us whatever it folds, a further batch of sorted rows only the columns right of its
                // first row's leading one)
                if (!tall_merges && (e - i) >= 2 && fold_rows + nodes[e].w > cap) break;
                fold_rows += nodes[e].w;
                lo = lo2; hi = hi2;
                ++e;
            }
            grps.push_back({i, e, lo, hi});
            i = e;
        }
        if (grps.size() == 1) { grps[0].lo = 0; grps[0].hi = N - 1; }   // the root spans every clone
        if (cut_rows > 0 && tops && lvl_cnt <= TRI_GATHER_MAX) {
            // a level that removes fewer than cut_rows rows is not worth its launch: the tree ends with this level's triangles
            int rows_in = 0, rows_out = 0;
            for (int k = lvl_base; k < end; ++k) rows_in += nodes[k].w;
            for (const Grp& g : grps) rows_out += 6 * (g.hi - g.lo + 1);
            if (rows_in - rows_out < cut_rows) {
                for (int k = lvl_base; k < end; ++k) tops->push_back(k);
                root = -1; root_off = 0; off_end = off;
                return;
            }
        }
        for (const Grp& g : grps) push(1, g.b, g.e, g.lo, g.hi);
        lvl_base = nb;
        lvl_cnt = (int)nodes.size() - nb;
        levels.push_back({lvl_base, lvl_cnt});
    }
    root = nodes.empty() ? -1 : (int)nodes.size() - 1;
    root_off = nodes.empty() ? 0 : (size_t)nodes.back().out_off;
    off_end = off;
}

// ---- band plan ----------------------------------------------------------------
// Tracks span at most SWEEP_MAX_W / 6 clone slots: the stacked system is a band matrix.
//   level 0 : leaves never cross a first-slot group (k_fold; every leaf window starts at the group's slot);
//   level 1 : one k_sweep workgroup per group with several leaves folds the leaf triangles (same first
//             column -> pipeline lag 1) into the group's triangle;
//   level 2 : ONE k_sweep workgroup folds the group triangles (first columns 6 slots apart -> lag 7)
//             into the band R = the root block [T | r_n].
// Returns false when the batch does not qualify (wide tracks, R band over the LDS budget): tree plan then.
constexpr int SWEEP_NW_BIG = 12;                 // k_sweep group merges of more than SWEEP_NW + 1 triangles: twelve fold slots, one round
constexpr int SWEEP_NW_MID = 11;                 // ... of up to 12 triangles: eleven, so that the level fits k_root_gain_m's launch (twelve wavefronts with the flusher)
void sweep_schedule(std::vector<SweepFold>& folds, int begin, int end, int* nsteps, int nf = SWEEP_NW, bool adopt = true) {
    int last = 0;
    if (end > begin && adopt) folds[begin].t0 = 0;         // adopted: copied into the empty R, no elimination steps
    const int first = adopt ? begin + 1 : begin;           // (not adopted: a streamed first triangle is folded like the others)
    for (int g = first; g < end; ++g) {
        int t0 = 1;                                        // step t0 - 1 publishes the fold's first column
        if (g > first) t0 = folds[g - 1].t0 + (folds[g].off - folds[g - 1].off) + 1;
        // (a fold runs one step per column of its ENVELOPE ew >= w: where R already reaches further right than the
        //  source triangle, the tile's rows fill in there and the fill has to be eliminated as well)
        if (g - first >= nf) t0 = std::max(t0, folds[g - nf].t0 + folds[g - nf].ew + 1);
        folds[g].t0 = t0;
        last = std::max(last, t0 + folds[g].ew);
    }
    *nsteps = last;
}

// Which sweep kernel runs the group merges and the root of a batch of N clones whose longest track spans
// `max_span` clone slots: 0 = k_sweep (tiles of 60 columns, the whole band R in LDS), 1 = k_wsweep<4> (the
// same tiles, R in a ring: any N), 2 = k_wsweep<6> (tiles of 90 columns, ring), -1 = none: merge tree.
constexpr int WS_RC_LOG2_4 = 8;            // ring rows of k_wsweep<4>: 256 x 64 doubles
constexpr int WS_RC_LOG2_6 = 7;            // ring rows of k_wsweep<6>: 128 x 96 doubles
int sweep_mode_for(const msckf_ctx* c, int N, int max_span) {
    if (c->cfg.flags & MSCKF_FLAG_TREE_PLAN) return -1;                   // tree plan forced
    if (N < 1) return -1;
    if (6 * max_span <= SWEEP_MAX_W) {
        if (sweep_lds_bytes(6 * N, SWEEP_NW, SWEEP_WPF) <= (size_t)FOLD_LDS_BYTES) return 0;
        return 1;
    }
    if (6 * max_span <= WSweepGeom<6>::MAX_W) return 2;
    return -1;
}
// THE rule for the group exchange of the sharded band pipeline (msckf_band_rule exports it): the record
// layout and the merging rank's sweeps are those of k_sweep.
bool band_rule(const msckf_ctx* c, int N, int max_span) { return sweep_mode_for(c, N, max_span) >= 0; }

// k_wsweep: rows of R no present or future fold step touches at the head of macro step t (the schedule is
// static).  Entry t = lo | n << 16: rows [lo, lo + n) leave the ring at the head of step t; entry nsteps covers
// the rest.  Returns false when some step would touch a row whose ring slot still holds an unflushed row.
bool sweep_flush_table(const std::vector<SweepFold>& folds, int begin, int end, int nsteps, int wtot, int rc,
                       std::vector<int>& tab) {
    const int first = (end > begin && folds[begin].t0 == 0) ? begin + 1 : begin;   // an adopted triangle runs no step
    int lprev = 0;
    bool ok = true;
    for (int t = 0; t <= nsteps; ++t) {
        int L = wtot, H = -1;
        if (t < nsteps) {
            for (int g = first; g < end; ++g) {
                const SweepFold& f = folds[g];
                if (t >= f.t0 + f.ew) continue;                           // finished (one step per envelope column)
                const int row = f.off + std::max(0, t - f.t0);            // its present (or first) pivot row
                L = std::min(L, row);
                if (t >= f.t0) H = std::max(H, row);
            }
        }
        L = std::max(L, lprev);
        if (H >= lprev + rc) ok = false;                                  // a touched row aliases one flushed in this step
        if (t == 0 && first > begin && folds[begin].off + folds[begin].w > rc) ok = false;   // the adopted rows fit the ring
        tab.push_back(lprev | ((L - lprev) << 16));
        lprev = L;
    }
    return ok;
}

// Streamed sources of a node (SweepFold::prod, k_root_gain's merge workgroups): which rows of which producer the fold
// wavefronts fetch at the head of which macro step -- k_sweep.h fetches rows [0, 8) of a fold's source before step 0 (the first
// nf folds) or in chunk max((ew' - 1) / 8 - 1, 0) of the fold that has the slot before it, and rows [8 KK + 8, 8 KK + 16) at
// the head of the fold's chunk KK.  Appended to `tab`: nsteps + 2 step entries (up to two requirements prod << 6 | rows, 12 bits
// each; a third moves to an earlier step, which only asks for it sooner) | the requirements of step 0.  Returns their count.
int sweep_gate_table(const std::vector<SweepFold>& folds, int begin, int end, int nsteps, int nf, std::vector<int>& tab) {
    const int first = (end > begin && folds[begin].t0 == 0) ? begin + 1 : begin;
    // (step, requirement) pairs, then a counting sort by step: no per-step containers on the one-shot call's host path
    static thread_local std::vector<std::pair<int, int>> req;
    req.clear();
    for (int i = first; i < end; ++i) {
        const SweepFold& f = folds[i];
        if (f.prod <= 0) continue;
        int step = 0;
        if (i - first >= nf) { const SweepFold& q = folds[i - nf]; step = q.t0 + 8 * std::max((q.ew - 1) / 8 - 1, 0); }
        req.push_back({std::min(step, nsteps), ((f.prod - 1) << 6) | std::min(f.w, 8)});
        for (int kk = 0; kk < 8 && 8 * kk < f.ew; ++kk)
            if (8 * kk + 8 < f.w) req.push_back({std::min(f.t0 + 8 * kk, nsteps), ((f.prod - 1) << 6) | std::min(f.w, 8 * kk + 16)});
    }
    std::sort(req.begin(), req.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first > y.first; });   // latest step first
    const size_t base = tab.size();
    tab.resize(base + nsteps + 2, 0);
    std::vector<int> step0;
    int carry[64], ncarry = 0;                      // requirements pushed to an earlier step (a step takes two)
    size_t k = 0;
    for (int t = nsteps + 1; t >= 0; --t) {
        int mine[2], n = 0;
        auto take = [&](int r) { if (t == 0) step0.push_back(r); else if (n < 2) mine[n++] = r; else if (ncarry < 64) carry[ncarry++] = r; else step0.push_back(r); };
        const int nc = ncarry; ncarry = 0;
        int prev[64];
        for (int j = 0; j < nc; ++j) prev[j] = carry[j];
        for (int j = 0; j < nc; ++j) take(prev[j]);
        while (k < req.size() && req[k].first == t) take(req[k++].second);
        if (t > 0) tab[base + t] = (n > 0 ? mine[0] : 0) | (n > 1 ? mine[1] << 12 : 0);
    }
    tab.insert(tab.end(), step0.begin(), step0.end());
    return (int)step0.size();
}

// k_wsweep with PUB: what wavefront 0 publishes at the head of macro step t -- the rows that were final WS_PUB_LAG + 1 steps
// earlier, where that count passes a boundary of k_gstream.h's 16-row blocks (they end at rows = wtot mod 16); 0: nothing.
// Appended behind the node's flush entries (tab[off .. off + nsteps]).
void sweep_publish_table(std::vector<int>& tab, size_t off, int nsteps, int wtot) {
    const int boff = (16 - (wtot & 15)) & 15;
    int published = 0;
    for (int t = 0; t <= nsteps; ++t) {
        int pr = 0;
        if (t > WS_PUB_LAG) {
            const int e = tab[off + t - WS_PUB_LAG - 1];
            const int rows = (e & 0xFFFF) + (e >> 16);
            if (((rows + boff) >> 4) > ((published + boff) >> 4)) { pr = rows; published = rows; }
        }
        tab.push_back(pr);
    }
}

bool build_plan_band(msckf_ctx* c, const std::vector<int>& fmin, const std::vector<int>& fmax,
                     const std::vector<int>& view_sorted, const std::vector<unsigned char>* valid, const std::vector<Run>& runs) {
    const int N = c->N, dc = 6 * N;
    auto live = [&](int i) { return !valid || ((*valid)[i] & 1); };
    int max_span = 0, mm_live = 0, F = 0;                 // F: entries of the runs (they size the leaves)
    for (const Run& run : runs) {
        F += run.e - run.b;
        for (int f = run.b; f < run.e; ++f)
            if (live(f)) { max_span = std::max(max_span, fmax[f] - fmin[f] + 1); mm_live = std::max(mm_live, view_sorted[f + 1] - view_sorted[f]); }
    }
    if (c->xchg && c->xchg_span > 0) {
        if (max_span > c->xchg_span) return false;                        // the caller's figure does not cover this shard: root blocks
        max_span = c->xchg_span;                                          // every shard plans with the mode of the whole batch
    }
    const int mode = sweep_mode_for(c, N, max_span);
    if (mode < 0) return false;
    if (c->xchg && c->xchg_span == 0 && mode != 0) return false;          // (not told the batch's span: the 60-column k_sweep form only)
    c->sweep_mode = mode;
    const size_t XCHG_SLOT = xchg_slot(mode);
    const int XW = xchg_w(mode);
    // Leaf nodes (k_lsweep): one workgroup folds NF row blocks at a time, a block holds `fpb` features, so a node
    // gets a multiple of NF * fpb features: enough nodes to fill the chip once, at most 128 features each.
    // cfg.leaf_rows > 0 (tests) cuts the leaves by stacked rows instead.
    const int leaf_rows = c->cfg.leaf_rows > 0 ? c->cfg.leaf_rows : (1 << 30);
    int leaf_feats = 128;
    {
        const bool wide_leaf = 6 * max_span + 1 > 64;
        static const int tall_mode = [] { const char* e = std::getenv("MSCKF_LS_TALL"); return e ? atoi(e) : 1; }();
        static const int big_batch = [] { const char* e = std::getenv("MSCKF_LS_BIG_BATCH"); return e ? atoi(e) : LS_BIG_BATCH; }();
        // (the same for the 60-column leaves -- three tracks per block, eight wavefronts, prefetched tile -- is slower than twelve
        //  wavefronts of two: 142 against 116 us at 10000 features)
        c->leaf_tall = wide_leaf && tall_mode != 0;
        const int rb = wide_leaf ? (c->leaf_tall ? LSweepGeom<6, LS_RS6T>::RB : LSweepGeom<6, LS_RS6>::RB) : LSweepGeom<4, LS_RS4>::RB;
        const int mm = std::max(mm_live, 1);                                     // longest track the leaves will see
        const int fpb = std::max(1, std::min(LS_FB, rb / std::max(2 * mm - 3, 1)));
        c->leaf_nf = (F >= big_batch && !c->leaf_tall) ? 12 : 8;
        const int unit = c->leaf_nf * fpb;
        static const int leaf_target = [] { const char* e = std::getenv("MSCKF_LEAF_TARGET"); return e ? std::max(1, atoi(e)) : 240; }();
        int want = (F + leaf_target - 1) / leaf_target;
        want = ((want + unit - 1) / unit) * unit;
        leaf_feats = std::max(std::min(unit, 128), std::min(want, 128));
        if (c->cfg.leaf_rows > 0) leaf_feats = 128;
    }
    c->leaf_narrow = c->leaf_wide = false;
    c->nodes.clear(); c->levels.clear(); c->snodes.clear(); c->sfolds.clear();
    c->sweep_levels.clear(); c->n_group_merges = 0;
    // group exchange: the record [N flags | accepted count | gate bytes (msckf_set_exchange_mask) | N slots of XCHG_SLOT doubles]
    // heads the workspace; the triangle of
    // group s (fixed window of min(10, N - s) slots, so its shape depends on (N, s) only) is produced in slot s
    const bool xchg = c->xchg;
    size_t off = xchg ? rec_head(c) + (size_t)N * XCHG_SLOT : 0;
    if (xchg) c->h_xflags.assign(N, 0.0);
    struct Tri { long long src; int lo, w; int lvl = -1, idx = -1, ld = 0; };     // (lvl, idx): the merge node that writes it, if one does; ld: its row stride (0: w + 1)
    const int merge_ld = (c->stream_enabled && mode == 0 && !xchg) ? 64 : 0;      // merge outputs with whole cache lines per row (streamable, k_sweep.h)
    std::vector<Tri> group_tri;                                           // one triangle per group, by first slot
    std::vector<std::vector<SweepNode>> merge_levels;                     // [level] -> nodes of every group at that depth
    // every run is sorted by (first slot, last slot) on its own (the tracks; the narrow blocks of split long tracks): the
    // groups are walked over all runs together, a leaf takes entries of ONE run
    std::vector<int> cur(runs.size());
    for (size_t r = 0; r < runs.size(); ++r) cur[r] = runs[r].b;
    for (;;) {
        int s = 1 << 30;
        for (size_t r = 0; r < runs.size(); ++r) {
            while (cur[r] < runs[r].e && !live(cur[r])) ++cur[r];
            if (cur[r] < runs[r].e) s = std::min(s, fmin[cur[r]]);
        }
        if (s == (1 << 30)) break;
        // leaves of this group
        std::vector<Tri> leaves;
        for (size_t r = 0; r < runs.size(); ++r) {
        int& f = cur[r];
        const int F = runs[r].e;
        while (f < F && (!live(f) || fmin[f] == s)) {
            if (!live(f)) { ++f; continue; }
            int hi = fmax[f], rows = 0, e = f, last = f, nlive = 0;
            while (e < F && (e - f) < 128 && (!live(e) || fmin[e] == s)) {
                if (live(e)) {
                    const int r = std::max(2 * (view_sorted[e + 1] - view_sorted[e]) - 3, 1);
                    if (e > f && (rows + r > leaf_rows || nlive >= leaf_feats)) break;
                    rows += r; ++nlive;
                    hi = std::max(hi, fmax[e]);
                    last = e;
                }
                ++e;
            }
            FoldNode n{};
            if (xchg) hi = std::min(s + XW / 6, N) - 1;
            n.kind = 0; n.src_begin = f; n.src_end = last + 1; n.win_lo = s; n.w = 6 * (hi - s + 1); n.pad = 0;
            n.out_off = (long long)off;
            off += (size_t)n.w * (n.w + 1);
            c->nodes.push_back(n);
            if (n.w + 1 > 64) c->leaf_wide = true; else c->leaf_narrow = true;
            leaves.push_back({n.out_off, s, n.w});
            f = last + 1;
        }
        }
        // merge levels of this group: one k_sweep node folds up to 2 * SWEEP_NW triangles (two rounds of the
        // fold slots); larger groups first reduce chunks of SWEEP_NW triangles in parallel workgroups
        std::vector<Tri> cur = leaves;
        const long long xdest = xchg ? (long long)(rec_head(c) + (size_t)s * XCHG_SLOT) : -1;
        auto merge_node = [&](size_t b, size_t e, int level, long long dest = -1) -> Tri {
            SweepNode m{};
            m.fold_begin = (int)c->sfolds.size();
            int wtot = 0, env = 0;
            for (size_t i = b; i < e; ++i) {
                env = std::max(env, cur[i].w);
                SweepFold sf{}; sf.src_off = cur[i].src; sf.off = 0; sf.w = cur[i].w; sf.ew = env; sf.ld = cur[i].ld;
                c->sfolds.push_back(sf);
                wtot = std::max(wtot, cur[i].w);
            }
            m.fold_end = (int)c->sfolds.size();
            m.wtot = wtot;
            m.nsteps = 0;                                                  // (scheduled per level below: one kernel, one slot count)
            if (dest >= 0) m.out_off = dest;
            else { m.ldo = merge_ld; m.out_off = (long long)off; off += (size_t)wtot * (merge_ld ? merge_ld : wtot + 1); }
            if ((int)merge_levels.size() <= level) merge_levels.resize(level + 1);
            merge_levels[level].push_back(m);
            return Tri{m.out_off, s, wtot, level, (int)merge_levels[level].size() - 1, m.ldo};
        };
        int level = 0;
        while (cur.size() > (size_t)(2 * SWEEP_NW)) {
            std::vector<Tri> nxt;
            for (size_t b = 0; b < cur.size(); b += SWEEP_NW) {
                const size_t e = std::min(cur.size(), b + SWEEP_NW);
                nxt.push_back(e - b == 1 ? cur[b] : merge_node(b, e, level));
            }
            cur.swap(nxt);
            ++level;
        }
        if (cur.size() > 1) { const Tri tmerged = merge_node(0, cur.size(), level, xdest); cur.assign(1, tmerged); }
        else if (xchg) {
            // a single leaf: it writes straight into the record slot
            for (FoldNode& nd : c->nodes) if (nd.out_off == cur[0].src) { nd.out_off = xdest; break; }
            cur[0].src = xdest;
        }
        if (xchg) c->h_xflags[s] = 1.0;
        group_tri.push_back(cur[0]);
    }
    // the leaves of the first run (the short tracks) in front of the others (the narrow blocks of split long tracks, written by a
    // kernel on the second stream): they can start while that kernel runs.  (The merges name their sources by offset, not by node.)
    c->n_leaves0 = (int)c->nodes.size();
    if (runs.size() > 1) {
        const int b1 = runs[1].b;
        auto mid = std::stable_partition(c->nodes.begin(), c->nodes.end(), [b1](const FoldNode& n) { return n.src_begin < b1; });
        c->n_leaves0 = (int)(mid - c->nodes.begin());
    }
    c->n_leaves = (int)c->nodes.size();
    if (c->n_leaves > 0) c->levels.push_back({0, c->n_leaves});
    c->sweep_levels.clear();
    c->sweep_level_nf.clear();
    for (auto& lv : merge_levels) {
        // a level is one launch: twelve fold slots when a node has more triangles than eight slots take in one round
        // (groups of 9 - 16 leaf triangles at >= 10000 features: 60 + 9 macro steps instead of two rounds of 61)
        int most = 0;
        for (const SweepNode& m : lv) most = std::max(most, m.fold_end - m.fold_begin - 1);
        const int nf = (mode == 0 && most > SWEEP_NW) ? ((most <= SWEEP_NW_MID && c->stream_enabled && !xchg) ? SWEEP_NW_MID : SWEEP_NW_BIG) : SWEEP_NW;
        for (SweepNode& m : lv) sweep_schedule(c->sfolds, m.fold_begin, m.fold_end, &m.nsteps, nf);
        c->sweep_levels.push_back({(int)c->snodes.size(), (int)lv.size()});
        c->sweep_level_nf.push_back(nf);
        c->snodes.insert(c->snodes.end(), lv.begin(), lv.end());
    }
    c->n_group_merges = (int)c->snodes.size();
    // the last merge level goes into the root's launch where that launch exists (k_root_gain: 60-column sweeps, eight fold
    // slots) and its triangles are streamed to the root
    c->root_streamed = false; c->stream_level = -1; c->h_mflush.clear();
    {
        const int last = (int)merge_levels.size() - 1;
        // (every workgroup of that launch -- root, strips, merge nodes -- holds a CU of its own while it waits for the others:
        //  at most half of the device, so that a partitioned GPU or a kernel on another stream cannot keep a producer out)
        if (c->stream_enabled && mode == 0 && !xchg && last >= 0 && c->sweep_level_nf[last] <= SWEEP_NW_MID && (int)merge_levels[last].size() <= 64 &&
            group_tri.size() > 1 && 2 * (2 + (dc + 15) / 16 + (int)merge_levels[last].size()) <= c->n_cu) {
            c->root_streamed = true; c->stream_level = last;
        }
    }
    if (!group_tri.empty()) {
        SweepNode r{};
        r.fold_begin = (int)c->sfolds.size();
        int env = 0;
        for (const Tri& g : group_tri) {
            env = std::max(env, 6 * g.lo + g.w);
            SweepFold sf{}; sf.src_off = g.src; sf.off = 6 * g.lo; sf.w = g.w; sf.ew = env - 6 * g.lo; sf.ld = g.ld;
            if (c->root_streamed && g.lvl == c->stream_level) sf.prod = g.idx + 1;
            c->sfolds.push_back(sf);
        }
        r.fold_end = (int)c->sfolds.size();
        r.wtot = dc;
        sweep_schedule(c->sfolds, r.fold_begin, r.fold_end, &r.nsteps, SWEEP_NW, !c->root_streamed);
        r.out_off = (long long)off;
        c->root_off = off;
        off += (size_t)dc * (dc + 1);
        c->snodes.push_back(r);
        c->root = 0;
    } else {
        c->root = -1;
        c->root_off = 0;
    }
    c->zero_off = off;
    off += 16;
    c->rbuf_doubles = off;
    c->xchg_planned = xchg;
    c->h_flush.clear(); c->h_flush_off.clear();
    c->h_root_flush.clear();
    c->root_band = 0;
    if (!group_tri.empty()) {
        // the root's rows become final one by one as the sweep passes them: K6-K7 (k_gstream.h) follows them block by block
        const SweepNode& rn = c->snodes.back();
        for (int g = rn.fold_begin; g < rn.fold_end; ++g) c->root_band = std::max(c->root_band, c->sfolds[g].ew);
        if (mode == 0) sweep_flush_table(c->sfolds, rn.fold_begin, rn.fold_end, rn.nsteps, rn.wtot, 1 << 29, c->h_root_flush);
        c->root_n_gate = -1;
        if (c->root_streamed) {
            c->root_n_gate = sweep_gate_table(c->sfolds, rn.fold_begin, rn.fold_end, rn.nsteps, SWEEP_NW, c->h_root_flush);
            const auto& lv = c->sweep_levels[c->stream_level];
            c->h_mflush.assign(lv.second, 0);
            // (nodes with the same folds -- the same number of triangles of the same widths: most of them -- share one table)
            std::vector<std::pair<std::vector<int>, int>> seen_shapes;
            for (int i = 0; i < lv.second; ++i) {
                const SweepNode& m = c->snodes[lv.first + i];
                std::vector<int> shape{m.wtot, m.nsteps};
                for (int g = m.fold_begin; g < m.fold_end; ++g) { shape.push_back(c->sfolds[g].w); shape.push_back(c->sfolds[g].ew); shape.push_back(c->sfolds[g].t0); shape.push_back(c->sfolds[g].off); }
                int at = -1;
                for (const auto& sh : seen_shapes) if (sh.first == shape) { at = sh.second; break; }
                if (at < 0) {
                    at = (int)c->h_mflush.size() - lv.second;
                    sweep_flush_table(c->sfolds, m.fold_begin, m.fold_end, m.nsteps, m.wtot, 1 << 29, c->h_mflush);
                    seen_shapes.push_back({std::move(shape), at});
                }
                c->h_mflush[i] = at;
            }
            c->mflush_at = (int)c->h_root_flush.size();         // (one upload: the merge nodes' tables ride behind the root's)
            c->h_root_flush.insert(c->h_root_flush.end(), c->h_mflush.begin(), c->h_mflush.end());
        }
    }
    if (mode > 0) {
        const int rc = 1 << (mode == 1 ? WS_RC_LOG2_4 : WS_RC_LOG2_6);
        for (const SweepNode& nd : c->snodes) {
            const size_t o = c->h_flush.size();
            c->h_flush_off.push_back((int)o);
            if (!sweep_flush_table(c->sfolds, nd.fold_begin, nd.fold_end, nd.nsteps, nd.wtot, rc, c->h_flush)) return false;
            sweep_publish_table(c->h_flush, o, nd.nsteps, nd.wtot);
        }
    }
    return true;
}

// plan for the current batch: band pipeline when it qualifies, else the tree.  `valid_in` (k_select): one byte per TRACK.
void plan_batch(msckf_ctx* c, const std::vector<int>& fmin, const std::vector<int>& fmax,
                const std::vector<int>& view_sorted, const std::vector<unsigned char>* valid_in = nullptr) {
    c->xchg_planned = false;
    c->wide_active = false;
    c->rnodes.clear(); c->rlevels.clear(); c->rroot = -1; c->rroot_off = 0; c->rtops.clear(); c->rtop_rows = 0;
    const int F = c->F, Fs = c->Fs;
    // a block is as valid as the track it was split off
    std::vector<unsigned char> vfull;
    const std::vector<unsigned char>* valid = valid_in;
    if (valid_in && Fs > F) {
        vfull.assign(valid_in->begin(), valid_in->begin() + F);
        vfull.resize(Fs);
        for (int i = F; i < Fs; ++i) vfull[i] = (*valid_in)[c->h_parent[i - F]];
        valid = &vfull;
    }
    auto tree_only = [&](const std::vector<Run>& runs) {
        c->band_plan = false;
        c->snodes.clear(); c->sfolds.clear(); c->sweep_levels.clear(); c->n_group_merges = 0;
        c->h_root_flush.clear();
        size_t off_end = 0;
        build_tree(c, fmin, fmax, view_sorted, valid, runs, 0, c->nodes, c->levels, c->root, c->root_off, off_end, c->n_leaves);
        c->rbuf_doubles = off_end;
        c->root_band = c->dc;                                             // a dense triangle
    };
    if (c->split_on) {
        // (three runs, each sorted by first slot on its own: a leaf's rows must come sorted by their first column)
        const std::vector<Run> band_runs{{0, c->Fb}, {F, F + c->nNarrow}}, all_runs{{0, c->Fb}, {F, F + c->nNarrow}, {F + c->nNarrow, Fs}};
        if (!c->no_wide && build_plan_band(c, fmin, fmax, view_sorted, valid, band_runs)) {
            c->band_plan = true; c->wide_active = true;
            // the remainder blocks: taken by K6-K7 as they are when they are few (rem_direct), else a merge tree of their own
            // behind the band plan's workspace
            if (c->rem_direct) return;
            size_t off_end = 0; int nl = 0;
            // (a leaf is ONE register batch of k_fold -- 192 rows at w = 180, ~290 us whatever it holds.  Measured at 2000 tracks
            //  ~ U[2, 30], 5850 rows, leaves of 96 / 128 / 160 / 176 / 192 / 208 / 256 rows: 2149 / 2151 / 2113 / 2097 / 1940 / 2164 /
            //  2175 us per update; by the 3-per-group bound, i.e. two batches per leaf, 2160; leaves of 1024 rows with merges of up
            //  to six triangles 2415)
            build_tree(c, fmin, fmax, view_sorted, valid, {{F + c->nNarrow, Fs}}, c->rbuf_doubles, c->rnodes, c->rlevels, c->rroot,
                       c->rroot_off, off_end, nl, c->rem_leaf_rows > 0 ? c->rem_leaf_rows : (c->dc > FOLD_RLDS_MAX_W ? 512 : fold_bmax(c->dc)), false,
                       c->rem_cut_rows >= 0 ? c->rem_cut_rows : (c->dc > FOLD_RLDS_MAX_W ? 0 : 600), &c->rtops);
            for (int k : c->rtops) c->rtop_rows += c->rnodes[k].w;
            c->rbuf_doubles = off_end;
            return;
        }
        tree_only(all_runs);           // (blocks that leave the context, MSCKF_FLAG-forced plans: ONE root block)
        return;
    }
    const std::vector<Run> runs{{0, F}};
    c->band_plan = build_plan_band(c, fmin, fmax, view_sorted, valid, runs);
    if (!c->band_plan) tree_only(runs);
}

int upload_plan(msckf_ctx* c);

int launch_fold_levels(msckf_ctx* c, const std::vector<std::pair<int, int>>& levels,
                       const std::vector<FoldNode>& all_nodes, hipStream_t st = nullptr, const FoldNode* dnodes = nullptr) {
    if (!st) st = c->stream;
    FoldArgs a{};
    a.nodes = dnodes ? dnodes : ptr<FoldNode>(c->dNodes);
    a.lds_doubles = FOLD_LDS_BYTES / 8;
    a.view_ptr = ptr<int>(c->dViewPtr);
    a.obs_slot = ptr<int>(c->dObsSlot);
    a.fmin = ptr<int>(c->dFmin);
    a.blk_off = ptr<long long>(c->dBlkOff);
    a.stack = c->dStack.p; a.stack_f32 = c->cfg.dtype == MSCKF_DTYPE_F32 ? 1 : 0;
    a.rank = ptr<int>(c->dRank);
    a.accepted = ptr<unsigned char>(c->dAcc);
    a.rbuf = ptr<double>(c->dRbuf);
    a.stamps = c->dStamps.p ? ptr<long long>(c->dStamps) : nullptr;
    for (auto& lv : levels) {
        a.node_base = lv.first;
        int maxw = 0, leaf_rows_max = 0;
        for (int i = lv.first; i < lv.first + lv.second; ++i) {
            const FoldNode& n = all_nodes[i];
            maxw = std::max(maxw, n.w);
            if (n.kind == 0) {
                const int rows = 2 * (c->h_view_sorted[n.src_end] - c->h_view_sorted[n.src_begin]);
                leaf_rows_max = std::max(leaf_rows_max, rows);
            } else {
                leaf_rows_max = 1 << 30;
            }
        }
        const dim3 grid(lv.second);
        if (maxw <= FOLD_RLDS_MAX_W) {          // R accumulator resident in LDS
            const dim3 block(FOLD_T);
            const bool leaf = all_nodes[lv.first].kind == 0;
            const bool small = leaf && leaf_rows_max <= FOLD_RL * FOLD_RPT_LEAF;
            const int cls = fold_class(maxw);
#define FOLD_LAUNCH(RPT, CPT) \
    hipLaunchKernelGGL((k_fold<FOLD_T, FOLD_RL, RPT, CPT>), grid, block, FOLD_LDS_BYTES, st, a)
            if (cls == 1) {
                if (small) FOLD_LAUNCH(FOLD_RPT_LEAF, FOLD_CPT1);
                else FOLD_LAUNCH(FOLD_RPT_BIG, FOLD_CPT1);
            } else if (cls == 2) {
                if (small) FOLD_LAUNCH(FOLD_RPT_LEAF, FOLD_CPT2);
                else FOLD_LAUNCH(FOLD_RPT_BIG, FOLD_CPT2);
            } else {
                FOLD_LAUNCH(FOLD_RPT_W3, FOLD_CPT3);
            }
#undef FOLD_LAUNCH
        } else {                                // wide windows: R streamed through HBM
            const dim3 block(FOLDG_T);
            if (maxw + 1 <= 6 * 32) hipLaunchKernelGGL((k_fold_g<FOLDG_T, 12, 6>), grid, block, FOLD_LDS_BYTES, st, a);
            else hipLaunchKernelGGL((k_fold_g<FOLDG_T, 6, 10>), grid, block, FOLD_LDS_BYTES, st, a);
        }
    }
    HIPCHK(c, hipGetLastError());
    return MSCKF_OK;
}

// band plan, level 0: the leaves fold their features' K4 blocks (k_lsweep)
int launch_leaves_band(msckf_ctx* c, int node_base = 0, int count = -1) {
    if (count < 0) count = c->n_leaves - node_base;
    if (count <= 0) return MSCKF_OK;
    LSweepArgs a{};
    a.nodes = ptr<FoldNode>(c->dNodes);
    a.node_base = node_base;
    a.info = ptr<FeatInfo>(c->dFeatInfo);
    a.stack = c->dStack.p; a.stack_f32 = c->cfg.dtype == MSCKF_DTYPE_F32 ? 1 : 0;
    a.rank = ptr<int>(c->dRank);
    a.accepted = ptr<unsigned char>(c->dAcc);
    a.rbuf = ptr<double>(c->dRbuf);
    a.zero_idx = c->stack_elems;
    const dim3 grid(count), block(64 * SWEEP_NW);
    if (c->leaf_narrow) {
        a.wide = 0;
        if (c->leaf_nf == 12) {          // large batches: twelve row blocks in flight (three wavefronts per SIMD), aligned rounds
            const size_t lds = lsweep_lds_bytes<4, LS_RS4>(12);
            hipLaunchKernelGGL((k_lsweep<12, 4, LS_RS4, false>), grid, dim3(64 * 12), lds, c->stream, a);
        } else {                         // eight, the next block prefetched into registers
            const size_t lds = lsweep_lds_bytes<4, LS_RS4>(8);
            hipLaunchKernelGGL((k_lsweep<8, 4, LS_RS4, true>), grid, dim3(64 * 8), lds, c->stream, a);
        }
    }
    if (c->leaf_wide) {
        a.wide = 1;
        if (c->leaf_tall) {              // two long tracks per row block, eight wavefronts (the tile alone is 168 registers)
            const size_t lds = lsweep_lds_bytes<6, LS_RS6T>(8);
            hipLaunchKernelGGL((k_lsweep<8, 6, LS_RS6T>), grid, dim3(64 * 8), lds, c->stream, a);
        } else if (c->leaf_nf == 12) {          // large batches: twelve row blocks in flight (three wavefronts per SIMD)
            const size_t lds = lsweep_lds_bytes<6, LS_RS6>(12);
            hipLaunchKernelGGL((k_lsweep<12, 6, LS_RS6, false>), grid, dim3(64 * 12), lds, c->stream, a);
        } else {
            const size_t lds = lsweep_lds_bytes<6, LS_RS6>(SWEEP_NW);
            hipLaunchKernelGGL((k_lsweep<SWEEP_NW, 6, LS_RS6>), grid, block, lds, c->stream, a);
        }
    }
    HIPCHK(c, hipGetLastError());
    return MSCKF_OK;
}

// band plan, levels 1-2: group merges (one workgroup each), then the root sweep
template <int CS>
void launch_wsweep(msckf_ctx* c, int node_base, int count, int rc_log2, int nsteps_max = -1, const double* zero = nullptr) {
    WSweepArgs a{};
    a.nodes = ptr<SweepNode>(c->dSweepNodes);
    a.folds = ptr<SweepFold>(c->dSweepFolds);
    a.node_base = node_base;
    a.rbuf = ptr<double>(c->dRbuf);
    a.zero = zero ? zero : ptr<double>(c->dRbuf) + c->zero_off;
    a.flush = ptr<int>(c->dFlush);
    a.flush_off = ptr<int>(c->dFlushOff);
    a.rc_log2 = rc_log2;
    int nsteps = nsteps_max;
    if (nsteps < 0) { nsteps = 0; for (int i = node_base; i < node_base + count; ++i) nsteps = std::max(nsteps, c->snodes[i].nsteps); }
    hipLaunchKernelGGL((k_wsweep<SWEEP_NW, CS>), dim3(count), dim3(64 * SWEEP_NW), wsweep_lds_bytes<CS>(1 << rc_log2, SWEEP_NW, nsteps),
                       c->stream, a);
}

int launch_sweeps(msckf_ctx* c, bool with_root = true, int skip_level = -1) {
    if (c->snodes.empty()) return MSCKF_OK;
    if (c->sweep_mode > 0) {
        auto go = [&](int base, int count) {
            if (c->sweep_mode == 1) launch_wsweep<4>(c, base, count, WS_RC_LOG2_4);
            else launch_wsweep<6>(c, base, count, WS_RC_LOG2_6);
        };
        for (const auto& lv : c->sweep_levels) go(lv.first, lv.second);
        if (with_root) go(c->n_group_merges, 1);
        HIPCHK(c, hipGetLastError());
        return MSCKF_OK;
    }
    SweepArgs a{};
    a.nodes = ptr<SweepNode>(c->dSweepNodes);
    a.folds = ptr<SweepFold>(c->dSweepFolds);
    a.rbuf = ptr<double>(c->dRbuf);
    a.stamps = c->dStamps.p ? ptr<long long>(c->dStamps) : nullptr;
    a.zero = ptr<double>(c->dRbuf) + c->zero_off;          // inside the plan's (zero-initialised, never written) region
    const dim3 block(64 * SWEEP_NW * SWEEP_WPF);
    for (size_t li = 0; li < c->sweep_levels.size(); ++li) {
        if ((int)li == skip_level) continue;               // (inside k_root_gain's launch, streamed to the root)
        const auto& lv = c->sweep_levels[li];
        int wmax = 0;
        for (int i = lv.first; i < lv.first + lv.second; ++i) wmax = std::max(wmax, c->snodes[i].wtot);
        a.node_base = lv.first;
        a.stamp_base = (int)c->nodes.size() + lv.first;
        if (li < c->sweep_level_nf.size() && c->sweep_level_nf[li] > SWEEP_NW)      // (an eleven-slot schedule runs on twelve slots as well)
            hipLaunchKernelGGL((k_sweep<SWEEP_NW_BIG, 1>), dim3(lv.second), dim3(64 * SWEEP_NW_BIG), sweep_lds_bytes(wmax, SWEEP_NW_BIG, 1), c->stream, a);
        else
            hipLaunchKernelGGL((k_sweep<SWEEP_NW, SWEEP_WPF>), dim3(lv.second), block, sweep_lds_bytes(wmax, SWEEP_NW, SWEEP_WPF), c->stream, a);
    }
    if (with_root) {
        a.node_base = c->n_group_merges;
        a.stamp_base = (int)c->nodes.size() + c->n_group_merges;
        // the root's folds start 6 columns apart: point-to-point progress words instead of a barrier per macro step
        hipLaunchKernelGGL((k_sweep<SWEEP_NW, SWEEP_WPF, SWEEP_P2P>), dim3(1), block, sweep_lds_bytes(c->snodes.back().wtot, SWEEP_NW, SWEEP_WPF), c->stream, a);
    }
    HIPCHK(c, hipGetLastError());
    return MSCKF_OK;
}

// node / fold tables of the current plan -> HBM (async on the context's stream)
int upload_plan(msckf_ctx* c) {
    hipStream_t ps = c->plan_stream ? c->plan_stream : c->stream;
    // The tables (leaf nodes, sweep nodes and folds, the flush / gate tables) go up as ONE image: pinned staging -> one arena in
    // HBM, the buffers the launches name are views into it.  (As four to six copies they cost the host ~5 us each inside the
    // window of K1-K4, and the device a blit kernel each.)  The staging image is reused: wait for the previous plan's copy
    // first (an event that has long passed by then).
    size_t need = 0;
    auto room = [&](size_t bytes) { const size_t o = need; need += (std::max<size_t>(bytes, 64) + 255) & ~(size_t)255; return o; };
    const size_t b_nodes = c->nodes.size() * sizeof(FoldNode), b_sn = c->snodes.size() * sizeof(SweepNode), b_sf = c->sfolds.size() * sizeof(SweepFold);
    const size_t b_rf = c->h_root_flush.size() * 4, b_fl = c->h_flush.size() * 4, b_fo = c->h_flush_off.size() * 4;
    const size_t b_rn = c->rnodes.size() * sizeof(FoldNode);
    const size_t o_nodes = room(b_nodes), o_sn = room(b_sn), o_sf = room(b_sf), o_rf = room(b_rf), o_fl = room(b_fl), o_fo = room(b_fo);
    const size_t o_rn = room(b_rn);
    if (c->plan_staged) { HIPCHK(c, hipEventSynchronize(c->ev_plan_up)); c->plan_staged = false; }
    if (c->hPlanCap < need) {
        if (c->hPlan) HIPCHK(c, hipHostFree(c->hPlan));
        c->hPlan = nullptr; c->hPlanCap = 0;
        HIPCHK(c, hipHostMalloc(&c->hPlan, 2 * need + 4096));
        c->hPlanCap = 2 * need + 4096;
    }
    // (buffers of their own from a merge plan that outgrew the arena: released before they become views again)
    for (Buf* bb : {&c->dNodes, &c->dSweepNodes, &c->dSweepFolds, &c->dRootFlush, &c->dFlush, &c->dFlushOff, &c->dRNodes})
        if (bb->p && !bb->view) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(bb->p)); bb->p = nullptr; bb->bytes = 0; }
    if (c->dPlanArena.bytes < need) {
        HIPCHK(c, hipStreamSynchronize(c->stream));        // (kernels of an earlier run may still read the old arena)
        if (int rc = ensure(c, c->dPlanArena, 2 * need)) return rc;
    }
    char* hp = static_cast<char*>(c->hPlan);
    if (b_nodes) std::memcpy(hp + o_nodes, c->nodes.data(), b_nodes);
    if (b_sn) std::memcpy(hp + o_sn, c->snodes.data(), b_sn);
    if (b_sf) std::memcpy(hp + o_sf, c->sfolds.data(), b_sf);
    if (b_rf) std::memcpy(hp + o_rf, c->h_root_flush.data(), b_rf);
    if (b_fl) std::memcpy(hp + o_fl, c->h_flush.data(), b_fl);
    if (b_fo) std::memcpy(hp + o_fo, c->h_flush_off.data(), b_fo);
    if (b_rn) std::memcpy(hp + o_rn, c->rnodes.data(), b_rn);
    HIPCHK(c, hipMemcpyAsync(c->dPlanArena.p, hp, need, hipMemcpyHostToDevice, ps));
    set_view(c->dNodes, c->dPlanArena.p, o_nodes, std::max<size_t>(b_nodes, 64));
    set_view(c->dSweepNodes, c->dPlanArena.p, o_sn, std::max<size_t>(b_sn, 64));
    set_view(c->dSweepFolds, c->dPlanArena.p, o_sf, std::max<size_t>(b_sf, 64));
    set_view(c->dRootFlush, c->dPlanArena.p, o_rf, std::max<size_t>(b_rf, 64));
    set_view(c->dFlush, c->dPlanArena.p, o_fl, std::max<size_t>(b_fl, 64));
    set_view(c->dFlushOff, c->dPlanArena.p, o_fo, std::max<size_t>(b_fo, 64));
    set_view(c->dRNodes, c->dPlanArena.p, o_rn, std::max<size_t>(b_rn, 64));
    c->x_plan_valid = false;               // the sweep tables are rewritten: a cached merge plan behind them is gone
    if (c->xchg_planned)                   // (behind the workspace memset of set_features / replan, same stream)
        HIPCHK(c, hipMemcpyAsync(c->dRbuf.p, c->h_xflags.data(), c->h_xflags.size() * 8, hipMemcpyHostToDevice, ps));
    HIPCHK(c, hipEventRecord(c->ev_plan_up, ps));
    c->plan_staged = true;
    return MSCKF_OK;
}

int launch_feature(msckf_ctx* c) {
    if (c->F == 0) return MSCKF_OK;
    FeatureArgs a{};
    a.F = c->F; a.ldp = c->d;
    a.view_ptr = ptr<int>(c->dViewPtr); a.obs_uv = ptr<double>(c->dObsUV); a.obs_slot = ptr<int>(c->dObsSlot);
    a.idp_base = ptr<double>(c->dBase); a.idp_m = ptr<double>(c->dMvec); a.idp_rho = ptr<double>(c->dRho);
    a.cam_R = ptr<double>(c->dCamR); a.cam_t = ptr<double>(c->dCamT);
    a.cam_R0 = ptr<double>(c->dCamR0); a.cam_t0 = ptr<double>(c->dCamT0);
    a.P = ptr<double>(c->dP); a.chi2 = ptr<double>(c->dChi2); a.n_chi2 = c->n_chi2;
    for (int i = 0; i < 3; ++i) a.g[i] = c->g[i];
    for (int i = 0; i < 9; ++i) a.Kinv[i] = c->Kinv[i];
    a.sigma2 = c->sigma * c->sigma;
    a.blk_off = ptr<long long>(c->dBlkOff); a.stack = c->dStack.p; a.stack_f32 = c->cfg.dtype == MSCKF_DTYPE_F32 ? 1 : 0;
    a.rank = ptr<int>(c->dRank); a.accepted = ptr<unsigned char>(c->dAcc); a.gamma = ptr<double>(c->dGamma);
    a.select = c->use_select ? ptr<unsigned char>(c->dSelFlags) : nullptr;
    a.stamps = c->dStamps.p ? ptr<long long>(c->dStamps) + 8 * 8192 : nullptr;   // behind the fold stamps
    a.zero_idx = c->stack_elems;
    c->gate_direct = c->want_direct && !c->use_select;      // (long tracks: k_feature<64, true> mirrors its track's results as well)
    if (c->gate_direct) { a.rank_h = static_cast<int*>(c->hGate); a.acc_h = static_cast<unsigned char*>(c->hGate) + (size_t)c->Fs * 4; }
    // one launch per class of tracks: the short tracks [0, Fb) and the long ones [Fb, F), which are split (k_feature<64, true>
    // writes their narrow and remainder blocks)
    hipStream_t st = c->stream;
    auto go = [&](int f0, int nf, int mmax) {
        if (nf <= 0) return;
        a.f0 = f0; a.F = nf; a.split = nullptr;
        const bool chunked = 2 * mmax + 1 > 32;                // k_feature<64>: column chunks, S in registers
        int lds_d = 0;                   // (the footprint is not monotone in the track length: whole-view chunks)
        for (int m = 1; m <= mmax; ++m) lds_d = std::max(lds_d, feature_lds_doubles(m, chunked));
        const size_t lds = (size_t)lds_d * 8;
        if (mmax <= 10) hipLaunchKernelGGL(k_feature<24>, dim3(nf), dim3(64), lds, st, a);            // one chunk of <= 60 columns
        else if (2 * mmax + 1 <= 32) hipLaunchKernelGGL(k_feature<32>, dim3(nf), dim3(64), lds, st, a);
        else hipLaunchKernelGGL(k_feature<64>, dim3(nf), dim3(64), lds, st, a);
    };
    auto go_split = [&](int f0, int nf, int mmax) {
        if (nf <= 0) return;
        a.f0 = f0; a.F = nf; a.split = ptr<SplitRec>(c->dSplit);
        // few long tracks: four wavefronts per track (view groups and gate chunks in parallel: ~75 instead of 140 us at 30 views,
        // all of it in front of the band pipeline's leaves); many: one wavefront per track fills the chip better (three per CU)
        static const int mw_max = [] { const char* e = std::getenv("MSCKF_SPLIT_MW_MAX"); return e ? std::atoi(e) : 512; }();
        const int nwv = nf <= mw_max ? 4 : 1;
        int lds_d = 0;
        for (int m = 2; m <= mmax; ++m) lds_d = std::max(lds_d, feature_split_lds_doubles(m, nwv));
        if (nwv == 4) hipLaunchKernelGGL((k_feature<64, true, 4>), dim3(nf), dim3(256), (size_t)lds_d * 8, st, a);
        else hipLaunchKernelGGL((k_feature<64, true>), dim3(nf), dim3(64), (size_t)lds_d * 8, st, a);
        if (c->rem_direct) {             // few remainder rows: one dense matrix for K6-K7, no QR of their own
            RemScatterArgs r{};
            r.split = ptr<SplitRec>(c->dSplit); r.n_tracks = nf; r.rows_cap = c->rem_cap;
            r.dc = c->dc; r.N = c->N;
            r.view_ptr = a.view_ptr; r.obs_slot = a.obs_slot; r.blk_off = a.blk_off; r.stack = a.stack; r.stack_f32 = a.stack_f32;
            r.rank = a.rank; r.accepted = a.accepted; r.out = ptr<double>(c->dRem);
            r.nrows = reinterpret_cast<int*>(ptr<double>(c->dRem) + (size_t)16 * GS_MAX_NB2 * (6 * c->maxN + 1));   // behind the matrix
            hipLaunchKernelGGL(k_rem_scatter, dim3(nf + 1), dim3(256), 0, st, r);
        }
    };
    c->wide_on_stream2 = false;
    if (c->split_on && c->Fw > 0) {
        // a long track's kernel is ONE wavefront's latency per track (~6 us per view: 190 us at 30 views) however few they are: it
        // runs on the second stream beside the short tracks' launch; the leaves wait for both (ev_wfeat, run_pipeline)
        if (c->wide_concurrent) {
            HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
            st = c->stream2;
            go_split(c->Fb, c->Fw, c->Mmax_wide);
            HIPCHK(c, hipEventRecord(c->ev_wfeat, c->stream2));
            c->wide_on_stream2 = true;
            st = c->stream;
            go(0, c->Fb, c->Mmax_band);
        } else {
            go(0, c->Fb, c->Mmax_band); go_split(c->Fb, c->Fw, c->Mmax_wide);
        }
    } else go(0, c->F, c->Mmax);
    HIPCHK(c, hipGetLastError());
    return MSCKF_OK;
}

void gemm(msckf_ctx* c, const double* A, int lda, const double* B, int ldb, const double* C0, int ldc0,
          double* C, int ldc, int M, int N, int K, double alpha, double beta, double diag, int transB, int tri) {
    GemmArgs g{A, lda, B, ldb, C0, ldc0, C, ldc, M, N, K, alpha, beta, diag, transB, tri};
    hipLaunchKernelGGL(k_gemm_f64, dim3((N + 15) / 16, (M + 15) / 16), dim3(64 * GEMM_WAVES), 0, c->stream, g);
}

// One-sided / two-sided triangular sweeps over the rows of X (in place) with the unit-diagonal packed factor Lp.
template <int MODE>
void launch_tri_sweep(msckf_ctx* c, double* X, int ldx, int rows, const double* Lp, const double* invd, int n) {
    SolveArgs a{};
    a.Y = X; a.ldy = ldx; a.L = nullptr; a.U = nullptr; a.invd = invd; a.Lp = Lp; a.n = n;
    a.z = nullptr; a.zstride = 0; a.Kg = X; a.ldk = ldx; a.dx = nullptr; a.d = rows;
    constexpr int WV = SOLVE_WAVES;
    const dim3 grid((rows + WV - 1) / WV), block(64 * WV);
    const size_t lds_need = ((size_t)n * (n + 1) / 2 + n) * 8;
    const int nreg = (n + 63) / 64;
    if (nreg <= 1) hipLaunchKernelGGL((k_solve_lds<1, WV, 1, true, MODE>), grid, block, lds_need, c->stream, a);
    else if (nreg <= 2) hipLaunchKernelGGL((k_solve_lds<2, WV, 1, true, MODE>), grid, block, lds_need, c->stream, a);
    else hipLaunchKernelGGL((k_solve_lds<3, WV, 1, true, MODE>), grid, block, lds_need, c->stream, a);
}

// K6 for windows wider than one single-workgroup Cholesky (4 CHOL_TILE_MAX_NT < dc <= 2 GAIN_BLK): S is factored
// as a 2 x 2 block matrix, S = [A B^T; B C], L = [L11 0; W L22] with W = B L11^-T, L22 L22^T = C - W W^T; both
// diagonal factors come from launch_chol_small (k_chol16), everything else is triangular sweeps with the factor in LDS and MFMA
// GEMMs.  K = Y S^-1 row by row:  X1 = Y1 L11^-T,  K2 = (Y2 - X1 W^T) L22^-T L22^-1,  K1 = (X1 - K2 W) L11^-1.
constexpr int GAIN_BLK = 160;
// Cholesky of a matrix of at most 4 * CHOL_TILE_MAX_NT rows, one workgroup.
void launch_chol_small(msckf_ctx* c, const CholArgs& a) {
#if MSCKF_CHOL16
    hipLaunchKernelGGL(k_chol16, dim3(1), dim3(64 * CHOL16_W), 0, c->stream, a);
#else
    hipLaunchKernelGGL((k_chol_tile<CHOL_T>), dim3(1), dim3(CHOL_T), 0, c->stream, a);
#endif
}

int launch_chol_solve_blocked(msckf_ctx* c, double* S, const double* Y, double* Kg, const double* z, int zstride) {
    const int d = c->d, dc = c->dc, n1 = GAIN_BLK, n2 = dc - GAIN_BLK;
    double* work1 = ptr<double>(c->dCholWork);
    double* work2 = work1 + (size_t)n1 * (n1 + 1) / 2;
    double* invd = ptr<double>(c->dInvd);
    double* W = ptr<double>(c->dD);            // [n2][n1]  (D is written after the solve)
    double* C2 = ptr<double>(c->dB2);          // [n2][n2]  (B2 likewise)
    int* status = ptr<int>(c->dStatus);
    {   // A = L11 L11^T
        CholArgs a{};
        a.S = S; a.lds_ = dc; a.L = ptr<double>(c->dL); a.U = ptr<double>(c->dU); a.invd = invd; a.n = n1;
        a.work = work1; a.status = status;
        launch_chol_small(c, a);
    }
    // W = B L11^-T  (rows n1.. of S, first n1 columns), then the Schur complement C2 = C - W W^T
    HIPCHK(c, hipMemcpy2DAsync(W, (size_t)n1 * 8, S + (size_t)n1 * dc, (size_t)dc * 8, (size_t)n1 * 8, n2, hipMemcpyDeviceToDevice,
                               c->stream));
    launch_tri_sweep<1>(c, W, n1, n2, work1, invd, n1);
    gemm(c, W, n1, W, n1, S + (size_t)n1 * dc + n1, dc, C2, n2, n2, n2, n1, -1.0, 1.0, 0.0, 1, 0);
    {   // C2 = L22 L22^T
        CholArgs a{};
        a.S = C2; a.lds_ = n2; a.L = ptr<double>(c->dL); a.U = ptr<double>(c->dU); a.invd = invd + n1; a.n = n2;
        a.work = work2; a.status = status + 1;
        launch_chol_small(c, a);
    }
    // X1 = Y1 L11^-T (into K)
    HIPCHK(c, hipMemcpy2DAsync(Kg, (size_t)dc * 8, Y, (size_t)dc * 8, (size_t)n1 * 8, d, hipMemcpyDeviceToDevice, c->stream));
    launch_tri_sweep<1>(c, Kg, dc, d, work1, invd, n1);
    // K2 = (Y2 - X1 W^T) L22^-T L22^-1
    gemm(c, Kg, dc, W, n1, Y + n1, dc, Kg + n1, dc, d, n2, n1, -1.0, 1.0, 0.0, 1, 0);
    launch_tri_sweep<3>(c, Kg + n1, dc, d, work2, invd + n1, n2);
    // K1 = (X1 - K2 W) L11^-1
    gemm(c, Kg + n1, dc, W, n1, Kg, dc, Kg, dc, d, n1, n2, -1.0, 1.0, 0.0, 0, 0);
    launch_tri_sweep<2>(c, Kg, dc, d, work1, invd, n1);
    // dx = K r_n
    hipLaunchKernelGGL(k_matvec, dim3((d + 3) / 4), dim3(256), 0, c->stream, Kg, dc, dc, z, zstride, ptr<double>(c->dDx), d);
    HIPCHK(c, hipGetLastError());
    return MSCKF_OK;
}

// K6-K7 from the root block [T | r_n] (dc x (dc+1), row-major) and the prior P.
int launch_gain(msckf_ctx* c, const double* Tblk) {
    const int d = c->d, dc = c->dc, ldt = dc + 1;
    const double s2 = c->sigma * c->sigma;
    const double* P = ptr<double>(c->dP);
    double* Y = ptr<double>(c->dY);     // [d][dc]   Y = P[:,15:] T^T
    double* S = ptr<double>(c->dS);     // [dc][dc]
    double* Kg = ptr<double>(c->dK);    // [d][dc]
    double* B2 = ptr<double>(c->dB2);   // [d][d]    (I - K T) P
    double* D = ptr<double>(c->dD);     // [d][dc]   sigma^2 K - B2[:,15:] T^T
    double* Pn = ptr<double>(c->dPn);   // [d][d]
    // Y = P[:, 15:] T^T                      (P T_H^T, MSCKF.py:606)
    gemm(c, P + 15, d, Tblk, ldt, nullptr, 0, Y, dc, d, dc, dc, 1.0, 0.0, 0.0, 1, 1);
    // S = T Y[15:, :] + sigma^2 I            (MSCKF.py:605)
    gemm(c, Tblk, ldt, Y + (size_t)15 * dc, dc, nullptr, 0, S, dc, dc, dc, dc, 1.0, 0.0, s2, 0, 2);
    c->gain_blocked = dc > 4 * CHOL_TILE_MAX_NT && dc <= 2 * GAIN_BLK && dc - GAIN_BLK >= 4;
    if (c->gain_blocked) {
        if (int rcb = launch_chol_solve_blocked(c, S, Y, Kg, Tblk + dc, ldt)) return rcb;
    } else {
    // S = L L^T
    bool packed_L = false;
    {
        CholArgs a{};
        a.S = S; a.lds_ = dc; a.L = ptr<double>(c->dL); a.U = ptr<double>(c->dU); a.invd = ptr<double>(c->dInvd);
        a.n = dc; a.work = ptr<double>(c->dCholWork); a.status = ptr<int>(c->dStatus);
        if (dc <= 4 * CHOL_TILE_MAX_NT) {
            packed_L = true;
            launch_chol_small(c, a);   // matrix in registers
        } else {
            const size_t need = (size_t)dc * (dc + 1) / 2 * 8;
            a.use_lds = need <= (size_t)(LDS_MAX_BYTES - 1024) ? 1 : 0;
            hipLaunchKernelGGL(k_chol<512>, dim3(1), dim3(512), a.use_lds ? need : 0, c->stream, a);
        }
    }
    // K = Y S^-1, dx = K r_n                 (MSCKF.py:606-607)
    {
        SolveArgs a{};
        a.Y = Y; a.ldy = dc; a.L = ptr<double>(c->dL); a.U = ptr<double>(c->dU); a.invd = ptr<double>(c->dInvd);
        a.Lp = packed_L ? ptr<double>(c->dCholWork) : nullptr;
        a.n = dc; a.z = Tblk + dc; a.zstride = ldt; a.Kg = Kg; a.ldk = dc; a.dx = ptr<double>(c->dDx); a.d = d;
        const int nreg = (dc + 63) / 64;
        const size_t lds_need = ((size_t)dc * (dc + 1) / 2 + dc) * 8;
        if (nreg <= 3 && lds_need <= (size_t)(LDS_MAX_BYTES - 1024)) {
            constexpr int WV = SOLVE_WAVES;
            const dim3 grid((d + WV * SOLVE_ROWS - 1) / (WV * SOLVE_ROWS)), block(64 * WV);
#define SOLVE_LAUNCH(NR, UN) hipLaunchKernelGGL((k_solve_lds<NR, WV, SOLVE_ROWS, UN>), grid, block, lds_need, c->stream, a)
            if (packed_L) {
                if (nreg <= 1) SOLVE_LAUNCH(1, true); else if (nreg <= 2) SOLVE_LAUNCH(2, true); else SOLVE_LAUNCH(3, true);
            } else {
                if (nreg <= 1) SOLVE_LAUNCH(1, false); else if (nreg <= 2) SOLVE_LAUNCH(2, false); else SOLVE_LAUNCH(3, false);
            }
#undef SOLVE_LAUNCH
        } else if (nreg <= 4) hipLaunchKernelGGL(k_solve<4>, dim3(d), dim3(64), 0, c->stream, a);
        else hipLaunchKernelGGL(k_solve<5>, dim3(d), dim3(64), 0, c->stream, a);
    }
    }
    if (c->cfg.dtype == MSCKF_DTYPE_F32) {
        // Joseph form on the f32 matrix cores: fp32 operands (the fp64 P, K, Y, T are rounded as they are loaded),
        // fp32 intermediates B2, D, Pn, fp32 accumulation; P_out leaves as double
        auto gemm32 = [&](const void* A, int lda, int af, const void* B, int ldb, int bf, const void* C0, int ldc0, int c0f,
                          void* C, int ldc, int M_, int N_, int K_, float alpha, float beta, int tri) {
            Gemm32Args g{A, lda, af, B, ldb, bf, C0, ldc0, c0f, C, ldc, 1, M_, N_, K_, alpha, beta, 1, tri};
            const dim3 grid((N_ + 15) / 16, (M_ + 15) / 16);
            if (!af && !c0f) hipLaunchKernelGGL((k_gemm_f32<false, false>), grid, dim3(64 * GEMM_WAVES), 0, c->stream, g);
            else if (af && !c0f) hipLaunchKernelGGL((k_gemm_f32<true, false>), grid, dim3(64 * GEMM_WAVES), 0, c->stream, g);
            else hipLaunchKernelGGL((k_gemm_f32<true, true>), grid, dim3(64 * GEMM_WAVES), 0, c->stream, g);
        };
        float* B2f = reinterpret_cast<float*>(B2);
        float* Df = reinterpret_cast<float*>(D);
        float* Pnf = reinterpret_cast<float*>(Pn);
        gemm32(Kg, dc, 0, Y, dc, 0, P, d, 0, B2f, d, d, d, dc, -1.f, 1.f, 0);                     // B2 = P - K Y^T
        gemm32(B2f + 15, d, 1, Tblk, ldt, 0, Kg, dc, 0, Df, dc, d, dc, dc, -1.f, (float)s2, 1);   // D = s2 K - B2[:,15:] T^T
        gemm32(Df, dc, 1, Kg, dc, 0, B2f, d, 1, Pnf, d, d, d, dc, 1.f, 1.f, 0);                   // Pn = B2 + D K^T
        hipLaunchKernelGGL(k_symmetrize_f32, dim3((d + 15) / 16, (d + 15) / 16), dim3(16, 16), 0, c->stream, Pnf,
                           ptr<double>(c->dPout), d, d);
        HIPCHK(c, hipGetLastError());
        return MSCKF_OK;
    }
    // Joseph form (MSCKF.py:613) with T_H = [0 | T], expanded:  (I - K T_H) P (I - K T_H)^T + sigma^2 K K^T
    //   = P - K Y^T - Y K^T + K S K^T = P - K Y^T + W K^T,   W = K S - Y   (S = T_H P T_H^T + sigma^2 I as factored above)
    gemm(c, Kg, dc, S, dc, Y, dc, D, dc, d, dc, dc, 1.0, -1.0, 0.0, 0, 0);
    //   P_out = (Pn + Pn^T) / 2                         (MSCKF.py:614), both tiles of a mirror pair in one workgroup
    {
        JosephArgs j{P, d, Kg, Y, D, ptr<double>(c->dPout), d, d, dc};
        const int nt = (d + 15) / 16;
        hipLaunchKernelGGL(k_joseph_f64, dim3(nt, nt), dim3(64 * GEMM_WAVES), 0, c->stream, j);
    }
    HIPCHK(c, hipGetLastError());
    (void)B2; (void)Pn;
    return MSCKF_OK;
}

// ---- K6-K7 as the sequential block update of k_gstream.h ---------------------------------------------------------
bool gstream_ok(const msckf_ctx* c, int band) {
    // (dtype f32 = fp32 STORAGE of the K4 stack and the P-update's rank-16 products on the f32 matrix cores inside the
    //  sequential block update -- in place of the f32-MFMA Joseph launches of rounds 2-3, 200 us behind the sweep at N = 50;
    //  MSCKF_GAIN_STREAM=0 brings those back)
    if (!c->gs_enabled || c->dc < 1) return false;
    const int nb = (c->dc + 15) / 16, ns = nb + 1;
    if (ns > GS_MAX_NS) return false;
    const int ncb = c->wide_active ? nb : gstream_ncb(c->dc, band);      // (the remainder blocks' root is dense)
    return gstream_lds_doubles(ns, ncb) * 8 <= (size_t)(LDS_MAX_BYTES - 1024);
}
// Tblk: the root block [T | r_n]; band: its widest row in columns
void fill_gstream_args(msckf_ctx* c, GStreamArgs& a, const double* Tblk, int band, bool beside) {
    const int d = c->d, dc = c->dc, nb = (dc + 15) / 16;
    a = GStreamArgs{};
    a.P = ptr<double>(c->dP); a.ldp = d;
    a.T = Tblk; a.ldt = dc + 1;
    a.progress = beside ? ptr<unsigned long long>(c->dGsProg) : nullptr;
    a.epoch = c->gs_epoch;
    a.ex = ptr<double>(c->dGsEx); a.exflag = ptr<unsigned long long>(c->dGsFlag);
    a.dx = ptr<double>(c->dDx); a.Pout = ptr<double>(c->dPout); a.ldo = d;
    a.status = ptr<int>(c->dStatus);
    if (c->res_direct && c->want_direct) {
        char* h = static_cast<char*>(c->hRes);
        a.status_h = reinterpret_cast<int*>(h); a.dx_h = reinterpret_cast<double*>(h + c->res_dx_off);
        a.Pout_h = reinterpret_cast<double*>(h + c->res_p_off);
        reinterpret_cast<int*>(h)[0] = 3; reinterpret_cast<int*>(h)[1] = 0; reinterpret_cast<int*>(h)[2] = 0;     // 3: not written
    }
    a.sigma2 = c->sigma * c->sigma;
    a.d = d; a.dc = dc; a.nb = nb; a.ns = nb + 1; a.ncb = gstream_ncb(dc, band);
    a.nb1 = Tblk ? nb : 0;
    // (dtype = f32: the rank-16 products of the P-update on the f32 matrix cores -- but not for a batch with split long tracks: its
    //  dense remainder rows add tens of row blocks, every one an fp32-rounded product against a covariance that keeps shrinking;
    //  tools/soak_holes.py 150 8 f32, (48, 370, <= 22 views): dx off by 2.5e-4 with them, against the mode's 1e-4)
    a.f32_update = (c->cfg.dtype == MSCKF_DTYPE_F32 && !(c->wide_active && !c->in_merge)) ? 1 : 0;
    if (c->wide_active && c->rem_direct && !c->t2_early && !c->in_merge) {
        a.T2 = ptr<double>(c->dRem); a.ldt2 = dc + 1; a.nb2 = (c->rem_cap + 15) / 16;
        a.nb2_dev = reinterpret_cast<const int*>(ptr<double>(c->dRem) + (size_t)16 * GS_MAX_NB2 * (6 * c->maxN + 1));
    }
    if (c->t2_early && !c->in_merge) { a.P = ptr<double>(c->dPout); a.ldp = d; a.dx0 = ptr<double>(c->dDx); }
    a.stamps = nullptr;
    a.tstamp = c->gs_stamp ? ptr<long long>(c->dGsProg) + 32 : nullptr;
}
// K6-K7 behind a complete root block: nothing is polled
int launch_gain_stream(msckf_ctx* c, const double* Tblk, int band) {
    ++c->gs_epoch;
    GStreamArgs a;
    fill_gstream_args(c, a, Tblk, band, false);
    const size_t lds = gstream_lds_doubles(a.ns, a.nb2 > 0 ? a.nb : a.ncb) * 8;
    if (a.ns <= GS_WAVES) hipLaunchKernelGGL(k_gain_stream<1>, dim3(a.ns), dim3(64 * GS_WAVES), lds, c->stream, a);
    else hipLaunchKernelGGL(k_gain_stream<2>, dim3(a.ns), dim3(64 * GS_WAVES), lds, c->stream, a);
    HIPCHK(c, hipGetLastError());
    c->gain_blocked = false;
    return MSCKF_OK;
}
// K6-K7 on a dense source of rows alone: two row blocks per exchange where the window allows it (k_gdense.h: 7.1 -> 6.1 us per
// block at N = 30; MSCKF_GAIN_DENSE=0: the one-block kernel)
void launch_gain_dense_rows(msckf_ctx* c, const GStreamArgs& a, hipStream_t st) {
    static const bool pairs = [] { const char* e = std::getenv("MSCKF_GAIN_DENSE"); return !e || std::atoi(e) != 0; }();
    const size_t ldsp = gdense_lds_doubles(a.ns, a.nb) * 8;
    if (pairs && a.nb1 == 0 && a.nb2 > 0 && a.ns <= 14 && ldsp <= (size_t)(LDS_MAX_BYTES - 1024)) {
        hipLaunchKernelGGL(k_gain_dense, dim3(a.ns), dim3(64 * GS_WAVES), ldsp, st, a);
        return;
    }
    const size_t lds = gstream_lds_doubles(a.ns, a.nb) * 8;
    if (a.ns <= GS_WAVES) hipLaunchKernelGGL(k_gain_stream<1>, dim3(a.ns), dim3(64 * GS_WAVES), lds, st, a);
    else hipLaunchKernelGGL(k_gain_stream<2>, dim3(a.ns), dim3(64 * GS_WAVES), lds, st, a);
}
// MANY dense remainder rows (more than the root sweep's ~110 us cover at ~9 us per block on the nine-wavefront strips of its
// launch): the update on them runs as a launch of its own, sixteen wavefronts per strip (~7 us per block), on the stream that made
// them -- beside the band pipeline's leaves, which do not touch P -- from the prior P into P_out / dx (status word 1); the update on
// the band root, inside the root sweep's launch, then starts from those.
int launch_gain_t2_early(msckf_ctx* c, hipStream_t st) {
    ++c->gs_epoch;
    GStreamArgs a;
    const bool keep = c->t2_early;
    c->t2_early = false;
    fill_gstream_args(c, a, nullptr, c->dc, false);        // (T2 = the dense rows, no first source)
    c->t2_early = keep;
    a.status = ptr<int>(c->dStatus) + 1;
    a.status_h = nullptr; a.dx_h = nullptr; a.Pout_h = nullptr; a.tstamp = nullptr;
    launch_gain_dense_rows(c, a, st);
    HIPCHK(c, hipGetLastError());
    return MSCKF_OK;
}
// K6-K7 on a SECOND source of rows -- the root of the remainder blocks' tree (split long tracks) -- behind the update on the
// first: the rows of both sources are measurements with independent sigma^2 noise, so the update on all of them is the
// update on the second against the covariance the first left (P = P_out, in place: a strip's workgroup stores its column
// strip after the last exchange, which every workgroup takes part in only after it has loaded its tiles), with the dx row
// carried on.  Its status goes to word 1.
int launch_gain_chain(msckf_ctx* c, const double* Tblk) {
    ++c->gs_epoch;
    GStreamArgs a;
    fill_gstream_args(c, a, Tblk, c->dc, false);
    a.P = ptr<double>(c->dPout); a.ldp = c->d; a.dx0 = ptr<double>(c->dDx);
    a.T2 = nullptr; a.nb2 = 0; a.nb2_dev = nullptr;
    a.status = ptr<int>(c->dStatus) + 1;
    a.status_h = nullptr; a.dx_h = nullptr; a.Pout_h = nullptr; a.tstamp = nullptr;
    const size_t lds = gstream_lds_doubles(a.ns, a.ncb) * 8;
    if (a.ns <= GS_WAVES) hipLaunchKernelGGL(k_gain_stream<1>, dim3(a.ns), dim3(64 * GS_WAVES), lds, c->stream, a);
    else hipLaunchKernelGGL(k_gain_stream<2>, dim3(a.ns), dim3(64 * GS_WAVES), lds, c->stream, a);
    HIPCHK(c, hipGetLastError());
    return MSCKF_OK;
}
// ... on the triangles a cut remainder tree ended with (k_tri_gather laid them down in dRem): a dense second source, nothing else
int launch_tri_gather(msckf_ctx* c, hipStream_t st) {
    TriGatherArgs g{};
    g.rbuf = ptr<double>(c->dRbuf); g.dst = ptr<double>(c->dRem);
    g.dc = c->dc; g.n = (int)c->rtops.size(); g.rows = c->rtop_rows; g.rows_pad = (c->rtop_rows + 15) / 16 * 16;
    if (g.n < 1 || g.n > TRI_GATHER_MAX || g.rows_pad > 16 * GS_MAX_NB2) { c->last_error = "remainder tree: too many triangles for K6-K7"; return MSCKF_ERR_STATE; }
    int r0 = 0;
    for (int k = 0; k < g.n; ++k) {
        const FoldNode& nd = c->rnodes[c->rtops[k]];
        g.off[k] = nd.out_off; g.w[k] = nd.w; g.col0[k] = 6 * nd.win_lo; g.row0[k] = r0;
        r0 += nd.w;
    }
    g.row0[g.n] = r0;
    const long long total = (long long)g.rows_pad * (c->dc + 1);
    hipLaunchKernelGGL(k_tri_gather, dim3((unsigned)std::min<long long>((total + 255) / 256, 1024)), dim3(256), 0, st, g);
    HIPCHK(c, hipGetLastError());
    return MSCKF_OK;
}
int launch_gain_chain_dense(msckf_ctx* c) {
    ++c->gs_epoch;
    GStreamArgs a;
    fill_gstream_args(c, a, nullptr, c->dc, false);
    a.P = ptr<double>(c->dPout); a.ldp = c->d; a.dx0 = ptr<double>(c->dDx);
    a.T2 = ptr<double>(c->dRem); a.ldt2 = c->dc + 1; a.nb2 = (c->rtop_rows + 15) / 16; a.nb2_dev = nullptr;
    a.status = ptr<int>(c->dStatus) + 1;
    a.status_h = nullptr; a.dx_h = nullptr; a.Pout_h = nullptr; a.tstamp = nullptr;
    launch_gain_dense_rows(c, a, c->stream);
    HIPCHK(c, hipGetLastError());
    return MSCKF_OK;
}
// The root sweep (k_sweep form) and K6-K7 in ONE launch (k_root_gain): workgroup 0 sweeps and publishes the rows of the
// root block as they become final, workgroups 1.. are the strips of the update.  `sa` carries the tables and the node index.
bool root_gain_ok(const msckf_ctx* c, int band) {
    return gstream_ok(c, band) && c->gs_overlap && (c->dc + 15) / 16 + 1 <= 2 * SWEEP_NW &&    // two tiles on each fold-slot wavefront
           2 * (2 + (c->dc + 15) / 16) <= c->n_cu;                                             // (the sweep and the strips wait for each other: all resident)
}
// A merge level that rides in the root's launch (k_gstream.h): its nodes, where their flush tables sit on the device
// ([count offsets | tables]) and the root's step-0 requirement count behind its own tables (sweep_gate_table).
struct MergeRide { int node_base, count, nf, n_gate; const int* flush; size_t lds; };    // lds: what its largest node asks for
int launch_root_and_gain(msckf_ctx* c, SweepArgs sa, int wtot, int nsteps, const int* flush_tab, const double* Tblk, int band,
                         const MergeRide* ride = nullptr) {
    ++c->gs_epoch;
    sa.flush_tab = flush_tab;
    sa.progress = ptr<unsigned long long>(c->dGsProg);
    sa.epoch = c->gs_epoch;
    sa.stamps = nullptr;
    sa.tstamp = c->gs_stamp ? ptr<long long>(c->dGsProg) + 32 : nullptr;      // (behind the progress word, same allocation)
    GStreamArgs ga;
    fill_gstream_args(c, ga, Tblk, band, true);
    // (every workgroup asks for more than half of a CU's LDS: one per CU, the sweep has its CU to itself)
    size_t lds = std::max<size_t>(std::max(sweep_lds_bytes_fl(wtot, SWEEP_NW, nsteps), gstream_lds_doubles(ga.ns, ga.nb2 > 0 ? ga.nb : ga.ncb) * 8),
                                  (size_t)84 * 1024);
    SweepArgs ma{};
    int nm = 0;
    bool mid = false;
    if (ride) {                  // a merge level: workgroups behind the strips, streaming their rows to the root
        nm = ride->count;
        ma = sa;
        ma.node_base = ride->node_base;
        ma.flush_off = ride->flush; ma.flush_tab = ride->flush + nm;
        ma.progress = ptr<unsigned long long>(c->dMProg); ma.prog_stride = 1; ma.pub_shift = 3;
        ma.tstamp = nullptr;
        sa.src_progress = ptr<unsigned long long>(c->dMProg); sa.n_prod = nm; sa.n_gate = ride->n_gate;
        lds = std::max(lds, sweep_lds_bytes_fl(wtot, SWEEP_NW, nsteps, ride->n_gate));
        mid = ride->nf == SWEEP_NW_MID;
        lds = std::max(lds, ride->lds);
    }
    if (mid) hipLaunchKernelGGL((k_root_gain_m<SWEEP_NW, SWEEP_NW_MID, 2>), dim3(1 + ga.ns + nm), dim3(64 * (SWEEP_NW_MID + 1)), lds, c->stream, sa, ga, ma);
    else hipLaunchKernelGGL((k_root_gain<SWEEP_NW, 2>), dim3(1 + ga.ns + nm), dim3(64 * (SWEEP_NW + 1)), lds, c->stream, sa, ga, ma);
    HIPCHK(c, hipGetLastError());
    c->gain_blocked = false;
    return MSCKF_OK;
}

// ... and for the ring-buffered root sweeps (k_wsweep form, sweep modes 1 and 2): k_root_gain_w
bool root_gain_w_ok(const msckf_ctx* c, int band) {
    return gstream_ok(c, band) && c->gs_overlap && (c->dc + 15) / 16 + 1 <= 3 * (SWEEP_NW - 1) &&
           2 * (2 + (c->dc + 15) / 16) <= c->n_cu;                                             // (as root_gain_ok: the sweep and the strips wait for each other)
}
template <int CS>
int launch_root_and_gain_w(msckf_ctx* c, int node, int nsteps, int rc_log2, const double* zero, const double* Tblk, int band) {
    ++c->gs_epoch;
    WSweepArgs a{};
    a.nodes = ptr<SweepNode>(c->dSweepNodes);
    a.folds = ptr<SweepFold>(c->dSweepFolds);
    a.node_base = node;
    a.rbuf = ptr<double>(c->dRbuf);
    a.zero = zero;
    a.flush = ptr<int>(c->dFlush);
    a.flush_off = ptr<int>(c->dFlushOff);
    a.rc_log2 = rc_log2;
    a.progress = ptr<unsigned long long>(c->dGsProg);
    a.epoch = c->gs_epoch;
    a.tstamp = c->gs_stamp ? ptr<long long>(c->dGsProg) + 32 : nullptr;
    GStreamArgs ga;
    fill_gstream_args(c, ga, Tblk, band, true);
    const size_t lds = std::max<size_t>(std::max(wsweep_lds_bytes<CS>(1 << rc_log2, SWEEP_NW, nsteps),
                                                 gstream_lds_doubles(ga.ns, ga.nb2 > 0 ? ga.nb : ga.ncb) * 8), (size_t)84 * 1024);
    if (ga.ns <= 2 * (SWEEP_NW - 1))
        hipLaunchKernelGGL((k_root_gain_w<SWEEP_NW, CS, 2>), dim3(1 + ga.ns), dim3(64 * SWEEP_NW), lds, c->stream, a, ga);
    else
        hipLaunchKernelGGL((k_root_gain_w<SWEEP_NW, CS, 3>), dim3(1 + ga.ns), dim3(64 * SWEEP_NW), lds, c->stream, a, ga);
    HIPCHK(c, hipGetLastError());
    c->gain_blocked = false;
    return MSCKF_OK;
}

// Gate results of the last run, summed on the host:
// {accepted, stacked rows, not-SPD gate matrices, not selected by k_select}.
int gate_counts(msckf_ctx* c, int out[4], std::vector<unsigned char>* acc_sorted, bool copied = false) {
    out[0] = out[1] = out[2] = out[3] = 0;
    if (c->F == 0) return MSCKF_OK;
    const size_t F = c->F, Fs = c->Fs;          // (the blocks of split long tracks sit behind the F tracks: not counted)
    if (!copied) {
        HIPCHK(c, hipMemcpyAsync(c->hGate, c->dGateArena.p, Fs * 5, hipMemcpyDeviceToHost, c->stream));   // rank | accepted
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    const int* rk = static_cast<const int*>(c->hGate);
    const unsigned char* acc = static_cast<const unsigned char*>(c->hGate) + Fs * 4;
    for (int s = 0; s < c->F; ++s) {
        if (acc[s] == 1) { out[0]++; out[1] += 2 * (c->h_view_sorted[s + 1] - c->h_view_sorted[s]) - rk[s]; }
        else if (acc[s] == 2) out[2]++;
        else if (acc[s] == 3) out[3]++;
    }
    if (acc_sorted) acc_sorted->assign(acc, acc + F);
    return MSCKF_OK;
}

const double* root_block(msckf_ctx* c) { return ptr<double>(c->dRbuf) + c->root_off; }

int run_pipeline(msckf_ctx* c, bool with_gain, hipEvent_t* stage_ev) {
    if (!c->have_state || !c->have_features) return MSCKF_ERR_STATE;
    int rc;
    if (stage_ev) HIPCHK(c, hipEventRecord(stage_ev[0], c->stream));
    if (c->state_pending) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_state, 0)); c->state_pending = false; }
    if (!c->feature_launched && (rc = launch_feature(c)) != MSCKF_OK) return rc;
    c->feature_launched = false;
    if (stage_ev) HIPCHK(c, hipEventRecord(stage_ev[1], c->stream));
    // split long tracks: their remainder blocks' tree follows k_feature<64, true> on the second stream, beside the band
    // pipeline; the leaves below read the narrow blocks that kernel wrote
    const bool chain = c->F > 0 && c->wide_active && (c->rroot >= 0 || !c->rtops.empty());
    if (chain) {
        hipStream_t rs = c->wide_on_stream2 ? c->stream2 : c->stream;
        // (the tree reads the plan's tables and writes the workspace: behind their upload / memset, which the main stream is
        //  behind by now -- ev_plan, set_features)
        if (c->wide_on_stream2) { HIPCHK(c, hipEventRecord(c->ev_fork, c->stream)); HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0)); }
        if ((rc = launch_fold_levels(c, c->rlevels, c->rnodes, rs, ptr<FoldNode>(c->dRNodes))) != MSCKF_OK) return rc;
        if (!c->rtops.empty() && (rc = launch_tri_gather(c, rs)) != MSCKF_OK) return rc;
        if (c->wide_on_stream2) HIPCHK(c, hipEventRecord(c->ev_rem, c->stream2));
    }
    const bool early = c->F > 0 && c->wide_on_stream2 && c->band_plan && c->n_leaves0 > 0 && c->n_leaves0 < c->n_leaves;
    if (early && (rc = launch_leaves_band(c, 0, c->n_leaves0)) != MSCKF_OK) return rc;          // (the short tracks' leaves need not wait)
    if (c->F > 0 && c->wide_on_stream2) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_wfeat, 0));
    if (c->F > 0 && c->band_plan && (rc = launch_leaves_band(c, early ? c->n_leaves0 : 0)) != MSCKF_OK) return rc;
    if (c->F > 0 && !c->band_plan && (rc = launch_fold_levels(c, c->levels, c->nodes)) != MSCKF_OK) return rc;
    // (a rank that exports its group triangles stops in front of the root sweep: rank 0 runs it over all shards)
    // K6-K7 beside the root sweep: the band plan's k_sweep root with the flusher wavefront and the update's strips in ONE launch
    const bool direct = c->F > 0 && c->wide_active && c->rem_direct;      // (the dense remainder rows: K6-K7's second source, taken first)
    const bool have_rows = c->root >= 0 || chain || direct;
    {
        static const int early_min = [] { const char* e = std::getenv("MSCKF_T2_EARLY_MIN"); return e ? std::atoi(e) : 20; }();
        c->t2_early = direct && with_gain && c->root >= 0 && gstream_ok(c, c->root_band) && (c->rem_cap + 15) / 16 >= early_min;
        if (c->t2_early) {
            hipStream_t rs = c->wide_on_stream2 ? c->stream2 : c->stream;
            if ((rc = launch_gain_t2_early(c, rs)) != MSCKF_OK) return rc;
            if (c->wide_on_stream2) HIPCHK(c, hipEventRecord(c->ev_rem, c->stream2));
        }
    }
    const bool gs = with_gain && c->F > 0 && have_rows && gstream_ok(c, c->root_band);
    if (chain && with_gain && !gs) { c->last_error = "split long tracks need the streamed K6-K7"; return MSCKF_ERR_STATE; }
    // (behind an early update on the dense remainder rows the root sweep runs as a launch of its own, BESIDE that update, and the update
    //  on its rows follows both: in one launch with it the sweep would wait for the remainder rows' update too -- a frame of 300 tracks
    //  ~ U[2, 30]: 695 -> ~610 us)
    static const bool t2_split = [] { const char* e = std::getenv("MSCKF_T2_SPLIT_ROOT"); return !e || std::atoi(e) != 0; }();
    const bool beside = gs && c->band_plan && c->root >= 0 && !(c->t2_early && t2_split) &&
                        (c->sweep_mode == 0 ? (!c->h_root_flush.empty() && root_gain_ok(c, c->root_band)) : root_gain_w_ok(c, c->root_band));
    const bool streamed = beside && c->sweep_mode == 0 && c->root_streamed;     // the last merge level rides in the root's launch
    if (c->F > 0 && c->band_plan && (rc = launch_sweeps(c, (with_gain || !c->xchg_planned) && !beside, streamed ? c->stream_level : -1)) != MSCKF_OK) return rc;
    if (c->F > 0 && c->xchg_planned && !with_gain) {       // the accepted count rides in the export record (double N)
        // (with msckf_set_exchange_mask the shard's gate bytes ride behind it, in input order)
        hipLaunchKernelGGL(k_count_accepted, dim3(1), dim3(256), 0, c->stream, ptr<unsigned char>(c->dAcc), c->F,
                           ptr<double>(c->dRbuf) + c->N, ptr<int>(c->dPerm),
                           c->xmask_doubles > 0 ? reinterpret_cast<unsigned char*>(ptr<double>(c->dRbuf) + c->N + 1) : nullptr);
        HIPCHK(c, hipGetLastError());
    }
    if (c->t2_early && c->wide_on_stream2) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_rem, 0));     // (its P_out / dx are what the update below starts from)
    c->gs_fused_last = beside;
    c->res_direct = c->want_direct && c->gate_direct && gs && !chain;       // (a second update behind the first writes the HBM copies only)
    if (beside && c->sweep_mode > 0) {
        const SweepNode& rn = c->snodes.back();
        const double* zero = ptr<double>(c->dRbuf) + c->zero_off;
        if (c->sweep_mode == 1) rc = launch_root_and_gain_w<4>(c, c->n_group_merges, rn.nsteps, WS_RC_LOG2_4, zero, root_block(c), c->root_band);
        else rc = launch_root_and_gain_w<6>(c, c->n_group_merges, rn.nsteps, WS_RC_LOG2_6, zero, root_block(c), c->root_band);
        if (rc != MSCKF_OK) return rc;
        if (stage_ev) HIPCHK(c, hipEventRecord(stage_ev[2], c->stream));
    } else if (beside) {
        SweepArgs a{};
        a.nodes = ptr<SweepNode>(c->dSweepNodes);
        a.folds = ptr<SweepFold>(c->dSweepFolds);
        a.rbuf = ptr<double>(c->dRbuf);
        a.zero = ptr<double>(c->dRbuf) + c->zero_off;
        a.node_base = c->n_group_merges;
        // (one launch: the stage boundary K5 | K6-K7 is not observable; the events report the pair under K5 and only what
        //  trails the launch -- nothing -- under K6-K7)
        MergeRide ride{};
        if (streamed) {
            const auto& lv = c->sweep_levels[c->stream_level];
            ride = MergeRide{lv.first, lv.second, c->sweep_level_nf[c->stream_level], c->root_n_gate, ptr<int>(c->dRootFlush) + c->mflush_at, 0};
            for (int i = lv.first; i < lv.first + lv.second; ++i)
                ride.lds = std::max(ride.lds, sweep_lds_bytes_fl(c->snodes[i].wtot, ride.nf, c->snodes[i].nsteps));
        }
        if ((rc = launch_root_and_gain(c, a, c->snodes.back().wtot, c->snodes.back().nsteps, ptr<int>(c->dRootFlush), root_block(c),
                                       c->root_band, streamed ? &ride : nullptr)) != MSCKF_OK) return rc;
        if (stage_ev) HIPCHK(c, hipEventRecord(stage_ev[2], c->stream));
    } else {
        if (stage_ev) HIPCHK(c, hipEventRecord(stage_ev[2], c->stream));
        if (with_gain && c->F > 0 && have_rows) {
            if (gs) rc = launch_gain_stream(c, c->root >= 0 ? root_block(c) : nullptr, c->root_band);
            else rc = launch_gain(c, root_block(c));
            if (rc != MSCKF_OK) return rc;
        }
    }
    if (chain && with_gain) {              // the remainder blocks' rows: a second update behind the first
        if (c->wide_on_stream2) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_rem, 0));
        if ((rc = c->rtops.empty() ? launch_gain_chain(c, ptr<double>(c->dRbuf) + c->rroot_off) : launch_gain_chain_dense(c)) != MSCKF_OK) return rc;
    }
    if (stage_ev) HIPCHK(c, hipEventRecord(stage_ev[3], c->stream));
    c->ran = true;
    c->ran_gain = with_gain;
    c->acc_override = -1;
    c->acc_from_dev = false;
    ++c->run_serial; c->run_pending = true;
    if (c->res_direct) c->direct_serial = c->run_serial;
    return MSCKF_OK;
}

}  // namespace

extern "C" {

const char* msckf_strerror(int code) {
    switch (code) {
        case MSCKF_OK: return "ok";
        case MSCKF_NOOP: return "no-op: no feature passed the gate";
        case MSCKF_ERR_ARG: return "bad argument or size over the context capacity";
        case MSCKF_ERR_HIP: return "HIP runtime error";
        case MSCKF_ERR_NO_DEVICE: return "no usable gfx950 device";
        case MSCKF_ERR_NOT_SPD: return "innovation covariance not positive definite";
        case MSCKF_ERR_STATE: return "call order: set_state and set_features must precede run";
        case MSCKF_ERR_DUP_SLOT: return "a track observes the same clone slot twice";
        case MSCKF_ERR_COMM: return "RCCL error";
        default: return "unknown";
    }
}

const char* msckf_last_error(const msckf_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int msckf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int msckf_create(msckf_ctx** out, const msckf_config* cfg) {
    if (!out || !cfg || cfg->abi_version != MSCKF_ABI_VERSION) return MSCKF_ERR_ARG;
    if (cfg->max_track < 1 || cfg->max_track > MSCKF_MAX_TRACK || cfg->max_clones < 1 || cfg->max_features < 0)
        return MSCKF_ERR_ARG;
    if (cfg->dtype != MSCKF_DTYPE_F64 && cfg->dtype != MSCKF_DTYPE_F32) return MSCKF_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || cfg->device < 0 || cfg->device >= n) return MSCKF_ERR_NO_DEVICE;
    msckf_ctx* c = new msckf_ctx();
    c->cfg = *cfg;
    c->device = cfg->device;
    c->maxN = cfg->max_clones; c->maxF = cfg->max_features; c->maxM = cfg->max_track;
    if (hipSetDevice(c->device) != hipSuccess) { delete c; return MSCKF_ERR_NO_DEVICE; }
    hipDeviceProp_t prop{};
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) { delete c; return MSCKF_ERR_NO_DEVICE; }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) { delete c; return MSCKF_ERR_NO_DEVICE; }
    c->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return MSCKF_ERR_HIP; }
    if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) { msckf_destroy(c); return MSCKF_ERR_HIP; }
    if (hipStreamCreateWithFlags(&c->stream_up, hipStreamNonBlocking) != hipSuccess) { msckf_destroy(c); return MSCKF_ERR_HIP; }
    {
        const char* e1 = std::getenv("MSCKF_GAIN_STREAM");
        const char* e2 = std::getenv("MSCKF_GAIN_OVERLAP");
        c->gs_enabled = !(e1 && std::atoi(e1) == 0);
        { const char* e3 = std::getenv("MSCKF_WIDE_STREAM"); c->wide_concurrent = !(e3 && std::atoi(e3) == 0); }
        c->gs_overlap = !(e2 && std::atoi(e2) == 0);
        { const char* e4 = std::getenv("MSCKF_REM_LEAF_ROWS"); if (e4 && std::atoi(e4) >= 16) c->rem_leaf_rows = std::atoi(e4); }
        { const char* e5 = std::getenv("MSCKF_REM_CUT_ROWS"); if (e5 && std::atoi(e5) >= 0) c->rem_cut_rows = std::atoi(e5); }
    }
    // every failure here is reported at create time (a dropped attribute would only surface later as an
    // opaque launch error of the first kernel that needs the LDS)
    hipError_t cerr = hipSuccess;
    const char* cwhat = "";
    auto CK = [&](hipError_t e, const char* what) { if (cerr == hipSuccess && e != hipSuccess) { cerr = e; cwhat = what; } };
    for (auto& e : c->ev) CK(hipEventCreate(&e), "hipEventCreate");
    CK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming), "hipEventCreate");
    CK(hipEventCreateWithFlags(&c->ev_wfeat, hipEventDisableTiming), "hipEventCreate");
    CK(hipEventCreateWithFlags(&c->ev_rem, hipEventDisableTiming), "hipEventCreate");
    CK(hipEventCreateWithFlags(&c->ev_plan, hipEventDisableTiming), "hipEventCreate");
    CK(hipEventCreateWithFlags(&c->ev_plan_up, hipEventDisableTiming), "hipEventCreate");
    CK(hipEventCreateWithFlags(&c->ev_state, hipEventDisableTiming), "hipEventCreate");
    CK(hipEventCreateWithFlags(&c->ev_gate, hipEventDisableTiming), "hipEventCreate");
    if (const char* e = std::getenv("MSCKF_DIRECT_RESULT")) c->direct_enabled = std::atoi(e) != 0;
    if (const char* e = std::getenv("MSCKF_ROOT_STREAM")) c->stream_enabled = std::atoi(e) != 0;
    auto lds_attr = [&](const void* f, int bytes, const char* what) {
        CK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), what);
    };
    // kernels that use more than the default 64 KiB of dynamic LDS
    {
#define FK(RPT, CPT) reinterpret_cast<const void*>(&k_fold<FOLD_T, FOLD_RL, RPT, CPT>)
        const void* fk[] = {FK(FOLD_RPT_LEAF, FOLD_CPT1), FK(FOLD_RPT_BIG, FOLD_CPT1), FK(FOLD_RPT_LEAF, FOLD_CPT2),
                            FK(FOLD_RPT_BIG, FOLD_CPT2), FK(FOLD_RPT_W3, FOLD_CPT3)};
#undef FK
        for (const void* f : fk) lds_attr(f, FOLD_LDS_BYTES, "k_fold LDS attribute");
    }
    lds_attr(reinterpret_cast<const void*>(&k_fold_g<FOLDG_T, 12, 6>), FOLD_LDS_BYTES, "k_fold_g LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_fold_g<FOLDG_T, 6, 10>), FOLD_LDS_BYTES, "k_fold_g LDS attribute");
    {
#define SK(NR, UN) reinterpret_cast<const void*>(&k_solve_lds<NR, SOLVE_WAVES, SOLVE_ROWS, UN>)
        const void* sk[] = {SK(1, true), SK(2, true), SK(3, true), SK(1, false), SK(2, false), SK(3, false)};
#undef SK
        for (const void* f : sk) lds_attr(f, LDS_MAX_BYTES - 1024, "k_solve_lds LDS attribute");
#define SM(NR, MD) reinterpret_cast<const void*>(&k_solve_lds<NR, SOLVE_WAVES, 1, true, MD>)
        const void* sm[] = {SM(1, 1), SM(2, 1), SM(3, 1), SM(1, 2), SM(2, 2), SM(3, 2), SM(1, 3), SM(2, 3), SM(3, 3)};
#undef SM
        for (const void* f : sm) lds_attr(f, LDS_MAX_BYTES - 1024, "k_solve_lds (one-sided) LDS attribute");
    }
    lds_attr(reinterpret_cast<const void*>(&k_sweep<SWEEP_NW, SWEEP_WPF>), FOLD_LDS_BYTES, "k_sweep LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_sweep<SWEEP_NW_BIG, 1>), FOLD_LDS_BYTES, "k_sweep<12> LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_root_gain<SWEEP_NW, 2>), LDS_MAX_BYTES - 1024, "k_root_gain LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_root_gain_m<SWEEP_NW, SWEEP_NW_MID, 2>), LDS_MAX_BYTES - 1024, "k_root_gain_m LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_root_gain_w<SWEEP_NW, 4, 2>), LDS_MAX_BYTES - 1024, "k_root_gain_w LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_root_gain_w<SWEEP_NW, 4, 3>), LDS_MAX_BYTES - 1024, "k_root_gain_w LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_root_gain_w<SWEEP_NW, 6, 2>), LDS_MAX_BYTES - 1024, "k_root_gain_w LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_root_gain_w<SWEEP_NW, 6, 3>), LDS_MAX_BYTES - 1024, "k_root_gain_w LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_feature<64, true, 4>), LDS_MAX_BYTES - 1024, "k_feature<64, true, 4> LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_gain_stream<1>), LDS_MAX_BYTES - 1024, "k_gain_stream<1> LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_gain_stream<2>), LDS_MAX_BYTES - 1024, "k_gain_stream<2> LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_gain_dense), LDS_MAX_BYTES - 1024, "k_gain_dense LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_lsweep<8, 4, LS_RS4, true>), FOLD_LDS_BYTES, "k_lsweep<8,4> LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_lsweep<12, 4, LS_RS4, false>), FOLD_LDS_BYTES, "k_lsweep<12,4> LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_lsweep<SWEEP_NW, 6, LS_RS6>), FOLD_LDS_BYTES, "k_lsweep<6> LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_lsweep<12, 6, LS_RS6, false>), FOLD_LDS_BYTES, "k_lsweep<12,6> LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_lsweep<8, 6, LS_RS6T>), FOLD_LDS_BYTES, "k_lsweep<8,6,tall> LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_wsweep<SWEEP_NW, 4>), FOLD_LDS_BYTES, "k_wsweep<4> LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_wsweep<SWEEP_NW, 6>), FOLD_LDS_BYTES, "k_wsweep<6> LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_chol<512>), LDS_MAX_BYTES - 1024, "k_chol LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_feature<24>), LDS_MAX_BYTES - 1024, "k_feature LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_feature<32>), LDS_MAX_BYTES - 1024, "k_feature LDS attribute");
    lds_attr(reinterpret_cast<const void*>(&k_feature<64>), LDS_MAX_BYTES - 1024, "k_feature LDS attribute");
    // k_propagate keeps T = Phi P[:15, :] (15 x d) + a 15 x 15 block in LDS
    if (((size_t)15 * (15 + 6 * c->maxN) + 225) * 8 > (size_t)(LDS_MAX_BYTES - 1024)) { msckf_destroy(c); return MSCKF_ERR_ARG; }
    lds_attr(reinterpret_cast<const void*>(&k_propagate), LDS_MAX_BYTES - 1024, "k_propagate LDS attribute");
    if (cerr != hipSuccess) {
        std::fprintf(stderr, "msckf_create: %s: %s\n", cwhat, hipGetErrorString(cerr));
        msckf_destroy(c);
        return MSCKF_ERR_HIP;
    }
    const int N = c->maxN, d = 15 + 6 * N, dc = 6 * N;
    int rc = MSCKF_OK;
    auto E = [&](Buf& b, size_t bytes, bool z = false) { if (rc == MSCKF_OK) rc = ensure(c, b, bytes, z); };
    E(c->dP, (size_t)d * d * 8);
    E(c->dPoseArena, (size_t)24 * N * 8);
    c->res_mask_cap = ((size_t)std::max(c->maxF, 1) + 7) & ~(size_t)7;
    c->res_cap = 64 + (size_t)d * 8 + (size_t)d * d * 8 + c->res_mask_cap;
    E(c->dResArena, c->res_cap, true);
    E(c->dGateArena, (size_t)5 * std::max(c->maxF, 1));
    if (rc == MSCKF_OK) {
        set_view(c->dCamR, c->dPoseArena.p, 0, (size_t)N * 72);
        set_view(c->dCamT, c->dPoseArena.p, (size_t)9 * N * 8, (size_t)N * 24);
        set_view(c->dCamR0, c->dPoseArena.p, (size_t)12 * N * 8, (size_t)N * 72);
        set_view(c->dCamT0, c->dPoseArena.p, (size_t)21 * N * 8, (size_t)N * 24);
        c->d = 15;                                                         // no clone yet
        seat_result_views(c);
        bool ok = hipHostMalloc(&c->hPose, (size_t)24 * N * 8) == hipSuccess &&
                  hipHostMalloc(&c->hRes, c->res_cap) == hipSuccess &&
                  hipHostMalloc(&c->hGate, (c->hGateCap = (size_t)5 * std::max(c->maxF, 1))) == hipSuccess &&
                  hipHostMalloc(&c->hP, (size_t)d * d * 8) == hipSuccess;
        if (!ok) rc = MSCKF_ERR_HIP;
    }
    E(c->dChi2, 1024 * 8);
    E(c->dKeep, (size_t)d * 4);                                            // index map of msckf_remove_clones
    E(c->dY, (size_t)d * dc * 8); E(c->dS, (size_t)dc * dc * 8); E(c->dL, (size_t)dc * dc * 8);
    E(c->dU, (size_t)dc * dc * 8); E(c->dInvd, (size_t)dc * 8); E(c->dK, (size_t)d * dc * 8);
    E(c->dB2, (size_t)d * d * 8); E(c->dD, (size_t)d * dc * 8); E(c->dPn, (size_t)d * d * 8);
    E(c->dCholWork, (size_t)dc * (dc + 1) / 2 * 8);
    {   // k_gain_stream: exchange tiles and (epoch-tagged, hence zeroed once) flags, the root sweep's progress word
        const size_t nbm = (size_t)(dc + 15) / 16, nsm = nbm + 1;
        E(c->dGsEx, (nbm + GS_MAX_NB2) * nsm * 256 * 8);      // (two sources of row blocks: the band root and the remainder rows of split long tracks)
        E(c->dRem, 16 * (size_t)GS_MAX_NB2 * (dc + 1) * 8 + 64);   // the dense remainder rows (k_rem_scatter) + their count
        E(c->dGsFlag, ((nbm + GS_MAX_NB2) * nsm + 8) * 8, true);
        E(c->dMProg, 512, true);                       // (progress words of the merge workgroups inside k_root_gain's launch)
        E(c->dGsProg, 512, true);                      // (progress word at 0, k_root_gain's time stamps on a line of their own at byte 256)
    }
    if (rc != MSCKF_OK) { msckf_destroy(c); return rc; }
    {
        const char* e = std::getenv("MSCKF_HOST_THREADS");
        int nw = e ? std::atoi(e) : 3;
        // the CPUs this thread may run on decide (a cpuset / taskset / a launcher that binds ranks to cores), not the machine's
        // count: every worker gets a CPU of its own beside the caller's, or there are fewer workers (none on one CPU)
        int hw = (int)std::thread::hardware_concurrency();
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) hw = std::min(hw > 0 ? hw : CPU_SETSIZE, CPU_COUNT(&allowed));
        nw = std::max(0, std::min(nw, std::max(0, hw - 1)));
        nw = std::min(nw, 15);
        const char* es = std::getenv("MSCKF_HOST_SPIN_US");
        c->pool = new HostPool(nw, es ? std::max(0, std::atoi(es)) : 1000);
    }
    {
        // The runtime sets up a copy engine the first time a copy is handed to it -- 8 ms, measured -- and picks the engine by
        // what is busy: the one-shot call's two uploads (state on stream_up, observations on the main stream) met a second
        // engine the first time they overlapped, on the 4th to 18th call of a process, inside the timed region of a short
        // benchmark.  Overlapping copies in both directions on all three streams now, where the time does not count.
        const size_t dm = 15 + 6 * (size_t)c->maxN;
        const size_t wb = std::min<size_t>(dm * dm * 8, c->res_cap);
        for (int round = 0; round < 6 && wb > 0; ++round) {
            (void)hipMemcpyAsync(c->dP.p, c->hP, wb, hipMemcpyHostToDevice, c->stream_up);
            (void)hipMemcpyAsync(c->dResArena.p, c->hRes, wb, hipMemcpyHostToDevice, c->stream);
            (void)hipMemcpyAsync(c->dPn.p, c->hP, wb, hipMemcpyHostToDevice, c->stream2);
            if (round & 1) {
                (void)hipMemcpyAsync(c->hRes, c->dResArena.p, wb, hipMemcpyDeviceToHost, c->stream);
                (void)hipMemcpyAsync(c->hP, c->dP.p, wb, hipMemcpyDeviceToHost, c->stream_up);
            }
        }
        (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->stream_up); (void)hipStreamSynchronize(c->stream2);
        (void)hipGetLastError();
    }
    *out = c;
    return MSCKF_OK;
}

void msckf_destroy(msckf_ctx* c) {
    if (!c) return;
    delete c->pool;
    c->pool = nullptr;
    if (c->hp_calls > 0 && std::getenv("MSCKF_HOSTPROF")) {
        const double n = (double)c->hp_calls;
        std::fprintf(stderr, "msckf_update host phases, us per call over %ld calls: set_state %.1f | pinned image + validate %.1f, sort %.1f, "
                     "k_gather + K1-K4 launch %.1f, plan %.1f, plan upload %.1f | K5-K7 launches %.1f | wait for K1-K4 + gate sums %.1f | "
                     "wait for device %.1f, unpack %.1f | whole call %.1f\n", c->hp_calls, c->hp[0] / n, c->hp[1] / n, c->hp[2] / n, c->hp[4] / n,
                     c->hp[5] / n, c->hp[6] / n, c->hp[7] / n, c->hp[11] / n, c->hp[8] / n, c->hp[9] / n, c->hp[10] / n);
    }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->stream_up) (void)hipStreamSynchronize(c->stream_up);
    if (c->comm) { (void)rccl().CommDestroy(c->comm); c->comm = nullptr; }
    Buf* all[] = {&c->dP, &c->dPout, &c->dCamR, &c->dCamT, &c->dCamR0, &c->dCamT0, &c->dChi2, &c->dViewPtr,
                  &c->dObsUV, &c->dObsSlot, &c->dBase, &c->dMvec, &c->dRho, &c->dFmin, &c->dBlkOff, &c->dStack,
                  &c->dRank, &c->dAcc, &c->dGamma, &c->dKeep, &c->dNodes, &c->dRbuf, &c->dStamps, &c->dSweepNodes, &c->dSweepFolds, &c->dY, &c->dS, &c->dL,
                  &c->dU, &c->dInvd, &c->dK, &c->dB2, &c->dD, &c->dPn, &c->dDx, &c->dCholWork, &c->dStatus,
                  &c->dLineBase, &c->dLineDir, &c->dLineConf, &c->dLostFor, &c->dTrackedFor, &c->dSelFlags, &c->dWorld,
                  &c->dFlush, &c->dFlushOff, &c->dFeatInfo, &c->dCommBuf, &c->dAssocUV, &c->dAssocRes,
                  &c->dGsEx, &c->dGsFlag, &c->dGsProg, &c->dMProg, &c->dMFlush, &c->dRootFlush, &c->dXRootFlush,
                  &c->dSplit, &c->dRem};
    for (Buf* b : all) if (b->p && !b->view) (void)hipFree(b->p);
    for (Buf* b : {&c->dPoseArena, &c->dFeatArena, &c->dRawArena, &c->dResArena, &c->dGateArena, &c->dPlanArena}) if (b->p) (void)hipFree(b->p);
    for (void* h : {c->hPose, c->hFeat, c->hRes, c->hGate, c->hP}) if (h) (void)hipHostFree(h);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_wfeat) (void)hipEventDestroy(c->ev_wfeat);
    if (c->ev_rem) (void)hipEventDestroy(c->ev_rem);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->ev_plan) (void)hipEventDestroy(c->ev_plan);
    if (c->ev_plan_up) (void)hipEventDestroy(c->ev_plan_up);
    if (c->hPlan) (void)hipHostFree(c->hPlan);
    if (c->ev_state) (void)hipEventDestroy(c->ev_state);
    if (c->ev_gate) (void)hipEventDestroy(c->ev_gate);
    if (c->stream_up) (void)hipStreamDestroy(c->stream_up);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int msckf_set_state(msckf_ctx* c, int32_t N, const double* P, const double* cam_R, const double* cam_t,
                    const double* cam_R0, const double* cam_t0, const double* gravity, const double* Kinv,
                    double sigma, const double* chi2_crit, int32_t n_crit) {
    if (!c || !P || !gravity || !Kinv || !chi2_crit) return MSCKF_ERR_ARG;
    if (N < 0 || N > c->maxN || n_crit < 2 || n_crit > 1024) return MSCKF_ERR_ARG;     // N = 0: no clone yet (f2)
    if (N > 0 && (!cam_R || !cam_t || !cam_R0 || !cam_t0)) return MSCKF_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const double t0 = now_us();
    if (N != c->N) c->have_features = false;
    c->N = N; c->d = 15 + 6 * N; c->dc = 6 * N;
    seat_result_views(c);
    c->sigma = sigma; c->n_chi2 = n_crit;
    std::memcpy(c->g, gravity, 24);
    std::memcpy(c->Kinv, Kinv, 72);
    const size_t d = c->d;
    // (one-shot call: the state goes up on stream_up, beside the tracks on the main stream)
    hipStream_t st = c->defer_state_sync ? c->stream_up : c->stream;
    if (c->pool) c->pool->copy(c->hP, P, d * d * 8); else std::memcpy(c->hP, P, d * d * 8);
    HIPCHK(c, hipMemcpyAsync(c->dP.p, c->hP, d * d * 8, hipMemcpyHostToDevice, st));
    if (N > 0) {
        c->h_cam[0].assign(cam_R, cam_R + (size_t)N * 9); c->h_cam[1].assign(cam_t, cam_t + (size_t)N * 3);
        c->h_cam[2].assign(cam_R0, cam_R0 + (size_t)N * 9); c->h_cam[3].assign(cam_t0, cam_t0 + (size_t)N * 3);
        if (int rcp = upload_poses(c, st)) return rcp;
    } else {
        for (auto& v : c->h_cam) v.clear();
    }
    if ((int)c->chi2_cache.size() != n_crit || std::memcmp(c->chi2_cache.data(), chi2_crit, (size_t)n_crit * 8) != 0) {
        c->chi2_cache.assign(chi2_crit, chi2_crit + n_crit);           // the table rarely changes between calls
        HIPCHK(c, hipMemcpyAsync(c->dChi2.p, c->chi2_cache.data(), (size_t)n_crit * 8, hipMemcpyHostToDevice, st));
    }
    if (!c->defer_state_sync) HIPCHK(c, hipStreamSynchronize(c->stream));
    else { HIPCHK(c, hipEventRecord(c->ev_state, c->stream_up)); c->state_pending = true; }
    c->us_h2d = (float)(now_us() - t0);
    c->have_state = true;
    c->ran = false;
    return MSCKF_OK;
}

int msckf_set_features(msckf_ctx* c, int32_t F, const int32_t* view_ptr, const double* obs_uv,
                       const int32_t* obs_slot, const double* idp_base, const double* idp_m,
                       const double* idp_rho) {
    if (!c || F < 0 || F > c->maxF) return MSCKF_ERR_ARG;
    if (!c->have_state) return MSCKF_ERR_STATE;
    if (F > 0 && (!view_ptr || !obs_uv || !obs_slot || !idp_base || !idp_m || !idp_rho)) return MSCKF_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const double t0 = now_us();
    const int N = c->N;
    if (c->feat_busy) { HIPCHK(c, hipStreamSynchronize(c->stream)); c->feat_busy = false; }     // (the pinned image is rewritten below)
    // the previous batch is gone from here on: a validation failure below must not leave `have_features`
    // standing over arenas / plan that describe another F (run() then returns MSCKF_ERR_STATE)
    c->have_features = false;
    c->feature_launched = false;
    c->ran = false;
    c->have_tracks = false;
    c->use_select = false;
    c->no_wide = false;
    if (F == 0) {
        c->F = 0;
        c->sumM = 0; c->Mmax = 0; c->nodes.clear(); c->levels.clear(); c->snodes.clear(); c->sfolds.clear(); c->sweep_levels.clear(); c->n_group_merges = 0;
        c->band_plan = false; c->root = -1; c->perm.clear();
        c->Fb = c->Fw = c->Fw1 = 0; c->wide_active = false;
        c->Fs = 0; c->nNarrow = 0; c->sumMs = 0; c->split_on = false; c->rem_cap = 0; c->rem_direct = false; c->h_split.clear(); c->h_parent.clear();
        c->rnodes.clear(); c->rlevels.clear(); c->rroot = -1; c->rtops.clear(); c->rtop_rows = 0;
        c->plan_valid = false;
        c->xchg_planned = false;
        if (c->xchg) {
            // a shard without tracks still takes part in the gather: an empty record (no flag, count 0, no gate byte)
            // heads the workspace; the triangles behind it are never read (the merging rank goes by the flags)
            const size_t need = (rec_head(c) + (size_t)N * rec_slot(c) + 16) * 8;
            if (c->dRbuf.bytes < need) {
                if (c->dRbuf.p) HIPCHK(c, hipFree(c->dRbuf.p));
                c->dRbuf.p = nullptr; c->dRbuf.bytes = 0;
                HIPCHK(c, hipMalloc(&c->dRbuf.p, need));
                c->dRbuf.bytes = need;
            }
            HIPCHK(c, hipMemsetAsync(c->dRbuf.p, 0, rec_head(c) * 8, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            c->xchg_planned = true;
            c->rbuf_doubles = rec_head(c) + (size_t)N * rec_slot(c);
            c->gather_off = c->rbuf_doubles;
        }
        c->have_features = true;
        c->us_host_prep = (float)(now_us() - t0);
        return MSCKF_OK;
    }
    // The caller's arrays stay as they are (k_gather.h brings them into the pipeline's order on the device): the observations
    // (60 % of the bytes) into the pinned image and up by DMA at once -- they cross PCIe while the host validates and sorts --,
    // the other arrays into the pinned image, from where k_gather reads them.
    if (view_ptr[0] != 0) return MSCKF_ERR_ARG;
    // the CSR offsets first, all of them (a few microseconds): the feature ranges below read obs_slot[view_ptr[f] ..] and must
    // not do so through an offset no earlier range has vouched for (a malformed view_ptr would send them out of bounds)
    for (int f = 0; f < F; ++f) {
        const int M = view_ptr[f + 1] - view_ptr[f];
        if (M < 1 || M > c->maxM) return MSCKF_ERR_ARG;
    }
    const int sumM = view_ptr[F];
    // raw image (input order): doubles first, then the ints
    const size_t r_uv = 0, r_base = r_uv + (size_t)sumM * 16, r_m = r_base + (size_t)F * 24, r_rho = r_m + (size_t)F * 24;
    const size_t r_slot = r_rho + (size_t)F * 8;
    const size_t raw_bytes = (r_slot + (size_t)sumM * 4 + 15) & ~(size_t)15;
    // the sort's records behind the arrays: one per entry of the sorted arrays -- the tracks and, for long tracks that are split,
    // their blocks: a narrow block per view group of 2+ views and one remainder block, i.e. at most M / 2 + 1 per track (a
    // ragged track may span 11+ slots with three views) -- and one SplitRec per long track
    const size_t rec_cap = 2 * (size_t)F + (size_t)sumM / 2 + 8;
    const size_t pin_bytes = raw_bytes + rec_cap * sizeof(GatherRec) + ((size_t)F + 1) * sizeof(SplitRec);
    if (c->hFeatCap < pin_bytes) {
        if (c->hFeat) HIPCHK(c, hipHostFree(c->hFeat));
        c->hFeat = nullptr; c->hFeatCap = 0;
        HIPCHK(c, hipHostMalloc(&c->hFeat, pin_bytes + pin_bytes / 2));
        c->hFeatCap = pin_bytes + pin_bytes / 2;
    }
    // small batches: the small arrays reach HBM through a copy KERNEL reading the pinned image (k_stage) and k_gather reads the
    // sort's records from it (zero-copy) -- no copy command but the observations', none of the ~9 us each one waits behind its
    // predecessor; large ones: DMA, beside the host's sort (measured at 10000 tracks: zero-copy 620 us per call, DMA 578)
    static const int zc_max = [] { const char* e = std::getenv("MSCKF_ZEROCOPY_MAX"); return e ? std::atoi(e) : 4096; }();
    const bool zc = F <= zc_max;
    if (int rca = ensure(c, c->dRawArena, pin_bytes)) return rca;
    char* hb = static_cast<char*>(c->hFeat);
    char* draw = static_cast<char*>(c->dRawArena.p);
    const bool par = c->pool && F >= host_par_min();
    if (par) c->pool->copy(hb + r_uv, obs_uv, (size_t)sumM * 16); else std::memcpy(hb + r_uv, obs_uv, (size_t)sumM * 16);
    HIPCHK(c, hipMemcpyAsync(draw, hb + r_uv, (size_t)sumM * 16, hipMemcpyHostToDevice, c->stream));
    // (the one-shot call's state went up on the side stream: the main stream learns of it HERE, behind a copy it has to wait for
    //  anyway -- between k_gather and k_feature the cross-stream wait cost ~4 us of an otherwise back-to-back pair)
    if (c->state_pending) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_state, 0)); c->state_pending = false; }
    // validate + first/last slot of each track + sort key (feature ranges on the host pool; the lowest failing range decides
    // the code), and the range's share of the remaining arrays into the pinned image
    // key = (class, first slot, last slot); class 0: tracks of up to WIDE_SPAN clone slots (and long ones that cannot be split:
    // views out of slot order, more than SPLIT_MAXG groups, more views than k_feature<64> holds); class 2: long tracks, split
    const bool split = split_ok(c, N);
    const size_t NN = (size_t)N * N;
    std::vector<int> key_in(F);
    std::vector<unsigned char> M_in(F);
    int Mmax = 0, Mmax_cls[3] = {0, 0, 0};
    int n_mid = 0, n_long = 0;               // tracks of 11 - 15 slots / of more (any class)
    {
        const int nch = par ? std::min(4 * (c->pool->workers() + 1), (F + 255) / 256) : 1;
        std::vector<int> ch_err(nch, MSCKF_OK), ch_mmax(5 * nch, 0);
        const int maxM = c->maxM;
        auto validate = [&](int ch) {
            const int f0 = (int)((long long)F * ch / nch), f1 = (int)((long long)F * (ch + 1) / nch);
            const int a0 = view_ptr[f0], a1 = view_ptr[f1];
            std::memcpy(hb + r_slot + (size_t)a0 * 4, obs_slot + a0, (size_t)(a1 - a0) * 4);
            std::memcpy(hb + r_base + (size_t)f0 * 24, idp_base + (size_t)f0 * 3, (size_t)(f1 - f0) * 24);
            std::memcpy(hb + r_m + (size_t)f0 * 24, idp_m + (size_t)f0 * 3, (size_t)(f1 - f0) * 24);
            std::memcpy(hb + r_rho + (size_t)f0 * 8, idp_rho + f0, (size_t)(f1 - f0) * 8);
            int mm[3] = {0, 0, 0}, nmid = 0, nlong = 0;
            for (int f = f0; f < f1; ++f) {
                const int a = view_ptr[f], b = view_ptr[f + 1], M = b - a;
                if (M < 1 || M > maxM) { ch_err[ch] = MSCKF_ERR_ARG; return; }
                int lo = N, hi = -1;
                bool ordered = true;
                unsigned long long seen = 0;    // N <= 64 fast path; general check below
                for (int i = a; i < b; ++i) {
                    const int sl = obs_slot[i];
                    if (sl < 0 || sl >= N) { ch_err[ch] = MSCKF_ERR_ARG; return; }
                    if (sl < hi) ordered = false;
                    lo = std::min(lo, sl); hi = std::max(hi, sl);
                    if (N <= 64) {
                        if (seen & (1ull << sl)) { ch_err[ch] = MSCKF_ERR_DUP_SLOT; return; }
                        seen |= 1ull << sl;
                    } else {
                        for (int k = a; k < i; ++k) if (obs_slot[k] == sl) { ch_err[ch] = MSCKF_ERR_DUP_SLOT; return; }
                    }
                }
                const int span = hi - lo + 1;
                if (span > 15) ++nlong; else if (span > WIDE_SPAN) ++nmid;
                const int cls = (split && span > WIDE_SPAN && ordered && M <= 31 && (span + SPLIT_GSLOTS - 1) / SPLIT_GSLOTS <= SPLIT_MAXG) ? 2 : 0;
                key_in[f] = (int)(cls * NN + (size_t)lo * N + hi);
                M_in[f] = (unsigned char)M;
                mm[cls] = std::max(mm[cls], M);
            }
            for (int k = 0; k < 3; ++k) ch_mmax[5 * ch + k] = mm[k];
            ch_mmax[5 * ch + 3] = nmid; ch_mmax[5 * ch + 4] = nlong;
        };
        if (nch > 1) c->pool->run(nch, validate); else validate(0);
        for (int ch = 0; ch < nch; ++ch) {
            if (ch_err[ch] != MSCKF_OK) { (void)hipStreamSynchronize(c->stream); return ch_err[ch]; }      // (the pinned image is in flight)
            for (int k = 0; k < 3; ++k) Mmax_cls[k] = std::max(Mmax_cls[k], ch_mmax[5 * ch + k]);
            n_mid += ch_mmax[5 * ch + 3]; n_long += ch_mmax[5 * ch + 4];
        }
        Mmax = std::max(Mmax_cls[0], std::max(Mmax_cls[1], Mmax_cls[2]));
    }
    // A batch MOST of whose tracks span 11 - 15 slots and none more (BASELINE configs[4]: every track 15 views) keeps the
    // 90-column band pipeline for all of them: split, each would leave 3 remainder rows to the dense tree
    // ... and so does one whose remainder rows (6 per such track) would be too many for K6-K7 to take as they are: the 90-column
    // pipeline (539 us at 2000 tracks ~ U[2, 15]) beats band pipeline + remainder tree (866 us) there
    if (Mmax_cls[2] > 0 && n_long == 0 && (2 * n_mid > F || 6 * n_mid > (c->dc > FOLD_RLDS_MAX_W ? c->rem_direct_max_wide : std::min(c->rem_direct_max, 2048)))) {
        for (int f = 0; f < F; ++f) key_in[f] = (int)((size_t)key_in[f] % NN);
        Mmax_cls[0] = Mmax; Mmax_cls[2] = 0;
    }
    if (!zc) HIPCHK(c, hipMemcpyAsync(draw + r_base, hb + r_base, raw_bytes - r_base, hipMemcpyHostToDevice, c->stream));
    else {
        const size_t n16 = (raw_bytes - r_base) / 16;           // (r_base and raw_bytes are multiples of 16)
        hipLaunchKernelGGL(k_stage, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 512)), dim3(256), 0, c->stream,
                           reinterpret_cast<const uint4*>(hb + r_base), reinterpret_cast<uint4*>(draw + r_base), n16);
    }
    const double tv = now_us();
    if (c->n_chi2 <= 2 * Mmax) { (void)hipStreamSynchronize(c->stream); return MSCKF_ERR_ARG; }
    c->F = F; c->sumM = sumM; c->Mmax = Mmax;            // validated: commit the batch size
    // counting sort by the key: stable, O(F + N^2); the short tracks first, the long ones behind them
    c->perm.resize(F);
    std::vector<int> h_view, h_fmin(F), h_fmax(F);
    long long blk = 0;
    int Fs = F, sumMs = sumM;
    GatherRec* rec = reinterpret_cast<GatherRec*>(hb + raw_bytes);
    {
        std::vector<int> cnt(3 * NN + 1, 0);
        for (int f = 0; f < F; ++f) cnt[key_in[f] + 1]++;
        // first / last slot of the sorted tracks: a run per occupied key
        for (size_t k = 0, pos = 0; k < 3 * NN; ++k) {
            const int n = cnt[k + 1];
            if (n == 0) continue;
            const int lo = (int)((k % NN) / N), hi = (int)(k % N);
            std::fill(h_fmin.begin() + pos, h_fmin.begin() + pos + n, lo);
            std::fill(h_fmax.begin() + pos, h_fmax.begin() + pos + n, hi);
            pos += n;
        }
        for (size_t i = 1; i < cnt.size(); ++i) cnt[i] += cnt[i - 1];
        c->Fb = cnt[NN];                                                  // short tracks: sorted positions [0, Fb)
        c->Fw1 = 0;
        c->Fw = F - c->Fb;                                                // long tracks, split: [Fb, F)
        int* perm = c->perm.data();
        for (int f = 0; f < F; ++f) perm[cnt[key_in[f]]++] = f;
        c->Mmax_band = Mmax_cls[0]; c->Mmax_w1 = 0; c->Mmax_wide = Mmax_cls[2];
        c->split_on = c->Fw > 0;
        // ---- the blocks of the long tracks (k_feature.h): view groups of at most SPLIT_GSLOTS slots -> narrow blocks, one
        //      remainder block per track
        c->h_split.clear(); c->h_parent.clear();
        struct Blk { int parent, f, a, M, lo, hi, pi, g; };
        std::vector<Blk> narrow, wide;
        int rem_cap = 0;
        if (c->split_on) {
            c->h_split.resize(c->Fw);
            for (int sidx = c->Fb; sidx < F; ++sidx) {
                const int f = perm[sidx], a = view_ptr[f], M = M_in[f], lo = h_fmin[sidx], hi = h_fmax[sidx];
                const int span = hi - lo + 1, ng0 = (span + SPLIT_GSLOTS - 1) / SPLIT_GSLOTS, pi = sidx - c->Fb;
                SplitRec sr{};
                for (int g = 0; g < SPLIT_MAXG; ++g) sr.child[g] = -1;
                int ng = 0, v = 0;
                for (int g0 = 0; g0 < ng0; ++g0) {
                    const int bnd = lo + (int)(((long long)(g0 + 1) * span) / ng0);       // slots [.., bnd): at most ceil(span / ng0) <= 10 of them
                    const int v0 = v;
                    while (v < M && obs_slot[a + v] < bnd) ++v;
                    if (v == v0) continue;                                 // no view in this stretch of slots
                    sr.gv[ng] = (unsigned char)v0;
                    if (v - v0 >= 2) narrow.push_back({sidx, f, a + v0, v - v0, obs_slot[a + v0], obs_slot[a + v - 1], pi, ng});
                    ++ng;
                }
                sr.gv[ng] = (unsigned char)M; sr.ng = (unsigned char)ng;
                sr.rows_cap = 3 * ng; rem_cap += 3 * ng;
                c->h_split[pi] = sr;
                wide.push_back({sidx, f, a, M, lo, hi, pi, -1});
            }
            // the narrow blocks among themselves by (first slot, last slot): the band plan walks them as a second run
            std::vector<int> cn(NN + 1, 0);
            for (const Blk& b : narrow) cn[(size_t)b.lo * N + b.hi + 1]++;
            for (size_t i = 1; i < cn.size(); ++i) cn[i] += cn[i - 1];
            std::vector<Blk> sorted(narrow.size());
            for (const Blk& b : narrow) sorted[cn[(size_t)b.lo * N + b.hi]++] = b;
            narrow.swap(sorted);
        }
        c->nNarrow = (int)narrow.size();
        if ((size_t)F + narrow.size() + wide.size() > rec_cap) { (void)hipStreamSynchronize(c->stream); c->last_error = "split: record capacity"; return MSCKF_ERR_STATE; }
        c->rem_cap = rem_cap;
        // (K6-K7 takes 16 dense rows in ~5 us; a merge tree over them is a leaf of ~100 us and ~170 us per level: the tree pays
        //  beyond some 2000 rows)
        // ... and the tree's kernels are at their slowest on windows wider than their LDS holds (N > 31: ~1 ms per level), where the
        // rows are taken as they are up to four times as many
        c->rem_direct = c->split_on && rem_cap <= (c->dc > FOLD_RLDS_MAX_W ? c->rem_direct_max_wide : c->rem_direct_max);
        Fs = F + (int)narrow.size() + (int)wide.size();
        h_view.resize(Fs + 1); h_fmin.resize(Fs); h_fmax.resize(Fs);
        c->h_parent.resize(Fs - F);
        // the sorted CSR offsets and the offsets of the K4 blocks: a prefix over the sorted order; k_gather's records
        int pos = 0;
        for (int sidx = 0; sidx < F; ++sidx) {
            const int f = perm[sidx], M = M_in[f];
            h_view[sidx] = pos;
            const bool parent = c->split_on && sidx >= c->Fb;              // (a split track has no block of its own)
            rec[sidx] = GatherRec{f, view_ptr[f], M, pos, parent ? 0 : blk};
            if (!parent) blk += (long long)(6 * M + 1) * (2 * M);
            pos += M;
        }
        int e = F;
        for (const Blk& b : narrow) {
            h_view[e] = pos; h_fmin[e] = b.lo; h_fmax[e] = b.hi; c->h_parent[e - F] = b.parent;
            rec[e] = GatherRec{b.f, b.a, b.M, pos, blk};
            c->h_split[b.pi].child[b.g] = e;
            blk += (long long)(6 * b.M + 1) * (2 * b.M);
            pos += b.M; ++e;
        }
        for (const Blk& b : wide) {
            h_view[e] = pos; h_fmin[e] = b.lo; h_fmax[e] = b.hi; c->h_parent[e - F] = b.parent;
            rec[e] = GatherRec{b.f, b.a, b.M, pos, blk};
            c->h_split[b.pi].wide = e;
            blk += (long long)(6 * b.M + 1) * (3 * SPLIT_MAXG);
            pos += b.M; ++e;
        }
        h_view[Fs] = pos;
        sumMs = pos;
    }
    c->Fs = Fs; c->sumMs = sumMs;
    const size_t split_off = raw_bytes + (((size_t)Fs * sizeof(GatherRec) + 15) & ~(size_t)15);
    if (c->split_on) std::memcpy(hb + split_off, c->h_split.data(), c->h_split.size() * sizeof(SplitRec));
    c->h_view_sorted = h_view;
    c->h_view_in.assign(view_ptr, view_ptr + F + 1);
    c->h_fmin = h_fmin; c->h_fmax = h_fmax;
    const double ts = now_us(), t1 = ts;

    // sorted image (what the kernels read), written by k_gather: Fs entries, sumMs views
    const size_t o_uv = 0, o_base = o_uv + (size_t)sumMs * 16, o_m = o_base + (size_t)Fs * 24, o_rho = o_m + (size_t)Fs * 24;
    const size_t o_blk = o_rho + (size_t)Fs * 8, o_view = o_blk + (size_t)Fs * 8;
    const size_t o_perm = o_view + (((size_t)(Fs + 1) * 4 + 7) & ~(size_t)7);                  // sorted position -> input index
    const size_t o_slot = (o_perm + (size_t)Fs * 4 + 7) & ~(size_t)7, o_fmin = o_slot + (((size_t)sumMs * 4 + 7) & ~(size_t)7);
    const size_t o_info = (o_fmin + (size_t)Fs * 4 + 15) & ~(size_t)15;                   // FeatInfo records (16-byte aligned)
    const size_t feat_bytes = o_info + (size_t)Fs * sizeof(FeatInfo);
    if (c->dFeatArena.bytes < feat_bytes && c->run_pending) { HIPCHK(c, hipStreamSynchronize(c->stream)); c->run_pending = false; }
    if (int rca = ensure(c, c->dFeatArena, feat_bytes)) return rca;
    set_view(c->dObsUV, c->dFeatArena.p, o_uv, (size_t)sumMs * 16);
    set_view(c->dBase, c->dFeatArena.p, o_base, (size_t)Fs * 24);
    set_view(c->dMvec, c->dFeatArena.p, o_m, (size_t)Fs * 24);
    set_view(c->dRho, c->dFeatArena.p, o_rho, (size_t)Fs * 8);
    set_view(c->dBlkOff, c->dFeatArena.p, o_blk, (size_t)Fs * 8);
    set_view(c->dViewPtr, c->dFeatArena.p, o_view, (size_t)(Fs + 1) * 4);
    set_view(c->dObsSlot, c->dFeatArena.p, o_slot, (size_t)sumMs * 4);
    set_view(c->dFmin, c->dFeatArena.p, o_fmin, (size_t)Fs * 4);
    set_view(c->dFeatInfo, c->dFeatArena.p, o_info, (size_t)Fs * sizeof(FeatInfo));
    set_view(c->dPerm, c->dFeatArena.p, o_perm, (size_t)Fs * 4);
    // gate results: rank[Fs] (int) then accepted[Fs] (byte), contiguous so they come back in one copy
    if ((size_t)5 * Fs > c->dGateArena.bytes || (size_t)5 * Fs > c->hGateCap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (int rcg = ensure(c, c->dGateArena, (size_t)8 * Fs)) return rcg;
        if (c->hGate) HIPCHK(c, hipHostFree(c->hGate));
        c->hGate = nullptr; c->hGateCap = 0;
        HIPCHK(c, hipHostMalloc(&c->hGate, (size_t)8 * Fs));
        c->hGateCap = (size_t)8 * Fs;
    }
    set_view(c->dRank, c->dGateArena.p, 0, (size_t)Fs * 4);
    set_view(c->dAcc, c->dGateArena.p, (size_t)Fs * 4, (size_t)Fs);
    int rc = MSCKF_OK;
    auto E = [&](Buf& b, size_t bytes, bool z = false) { if (rc == MSCKF_OK) rc = ensure(c, b, bytes, z); };
    const size_t stack_es = c->cfg.dtype == MSCKF_DTYPE_F32 ? 4 : 8;
    E(c->dStack, ((size_t)blk + 8) * stack_es); E(c->dGamma, (size_t)F * 8);
    if (c->split_on) E(c->dSplit, c->h_split.size() * sizeof(SplitRec));
    c->stack_elems = blk;
    if (rc != MSCKF_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
    // the permutation on the device; in the one-shot call K1-K4 starts behind it while the host plans K5
    if (!zc) HIPCHK(c, hipMemcpyAsync(draw + raw_bytes, hb + raw_bytes, (size_t)Fs * sizeof(GatherRec), hipMemcpyHostToDevice, c->stream));
    if (c->split_on) HIPCHK(c, hipMemcpyAsync(c->dSplit.p, hb + split_off, c->h_split.size() * sizeof(SplitRec), hipMemcpyHostToDevice, c->stream));
    if (c->oneshot) HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
    {
        GatherArgs g;
        g.uv_in = reinterpret_cast<const double*>(draw);
        g.slot_in = reinterpret_cast<const int*>(draw + r_slot); g.base_in = reinterpret_cast<const double*>(draw + r_base);
        g.m_in = reinterpret_cast<const double*>(draw + r_m); g.rho_in = reinterpret_cast<const double*>(draw + r_rho);
        g.rec = reinterpret_cast<const GatherRec*>((zc ? hb : draw) + raw_bytes);
        g.uv = ptr<double>(c->dObsUV); g.base = ptr<double>(c->dBase); g.m = ptr<double>(c->dMvec); g.rho = ptr<double>(c->dRho);
        g.slot = ptr<int>(c->dObsSlot); g.fmin = ptr<int>(c->dFmin); g.view = ptr<int>(c->dViewPtr); g.perm = ptr<int>(c->dPerm);
        g.blk = ptr<long long>(c->dBlkOff); g.info = ptr<FeatInfo>(c->dFeatInfo);
        g.F = Fs; g.sumM = sumMs;
        hipLaunchKernelGGL(k_gather, dim3((Fs + GATHER_THREADS / 32 - 1) / (GATHER_THREADS / 32)), dim3(GATHER_THREADS), 0, c->stream, g);
        HIPCHK(c, hipGetLastError());
    }
    // (k_lsweep's zero words behind the stack are written by k_feature itself: one launch less in front of it)
    c->feature_launched = false;
    if (c->state_pending) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_state, 0)); c->state_pending = false; }
    c->gate_event = false;
    if (c->oneshot) {
        if (int rcf = launch_feature(c)) return rcf;
        c->feature_launched = true;
        if (c->gate_direct) { HIPCHK(c, hipEventRecord(c->ev_gate, c->stream)); c->gate_event = true; }
    }
    const double t2 = now_us();
    const bool plan_hit = c->plan_valid && !c->plan_no_wide && c->plan_N == N && c->plan_xchg == c->xchg && c->plan_view == h_view &&
                          c->plan_fmin == h_fmin && c->plan_fmax == h_fmax;
    if (!plan_hit) {
        c->plan_valid = false;
        plan_batch(c, h_fmin, h_fmax, h_view);
        c->plan_N = N; c->plan_xchg = c->xchg; c->plan_view = h_view;   // (h_fmin / h_fmax are stored below, after their last use)
    }
    // room for gathered shard blocks behind the plan's blocks
    c->gather_off = c->rbuf_doubles;
    const double t3 = now_us();
    c->us_host_prep = (float)((t1 - t0) + (t3 - t2));
    if (!plan_hit) {
        // the R workspace is zero-initialised once: entries below a block's diagonal are never written
        const size_t need = (c->rbuf_doubles + 16) * 8;
        if (c->dRbuf.bytes < need) {
            if (c->dRbuf.p) HIPCHK(c, hipFree(c->dRbuf.p));
            c->dRbuf.p = nullptr; c->dRbuf.bytes = 0;
            const size_t want = need + need / 2;
            HIPCHK(c, hipMalloc(&c->dRbuf.p, want));
            c->dRbuf.bytes = want;
        }
        // (on the side stream unless a pipeline nobody waited for may still be reading the workspace or the tables)
        const bool side = c->oneshot || !c->run_pending;
        c->plan_stream = side ? c->stream_up : c->stream;
        HIPCHK(c, hipMemsetAsync(c->dRbuf.p, 0, need, c->plan_stream));
        c->plan_valid = false;
        const int rcp = upload_plan(c);
        if (side) {
            HIPCHK(c, hipEventRecord(c->ev_plan, c->stream_up));
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_plan, 0));
        }
        c->plan_stream = nullptr;
        if (rcp) return rcp;
        c->plan_fmin = h_fmin; c->plan_fmax = h_fmax;
        c->plan_valid = true;
        c->plan_no_wide = false;
    }
    c->feat_busy = true; c->main_busy = true;       // (resident call: the uploads and k_gather stay in the stream)
    c->us_h2d += (float)((t2 - t1) + (now_us() - t3));
    c->hp[1] += tv - t0; c->hp[2] += ts - tv; c->hp[3] += t1 - ts; c->hp[4] += t2 - t1; c->hp[5] += t3 - t2; c->hp[6] += now_us() - t3;
    c->have_features = true;
    return MSCKF_OK;
}

int msckf_run(msckf_ctx* c) {
    if (!c) return MSCKF_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    c->want_direct = c->direct_enabled;             // results mirrored into the pinned host buffers, as in the one-shot call
    const int rc = run_pipeline(c, true, nullptr);
    c->want_direct = false;
    c->main_busy = true;
    return rc;
}

// The current batch planned afresh.  one_plan: every block of the split long tracks with the short tracks in ONE merge tree
// (blocks that leave this context: msckf_run_compress); else the same kind of plan under changed switches (the retry of a
// timed-out launch).
static int replan_current(msckf_ctx* c, bool one_plan) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream2));
    if (one_plan) c->no_wide = true;
    c->plan_valid = false;
    plan_batch(c, c->h_fmin, c->h_fmax, c->h_view_sorted, nullptr);
    c->plan_no_wide = true;                // (not a plan the cache may hand to the next batch)
    c->gather_off = c->rbuf_doubles;
    const size_t need = (c->rbuf_doubles + 16) * 8;
    if (c->dRbuf.bytes < need) {
        if (c->dRbuf.p) HIPCHK(c, hipFree(c->dRbuf.p));
        c->dRbuf.p = nullptr; c->dRbuf.bytes = 0;
        HIPCHK(c, hipMalloc(&c->dRbuf.p, need + need / 2));
        c->dRbuf.bytes = need + need / 2;
    }
    HIPCHK(c, hipMemsetAsync(c->dRbuf.p, 0, need, c->stream));
    return upload_plan(c);
}

int msckf_run_compress(msckf_ctx* c) {
    if (!c) return MSCKF_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->have_features && c->wide_active) {
        // the compressed block leaves this context (msckf_export_block): one plan for every track, the wide ones included
        if (int rcp = replan_current(c, true)) return rcp;
    }
    return run_pipeline(c, false, nullptr);
}

int msckf_sync(msckf_ctx* c) {
    if (!c) return MSCKF_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream_up));
    HIPCHK(c, hipStreamSynchronize(c->stream2));        // (the long tracks' kernels; the main stream is behind them by events already)
    c->feat_busy = c->pose_busy = c->main_busy = c->run_pending = false;
    return MSCKF_OK;
}

int msckf_run_timed(msckf_ctx* c, int32_t iters, float* ms_total, float* stage_us) {
    if (!c || iters < 1) return MSCKF_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
    for (int i = 0; i < iters; ++i)
        if ((rc = run_pipeline(c, true, nullptr)) != MSCKF_OK) return rc;
    HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev[5]));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[4], c->ev[5]));
    if (ms_total) *ms_total = ms;
    c->us_total = ms * 1000.0f / iters;
    if (stage_us) {
        double acc[3] = {0, 0, 0};
        const int reps = std::min(iters, 20);
        for (int i = 0; i < reps; ++i) {
            c->gs_stamp = true;
            rc = run_pipeline(c, true, c->ev);
            c->gs_stamp = false;
            if (rc != MSCKF_OK) return rc;
            HIPCHK(c, hipEventSynchronize(c->ev[3]));
            for (int s = 0; s < 3; ++s) {
                float t = 0;
                HIPCHK(c, hipEventElapsedTime(&t, c->ev[s], c->ev[s + 1]));
                acc[s] += t * 1000.0;
            }
            if (c->gs_fused_last) {
                // one launch holds the root sweep and K6-K7: what trails the sweep's last published row is K6-K7's share
                long long ts[3] = {0, 0, 0};
                HIPCHK(c, hipMemcpy(ts, ptr<long long>(c->dGsProg) + 32, 24, hipMemcpyDeviceToHost));
                if (std::getenv("MSCKF_GS_DEBUG")) {
                    long long t8[32];
                    HIPCHK(c, hipMemcpy(t8, ptr<long long>(c->dGsProg) + 32, 256, hipMemcpyDeviceToHost));
                    std::fprintf(stderr, "k_root_gain (us after the sweep started): last row published %.1f, update done %.1f | strip 0 saw the row blocks at",
                                 (t8[1] - t8[0]) * 0.01, (t8[2] - t8[0]) * 0.01);
                    for (int b = 0; b < (c->dc + 15) / 16 && b < 15; ++b) std::fprintf(stderr, " %.1f", (t8[3 + b] - t8[0]) * 0.01);
                    std::fprintf(stderr, " | the flusher published them at");
                    for (int b = 1; b < (c->dc + 15) / 16 && b < 14; ++b) std::fprintf(stderr, " %.1f", (t8[18 + b] - t8[0]) * 0.01);
                    std::fprintf(stderr, "\n");
                }
                const double tail = (double)(ts[2] - ts[1]) * 0.01;
                if (tail > 0.0 && tail < 1000.0) { acc[1] -= tail; acc[2] += tail; }
            }
        }
        for (int s = 0; s < 3; ++s) { stage_us[s] = (float)(acc[s] / reps); c->us_stage[s] = stage_us[s]; }
    }
    return MSCKF_OK;
}

int msckf_get_result(msckf_ctx* c, double* dx, double* P_out, uint8_t* accepted, msckf_stats* st) {
    if (!c) return MSCKF_ERR_ARG;
    if (!c->ran) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    const double t0 = now_us();
    const size_t d = c->d;
    // gate results and (when the gain stage ran) status | dx | P_out: two copies behind the pipeline, one sync
    if (c->F > 0 && !c->gate_direct) HIPCHK(c, hipMemcpyAsync(c->hGate, c->dGateArena.p, (size_t)c->Fs * 5, hipMemcpyDeviceToHost, c->stream));
    const bool direct = c->res_direct && c->direct_serial == c->run_serial;     // (a merge behind the update wrote the HBM arena only)
    if (c->ran_gain && !direct) {
        const size_t bytes = P_out ? c->res_p_off + d * d * 8 : c->res_dx_off + d * 8;
        HIPCHK(c, hipMemcpyAsync(c->hRes, c->dResArena.p, bytes, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->feat_busy = c->pose_busy = c->main_busy = c->run_pending = false;
    const double tsync = now_us();
    int counters[4] = {0, 0, 0, 0};
    int status[4] = {0};
    std::vector<unsigned char> acc_sorted;
    const bool gate_done = c->gate_serial == c->run_serial && (!accepted || accepted == c->gate_mask_dst);
    if (gate_done) std::memcpy(counters, c->gate_cnt, sizeof(counters));
    else if (int rc0 = gate_counts(c, counters, &acc_sorted, true)) return rc0;
    if (c->ran_gain) std::memcpy(status, c->hRes, 16);
    const int n_acc = c->acc_from_dev ? status[2] : (c->acc_override >= 0) ? c->acc_override : counters[0];
    if (!(c->ran_gain && n_acc > 0)) status[0] = status[1] = 0;
    int rc = (n_acc == 0) ? MSCKF_NOOP : MSCKF_OK;
    if (rc == MSCKF_OK && c->ran_gain && (status[0] != 0 || ((c->gain_blocked || c->wide_active) && status[1] != 0))) rc = MSCKF_ERR_NOT_SPD;
    if (rc == MSCKF_ERR_NOT_SPD && status[0] == 3) {       // the mirror of the status word in host memory was never written
        c->last_error = "K6-K7 did not report a status";
        rc = MSCKF_ERR_HIP;
    }
    {   // MSCKF_DEBUG_FAKE_TIMEOUT=1 (tests): the first update of a context reads as timed out
        static const bool fake = [] { const char* e = std::getenv("MSCKF_DEBUG_FAKE_TIMEOUT"); return e && std::atoi(e) == 1; }();
        if (fake && !c->fake_timeout_done && rc == MSCKF_OK && c->ran_gain && c->gs_fused_last) { c->fake_timeout_done = true; status[0] = 2; rc = MSCKF_ERR_NOT_SPD; }
    }
    if (rc == MSCKF_ERR_NOT_SPD && (status[0] == 2 || (c->wide_active && status[1] == 2))) {       // k_gain_stream gave up waiting for rows of T or for another workgroup
        // The workgroups of k_root_gain / k_gain_stream wait for each other inside their launch; that they are all resident is
        // argued from their LDS footprint and the device's CU count, not promised by HIP: a partitioned or busy
        // device can keep one out until the 0.5 s bound.  ONE retry on kernels that never wait inside a launch (separate merge
        // levels and root, round 3's K6-K7 launches), and the context stays on them.  Not with split long tracks in the batch
        // (their second source of rows needs the sequential block update).
        if (c->have_features && !c->split_on && !c->retry_plain && c->gs_enabled) {
            c->retry_plain = true;
            c->gs_enabled = false; c->stream_enabled = false;
            if (int r2 = replan_current(c, false)) return r2;
            if (int r2 = run_pipeline(c, true, nullptr)) return r2;
            const int r3 = msckf_get_result(c, dx, P_out, accepted, st);
            c->last_error = "k_gain_stream timed out once: this context now runs its sweeps and K6-K7 as separate launches";
            return r3;
        }
        c->last_error = "k_gain_stream: timeout (the root sweep or a workgroup of the update did not make progress)";
        rc = MSCKF_ERR_HIP;
    }
    if (accepted && c->F > 0 && !gate_done) {
        for (int s = 0; s < c->F; ++s) accepted[c->perm[s]] = (acc_sorted[s] == 1) ? 1 : 0;
    }
    const char* hres = static_cast<const char*>(c->hRes);
    if (dx) {
        if (rc == MSCKF_OK && c->ran_gain) std::memcpy(dx, hres + c->res_dx_off, d * 8);
        else std::memset(dx, 0, d * 8);
    }
    if (P_out) {
        // no-op leaves the covariance untouched (reference early returns MSCKF.py:584-585)
        if (rc == MSCKF_OK && c->ran_gain) { if (c->pool) c->pool->copy(P_out, hres + c->res_p_off, d * d * 8); else std::memcpy(P_out, hres + c->res_p_off, d * d * 8); }
        else HIPCHK(c, hipMemcpy(P_out, c->dP.p, d * d * 8, hipMemcpyDeviceToHost));
    }
    c->us_d2h = (float)(now_us() - t0);
    c->hp[8] += tsync - t0; c->hp[9] += now_us() - tsync;
    c->fetched_serial = c->run_serial; c->fetched_rc = rc;
    if (st) {
        std::memset(st, 0, sizeof(*st));
        st->n_features = c->F - counters[3]; st->n_accepted = n_acc;
        // features whose gate matrix was not SPD never reached the chi-square test (the reference would have raised
        // LinAlgError at MSCKF.py:562): they are reported in not_spd, not counted as gate rejections (:578)
        st->n_rejected = std::max(0, c->F - counters[3] - n_acc - counters[2]);
        st->stacked_rows = counters[1]; st->not_spd = counters[2];
        st->n_leaves = c->n_leaves;
        st->n_levels = (int)c->levels.size() + (c->band_plan ? (int)c->sweep_levels.size() + 1 : 0);
        st->k5_launches = st->n_levels - ((c->band_plan && c->gs_fused_last && c->sweep_mode == 0 && c->root_streamed) ? 1 : 0);   // (a streamed merge level rides in the root's launch)
        st->us_total = c->us_total; st->us_feature = c->us_stage[0]; st->us_qr = c->us_stage[1];
        st->us_gain = c->us_stage[2];
        st->us_host_prep = c->us_host_prep; st->us_h2d = c->us_h2d; st->us_d2h = c->us_d2h;
    }
    return rc;
}

int msckf_commit_covariance(msckf_ctx* c) {
    if (!c || !c->ran || !c->ran_gain) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->fetched_serial == c->run_serial) {          // msckf_get_result has already decided this run: no second read-back
        if (c->fetched_rc != MSCKF_OK) return c->fetched_rc;
        HIPCHK(c, hipMemcpyAsync(c->dP.p, c->dPout.p, (size_t)c->d * c->d * 8, hipMemcpyDeviceToDevice, c->stream));
        c->main_busy = true;                           // (stays in the stream: every reader of P is behind it)
        return MSCKF_OK;
    }
    int counters[4] = {0, 0, 0, 0};
    if (int rc0 = gate_counts(c, counters, nullptr)) return rc0;
    // a non-positive Cholesky pivot leaves garbage in P_out: keep the prior (msckf_get_result reports the same code)
    int status[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(status, c->dStatus.p, 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int n_acc = c->acc_from_dev ? status[2] : (c->acc_override >= 0) ? c->acc_override : counters[0];
    if (n_acc == 0) return MSCKF_NOOP;
    if (status[0] == 2 || (c->wide_active && status[1] == 2)) return MSCKF_ERR_HIP;   // k_gain_stream timed out
    if (status[0] != 0 || ((c->gain_blocked || c->wide_active) && status[1] != 0)) return MSCKF_ERR_NOT_SPD;
    HIPCHK(c, hipMemcpyAsync(c->dP.p, c->dPout.p, (size_t)c->d * c->d * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MSCKF_OK;
}

int msckf_update(msckf_ctx* c, int32_t N, const double* P, const double* cam_R, const double* cam_t,
                 const double* cam_R0, const double* cam_t0, const double* gravity, const double* Kinv,
                 double sigma, int32_t F, const int32_t* view_ptr, const double* obs_uv, const int32_t* obs_slot,
                 const double* idp_base, const double* idp_m, const double* idp_rho, const double* chi2_crit,
                 int32_t n_crit, double* dx, double* P_out, uint8_t* accepted, msckf_stats* stats) {
    if (!c) return MSCKF_ERR_ARG;
    const double tu0 = now_us();
    if (c->main_busy || (c->ran && c->fetched_serial != c->run_serial)) {       // work nobody waited for: the side stream below must not overtake it
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->main_busy = c->feat_busy = c->pose_busy = c->run_pending = false;
    }
    c->defer_state_sync = F > 0;
    c->want_direct = c->direct_enabled && F > 0;
    struct Reset { msckf_ctx* c; ~Reset() { c->want_direct = false; } } reset{c};
    int rc = msckf_set_state(c, N, P, cam_R, cam_t, cam_R0, cam_t0, gravity, Kinv, sigma, chi2_crit, n_crit);
    c->defer_state_sync = false;
    if (rc != MSCKF_OK) { (void)hipStreamSynchronize(c->stream_up); c->state_pending = false; return rc; }
    c->hp[0] += now_us() - tu0;
    c->oneshot = F > 0;
    rc = msckf_set_features(c, F, view_ptr, obs_uv, obs_slot, idp_base, idp_m, idp_rho);
    c->oneshot = false;
    if (rc != MSCKF_OK) {            // uploads / K1-K4 may be in flight
        (void)hipStreamSynchronize(c->stream_up); (void)hipStreamSynchronize(c->stream);
        c->feature_launched = false; c->state_pending = false;
        return rc;
    }
    if (F == 0) {
        // empty feature dict: the reference returns at MSCKF.py:584-585
        const size_t d = c->d;
        if (dx) std::memset(dx, 0, d * 8);
        if (P_out) std::memcpy(P_out, P, d * d * 8);
        if (stats) { std::memset(stats, 0, sizeof(*stats)); }
        return MSCKF_NOOP;
    }
    const double tu1 = now_us();
    rc = run_pipeline(c, true, nullptr);               // K1-K4 is already in the stream (ev[6] sits in front of it)
    if (rc != MSCKF_OK) return rc;
    HIPCHK(c, hipEventRecord(c->ev[7], c->stream));
    c->hp[7] += now_us() - tu1;
    if (c->gate_event && c->gate_direct) {
        // K1-K4's results are in hGate once ev_gate has passed: sum them and fill the caller's mask while K5-K7 run
        const double tg0 = now_us();
        HIPCHK(c, hipEventSynchronize(c->ev_gate));
        if (c->wide_on_stream2) HIPCHK(c, hipEventSynchronize(c->ev_wfeat));      // (the long tracks' results come from the second stream)
        const int Fn = c->F;
        const int* rk = static_cast<const int*>(c->hGate);
        const unsigned char* acc = static_cast<const unsigned char*>(c->hGate) + (size_t)c->Fs * 4;
        const int* perm = c->perm.data();
        const int* hv = c->h_view_sorted.data();
        int cnt[4] = {0, 0, 0, 0};
        for (int s = 0; s < Fn; ++s) {
            const unsigned char a = acc[s];
            if (a == 1) { cnt[0]++; cnt[1] += 2 * (hv[s + 1] - hv[s]) - rk[s]; }
            else if (a == 2) cnt[2]++;
            else if (a == 3) cnt[3]++;
            if (accepted) accepted[perm[s]] = a == 1 ? 1 : 0;
        }
        std::memcpy(c->gate_cnt, cnt, sizeof(cnt));
        c->gate_serial = c->run_serial; c->gate_mask_dst = accepted;
        c->gate_event = false;
        c->hp[11] += now_us() - tg0;
    }
    // the result copies go into the stream right behind the kernels: ONE host wait for kernels + copies
    rc = msckf_get_result(c, dx, P_out, accepted, stats);
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[6], c->ev[7]));
    c->us_total = ms * 1000.0f;
    if (stats) stats->us_total = c->us_total;
    c->hp[10] += now_us() - tu0;
    ++c->hp_calls;
    {   // MSCKF_HOSTPROF=2: the phases of every call that took more than 2 ms (first uses, allocations)
        static const int lvl = [] { const char* e = std::getenv("MSCKF_HOSTPROF"); return e ? std::atoi(e) : 0; }();
        if (lvl >= 2) {
            static thread_local double prev[12] = {0};
            if (c->hp[10] - prev[10] > 2000.0)
                std::fprintf(stderr, "msckf_update call %ld took %.0f us: set_state %.0f | image + validate %.0f, sort %.0f, launch K1-K4 %.0f, plan %.0f, plan upload %.0f | "
                             "K5-K7 launches %.0f | gate %.0f | wait %.0f, unpack %.0f\n", c->hp_calls - 1, c->hp[10] - prev[10], c->hp[0] - prev[0], c->hp[1] - prev[1],
                             c->hp[2] - prev[2], c->hp[4] - prev[4], c->hp[5] - prev[5], c->hp[6] - prev[6], c->hp[7] - prev[7], c->hp[11] - prev[11], c->hp[8] - prev[8],
                             c->hp[9] - prev[9]);
            for (int i = 0; i < 12; ++i) prev[i] = c->hp[i];
        }
    }
    return rc;
}

// ---- f1: get_valid_features -------------------------------------------------
int msckf_set_tracks(msckf_ctx* c, const double* line_base, const double* line_dir, const double* line_conf,
                     const int32_t* lost_for, const int32_t* tracked_for) {
    if (!c) return MSCKF_ERR_ARG;
    if (!c->have_features) return MSCKF_ERR_STATE;
    const int F = c->F, sumM = c->sumM;
    c->have_tracks = false;
    c->use_select = false;
    if (F == 0) { c->have_tracks = true; return MSCKF_OK; }
    if (!line_base || !line_dir || !line_conf || !lost_for || !tracked_for) return MSCKF_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<double> hb((size_t)sumM * 3), hd((size_t)sumM * 3), hc(sumM);
    std::vector<int> hl(F), ht(F);
    for (int sidx = 0; sidx < F; ++sidx) {                 // same permutation as set_features
        const int f = c->perm[sidx];
        const int a = c->h_view_in[f], M = c->h_view_in[f + 1] - a, pos = c->h_view_sorted[sidx];
        std::memcpy(&hb[(size_t)pos * 3], &line_base[(size_t)a * 3], (size_t)M * 24);
        std::memcpy(&hd[(size_t)pos * 3], &line_dir[(size_t)a * 3], (size_t)M * 24);
        std::memcpy(&hc[pos], &line_conf[a], (size_t)M * 8);
        hl[sidx] = lost_for[f];
        ht[sidx] = tracked_for[f];
    }
    int rc = MSCKF_OK;
    auto E = [&](Buf& b, size_t bytes) { if (rc == MSCKF_OK) rc = ensure(c, b, bytes); };
    E(c->dLineBase, (size_t)sumM * 24); E(c->dLineDir, (size_t)sumM * 24); E(c->dLineConf, (size_t)sumM * 8);
    E(c->dLostFor, (size_t)F * 4); E(c->dTrackedFor, (size_t)F * 4); E(c->dSelFlags, (size_t)F); E(c->dWorld, (size_t)F * 24);
    if (rc != MSCKF_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(c->dLineBase.p, hb.data(), (size_t)sumM * 24, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dLineDir.p, hd.data(), (size_t)sumM * 24, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dLineConf.p, hc.data(), (size_t)sumM * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dLostFor.p, hl.data(), (size_t)F * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dTrackedFor.p, ht.data(), (size_t)F * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_tracks = true;
    return MSCKF_OK;
}

int msckf_run_select(msckf_ctx* c, const msckf_select_params* sp) {
    if (!c || !sp) return MSCKF_ERR_ARG;
    if (!c->have_state || !c->have_features || !c->have_tracks) return MSCKF_ERR_STATE;
    if (sp->width < 1 || sp->height < 1) return MSCKF_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    c->use_select = true;
    c->ran = false;
    c->sel_params = *sp;
    if (c->F == 0) return MSCKF_OK;
    SelectArgs a{};
    a.F = c->F;
    a.view_ptr = ptr<int>(c->dViewPtr); a.obs_slot = ptr<int>(c->dObsSlot);
    a.line_base = ptr<double>(c->dLineBase); a.line_dir = ptr<double>(c->dLineDir); a.line_conf = ptr<double>(c->dLineConf);
    a.lost_for = ptr<int>(c->dLostFor); a.tracked_for = ptr<int>(c->dTrackedFor);
    a.cam_R = ptr<double>(c->dCamR); a.cam_t = ptr<double>(c->dCamT);
    for (int i = 0; i < 9; ++i) { a.K[i] = sp->K[i]; a.Kinv[i] = c->Kinv[i]; }
    a.width = sp->width; a.height = sp->height;
    a.min_lost = std::max(sp->min_frames_lost, 1);          // MSCKF.py:119
    a.min_tracked = std::max(sp->min_frames_tracked, 2);    // MSCKF.py:120
    a.use_parallax = sp->use_parallax; a.min_parallax_deg = sp->min_parallax_deg;
    a.flags = ptr<unsigned char>(c->dSelFlags); a.idp_m = ptr<double>(c->dMvec); a.idp_rho = ptr<double>(c->dRho);
    a.world = ptr<double>(c->dWorld);
    hipLaunchKernelGGL(k_select, dim3((c->F * 8 + 255) / 256), dim3(256), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    return MSCKF_OK;
}

int msckf_replan(msckf_ctx* c) {
    if (!c) return MSCKF_ERR_ARG;
    if (!c->use_select || !c->have_features) return MSCKF_ERR_STATE;
    if (c->F == 0) return MSCKF_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const double t0 = now_us();
    std::vector<unsigned char> flags(c->F);
    HIPCHK(c, hipMemcpyAsync(flags.data(), c->dSelFlags.p, c->F, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->plan_valid = false;                 // the plan no longer covers every candidate: the next upload plans afresh
    plan_batch(c, c->h_fmin, c->h_fmax, c->h_view_sorted, &flags);
    c->gather_off = c->rbuf_doubles;
    // the blocks moved inside the R workspace: entries below their diagonals must read as zero again
    {
        const size_t need = (c->rbuf_doubles + 16) * 8;
        if (c->dRbuf.bytes < need) {
            if (c->dRbuf.p) HIPCHK(c, hipFree(c->dRbuf.p));
            c->dRbuf.p = nullptr; c->dRbuf.bytes = 0;
            HIPCHK(c, hipMalloc(&c->dRbuf.p, need + need / 2));
            c->dRbuf.bytes = need + need / 2;
        }
        HIPCHK(c, hipMemsetAsync(c->dRbuf.p, 0, need, c->stream));
    }
    if (int rcp = upload_plan(c)) return rcp;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->us_host_prep = (float)(now_us() - t0);
    c->ran = false;
    return MSCKF_OK;
}

int msckf_debug_time_select(msckf_ctx* c, int32_t iters, float* us_per_launch) {
    if (!c || iters < 1 || !us_per_launch) return MSCKF_ERR_ARG;
    if (!c->use_select) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    const msckf_select_params sp = c->sel_params;
    const bool ran = c->ran;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
    for (int i = 0; i < iters; ++i)
        if (int rc = msckf_run_select(c, &sp)) return rc;       // idempotent: reads lines, rewrites the same outputs
    HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev[5]));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[4], c->ev[5]));
    *us_per_launch = ms * 1000.0f / iters;
    c->ran = ran;                                                // outputs unchanged: results stay valid
    return MSCKF_OK;
}

int msckf_clear_selection(msckf_ctx* c) {
    if (!c) return MSCKF_ERR_ARG;
    c->use_select = false;
    c->ran = false;
    return MSCKF_OK;
}

int msckf_get_selection(msckf_ctx* c, uint8_t* flags, double* idp_m, double* idp_rho, double* world) {
    if (!c) return MSCKF_ERR_ARG;
    if (!c->use_select) return MSCKF_ERR_STATE;
    const int F = c->F;
    if (F == 0) return MSCKF_OK;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<unsigned char> hf(F);
    std::vector<double> hm((size_t)F * 3), hr(F), hw((size_t)F * 3);
    HIPCHK(c, hipMemcpyAsync(hf.data(), c->dSelFlags.p, F, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hm.data(), c->dMvec.p, (size_t)F * 24, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hr.data(), c->dRho.p, (size_t)F * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hw.data(), c->dWorld.p, (size_t)F * 24, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int s = 0; s < F; ++s) {
        const int f = c->perm[s];
        if (flags) flags[f] = hf[s];
        if (idp_rho) idp_rho[f] = hr[s];
        if (idp_m) std::memcpy(&idp_m[(size_t)f * 3], &hm[(size_t)s * 3], 24);
        if (world) std::memcpy(&world[(size_t)f * 3], &hw[(size_t)s * 3], 24);
    }
    return MSCKF_OK;
}

// ---- f4: geometric consistency tests of the matches --------------------------------
int msckf_run_associate(msckf_ctx* c, const msckf_assoc_params* ap, const double* matched_uv, uint8_t* result,
                        int32_t* fail_view) {
    if (!c || !ap) return MSCKF_ERR_ARG;
    if (!c->have_state || !c->have_features) return MSCKF_ERR_STATE;
    const int F = c->F;
    if (F == 0) return MSCKF_OK;
    if (!matched_uv || !result) return MSCKF_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    // inverse of K as the reference forms it for these tests (np.linalg.inv(self.K), MSCKF.py:345)
    const double* K = ap->K;
    const double det = K[0] * (K[4] * K[8] - K[5] * K[7]) - K[1] * (K[3] * K[8] - K[5] * K[6]) + K[2] * (K[3] * K[7] - K[4] * K[6]);
    if (det == 0.0) return MSCKF_ERR_ARG;
    AssocArgs a{};
    a.F = F;
    a.view_ptr = ptr<int>(c->dViewPtr); a.obs_uv = ptr<double>(c->dObsUV); a.obs_slot = ptr<int>(c->dObsSlot);
    a.cam_R = ptr<double>(c->dCamR); a.cam_t = ptr<double>(c->dCamT);
    std::memcpy(a.K, K, 72);
    const double id = 1.0 / det;
    a.Kinv[0] = (K[4] * K[8] - K[5] * K[7]) * id; a.Kinv[1] = (K[2] * K[7] - K[1] * K[8]) * id; a.Kinv[2] = (K[1] * K[5] - K[2] * K[4]) * id;
    a.Kinv[3] = (K[5] * K[6] - K[3] * K[8]) * id; a.Kinv[4] = (K[0] * K[8] - K[2] * K[6]) * id; a.Kinv[5] = (K[2] * K[3] - K[0] * K[5]) * id;
    a.Kinv[6] = (K[3] * K[7] - K[4] * K[6]) * id; a.Kinv[7] = (K[1] * K[6] - K[0] * K[7]) * id; a.Kinv[8] = (K[0] * K[4] - K[1] * K[3]) * id;
    std::memcpy(a.R2, ap->R_cur, 72); std::memcpy(a.t2, ap->t_cur, 24);
    a.thr_epipolar = ap->epipolar_threshold; a.thr_homography = ap->homography_threshold;
    // matched keypoints in sorted feature order; results come back through the same permutation
    std::vector<double> muv((size_t)F * 2);
    for (int s = 0; s < F; ++s) { muv[2 * s] = matched_uv[2 * c->perm[s]]; muv[2 * s + 1] = matched_uv[2 * c->perm[s] + 1]; }
    int rc = MSCKF_OK;
    auto E = [&](Buf& b, size_t bytes) { if (rc == MSCKF_OK) rc = ensure(c, b, bytes); };
    E(c->dAssocUV, (size_t)F * 16); E(c->dAssocRes, (size_t)F * 8);
    if (rc != MSCKF_OK) return rc;
    a.matched_uv = ptr<double>(c->dAssocUV);
    a.fail_view = ptr<int>(c->dAssocRes);
    a.result = ptr<unsigned char>(c->dAssocRes) + (size_t)F * 4;
    HIPCHK(c, hipMemcpyAsync(c->dAssocUV.p, muv.data(), (size_t)F * 16, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_assoc, dim3((F + 255) / 256), dim3(256), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    std::vector<unsigned char> back((size_t)F * 5);
    HIPCHK(c, hipMemcpyAsync(back.data(), c->dAssocRes.p, (size_t)F * 5, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int* fv = reinterpret_cast<const int*>(back.data());
    for (int s = 0; s < F; ++s) {
        result[c->perm[s]] = back[(size_t)F * 4 + s];
        if (fail_view) fail_view[c->perm[s]] = fv[s];
    }
    return MSCKF_OK;
}

// ---- f2 / f3: covariance steps either side of the update, P resident in HBM -----
int msckf_propagate(msckf_ctx* c, const double* Phi, const double* Q) {
    if (!c || !Phi || !Q) return MSCKF_ERR_ARG;
    if (!c->have_state) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    PropagateArgs a{};
    a.P = ptr<double>(c->dP); a.d = c->d;
    std::memcpy(a.Phi, Phi, sizeof(a.Phi));
    std::memcpy(a.Q, Q, sizeof(a.Q));
    const size_t lds = ((size_t)15 * c->d + 225) * 8;
    hipLaunchKernelGGL(k_propagate, dim3(1), dim3(256), lds, c->stream, a);
    if (c->d > 15) {
        const int nb = (c->d - 15 + 15) / 16;
        hipLaunchKernelGGL(k_symmetrize_tail, dim3(nb, nb), dim3(256), 0, c->stream, ptr<double>(c->dP), c->d, 15);
    }
    HIPCHK(c, hipGetLastError());
    c->ran = false;                       // results of a previous update refer to the old prior
    return MSCKF_OK;
}

int msckf_augment(msckf_ctx* c, const double* J15, const double* R, const double* t) {
    if (!c || !J15 || !R || !t) return MSCKF_ERR_ARG;
    if (!c->have_state) return MSCKF_ERR_STATE;
    if (c->N + 1 > c->maxN) return MSCKF_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    AugmentArgs a{};
    a.P = ptr<double>(c->dP); a.out = ptr<double>(c->dB2); a.d = c->d;
    std::memcpy(a.J, J15, sizeof(a.J));
    const int n = c->d + 6;
    hipLaunchKernelGGL(k_augment, dim3((n * n + 255) / 256), dim3(256), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    std::swap(c->dP, c->dB2);             // same capacity (d_max^2); dB2 is scratch of the gain stage
    // the new clone's null pose IS its pose (Camera.py:11)
    c->h_cam[0].insert(c->h_cam[0].end(), R, R + 9); c->h_cam[1].insert(c->h_cam[1].end(), t, t + 3);
    c->h_cam[2].insert(c->h_cam[2].end(), R, R + 9); c->h_cam[3].insert(c->h_cam[3].end(), t, t + 3);
    c->N += 1; c->d = 15 + 6 * c->N; c->dc = 6 * c->N;
    seat_result_views(c);
    if (int rc = upload_poses(c)) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    invalidate_batch(c);
    return MSCKF_OK;
}

int msckf_remove_clones(msckf_ctx* c, int32_t n, const int32_t* slots) {
    if (!c || n < 0 || (n > 0 && !slots)) return MSCKF_ERR_ARG;
    if (!c->have_state) return MSCKF_ERR_STATE;
    if (n == 0) return MSCKF_OK;
    std::vector<char> drop(c->N, 0);
    for (int i = 0; i < n; ++i) {
        if (slots[i] < 0 || slots[i] >= c->N || drop[slots[i]]) return MSCKF_ERR_ARG;
        drop[slots[i]] = 1;
    }
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<int> keep;
    for (int i = 0; i < 15; ++i) keep.push_back(i);
    std::vector<double> nc[4];
    for (int s = 0; s < c->N; ++s) {
        if (drop[s]) continue;
        for (int k = 0; k < 6; ++k) keep.push_back(15 + 6 * s + k);
        for (int q = 0; q < 4; ++q) {
            const int w = (q & 1) ? 3 : 9;
            nc[q].insert(nc[q].end(), c->h_cam[q].begin() + (size_t)s * w, c->h_cam[q].begin() + (size_t)(s + 1) * w);
        }
    }
    const int nn = (int)keep.size();
    HIPCHK(c, hipMemcpyAsync(c->dKeep.p, keep.data(), (size_t)nn * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_compact, dim3((nn * nn + 255) / 256), dim3(256), 0, c->stream, ptr<double>(c->dP), c->d,
                       ptr<int>(c->dKeep), nn, ptr<double>(c->dB2));
    HIPCHK(c, hipGetLastError());
    std::swap(c->dP, c->dB2);
    for (int q = 0; q < 4; ++q) c->h_cam[q].swap(nc[q]);
    c->N -= n; c->d = 15 + 6 * c->N; c->dc = 6 * c->N;
    seat_result_views(c);
    if (int rc = upload_poses(c)) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    invalidate_batch(c);
    return MSCKF_OK;
}

int msckf_set_poses(msckf_ctx* c, const double* cam_R, const double* cam_t, const double* cam_R0, const double* cam_t0) {
    if (!c || !cam_R || !cam_t || !cam_R0 || !cam_t0) return MSCKF_ERR_ARG;
    if (!c->have_state || c->N < 1) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = c->N;
    c->h_cam[0].assign(cam_R, cam_R + N * 9); c->h_cam[1].assign(cam_t, cam_t + N * 3);
    c->h_cam[2].assign(cam_R0, cam_R0 + N * 9); c->h_cam[3].assign(cam_t0, cam_t0 + N * 3);
    if (int rc = upload_poses(c)) return rc;       // (stays in the stream: whatever reads the poses is behind it)
    c->main_busy = true;
    c->ran = false;
    return MSCKF_OK;
}

int msckf_get_covariance(msckf_ctx* c, double* P, int32_t* N) {
    if (!c) return MSCKF_ERR_ARG;
    if (!c->have_state) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    if (N) *N = c->N;
    if (P) {
        HIPCHK(c, hipMemcpyAsync(P, c->dP.p, (size_t)c->d * c->d * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return MSCKF_OK;
}

size_t msckf_block_doubles(const msckf_ctx* c) { return c ? (size_t)c->dc * (c->dc + 1) : 0; }

int msckf_export_block(msckf_ctx* c, void* dst, int device_ptr, int32_t* n_accepted) {
    if (!c || !dst) return MSCKF_ERR_ARG;
    if (!c->ran) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bytes = msckf_block_doubles(c) * 8;
    if (c->F == 0 || c->root < 0) {
        if (device_ptr) HIPCHK(c, hipMemsetAsync(dst, 0, bytes, c->stream));
        else std::memset(dst, 0, bytes);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (n_accepted) *n_accepted = 0;
        return MSCKF_OK;
    }
    if (c->xchg_planned && !c->ran_gain) return MSCKF_ERR_STATE;   // the root sweep was left to the merging rank: msckf_export_groups
    HIPCHK(c, hipMemcpyAsync(dst, root_block(c), bytes, device_ptr ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                             c->stream));
    int counters[4] = {0, 0, 0, 0};
    if (int rc0 = gate_counts(c, counters, nullptr)) return rc0;
    if (n_accepted) *n_accepted = counters[0];
    return MSCKF_OK;
}

struct MergeScope { msckf_ctx* c; explicit MergeScope(msckf_ctx* c_) : c(c_) { c->in_merge = true; } ~MergeScope() { c->in_merge = false; } };

int msckf_run_merge_gain(msckf_ctx* c, const void* blocks, int32_t n_blocks, int device_ptr,
                         int32_t total_accepted) {
    if (!c || !blocks || n_blocks < 1) return MSCKF_ERR_ARG;
    MergeScope merge_scope(c);
    if (!c->have_state) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    const int N = c->N, dc = c->dc;
    const size_t blk = (size_t)dc * (dc + 1);
    // workspace: gathered blocks + merge outputs, behind the local plan's region
    std::vector<FoldNode> nodes;
    std::vector<std::pair<int, int>> levels;
    size_t off = c->gather_off;
    for (int i = 0; i < n_blocks; ++i) {
        FoldNode n{}; n.kind = 1; n.src_begin = 0; n.src_end = 0; n.win_lo = 0; n.w = dc; n.out_off = (long long)off;
        off += blk;
        nodes.push_back(n);
    }
    int base = 0, cnt = n_blocks;
    while (cnt > 1) {
        const int nb = (int)nodes.size();
        for (int i = base; i < base + cnt; i += 2) {
            FoldNode n{}; n.kind = 1; n.src_begin = i; n.src_end = std::min(i + 2, base + cnt); n.win_lo = 0; n.w = dc;
            n.out_off = (long long)off;
            off += blk;
            nodes.push_back(n);
        }
        base = nb; cnt = (int)nodes.size() - nb;
        levels.push_back({base, cnt});
    }
    const size_t need = (off + 16) * 8;
    if (c->dRbuf.bytes < need) {
        // grow, keeping the local plan's blocks
        void* np = nullptr;
        HIPCHK(c, hipMalloc(&np, need));
        HIPCHK(c, hipMemset(np, 0, need));
        if (c->dRbuf.p) {
            HIPCHK(c, hipMemcpy(np, c->dRbuf.p, std::min(c->dRbuf.bytes, c->gather_off * 8), hipMemcpyDeviceToDevice));
            HIPCHK(c, hipFree(c->dRbuf.p));
        }
        c->dRbuf.p = np; c->dRbuf.bytes = need;
    }
    double* rb = ptr<double>(c->dRbuf);
    HIPCHK(c, hipMemsetAsync(rb + c->gather_off, 0, (off - c->gather_off) * 8, c->stream));
    HIPCHK(c, hipMemcpyAsync(rb + c->gather_off, blocks, (size_t)n_blocks * blk * 8,
                             device_ptr ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    // the merge nodes live behind the local plan's nodes in dNodes
    const size_t local_nodes = c->nodes.size();
    std::vector<FoldNode> all(c->nodes);
    for (auto n : nodes) { if (n.src_end > 0) { n.src_begin += (int)local_nodes; n.src_end += (int)local_nodes; } all.push_back(n); }
    for (auto& lv : levels) lv.first += (int)local_nodes;
    if (int rc = ensure(c, c->dNodes, all.size() * sizeof(FoldNode))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->dNodes.p, all.data(), all.size() * sizeof(FoldNode), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!levels.empty()) { if (int rc = launch_fold_levels(c, levels, all)) return rc; }
    const double* root = rb + all.back().out_off;
    // counters[0] decides OK / NOOP in get_result: mark "accepted" when any block is non-empty
    (void)N;
    int rc;
    if (gstream_ok(c, dc)) rc = launch_gain_stream(c, root, dc);
    else rc = launch_gain(c, root);
    if (rc != MSCKF_OK) return rc;
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, c->stream, ptr<int>(c->dStatus) + 2, (int)total_accepted);   // (shared result)
    HIPCHK(c, hipGetLastError());
    c->ran = true; c->ran_gain = true;
    c->acc_override = total_accepted;
    ++c->run_serial; c->run_pending = true;
    return MSCKF_OK;
}

// ---- group exchange: the sharded band pipeline --------------------------------------
int msckf_set_group_exchange(msckf_ctx* c, int on) {
    if (!c) return MSCKF_ERR_ARG;
    c->xchg = on != 0;
    c->have_features = false;             // the next msckf_set_features plans with the new layout
    c->plan_valid = false;
    c->ran = false;
    return MSCKF_OK;
}

int msckf_set_exchange_span(msckf_ctx* c, int32_t max_span) {
    if (!c || max_span < 0) return MSCKF_ERR_ARG;
    c->xchg_span = max_span;
    c->have_features = false;
    c->plan_valid = false;
    c->x_plan_valid = false;
    c->ran = false;
    return MSCKF_OK;
}

int msckf_band_rule(const msckf_ctx* c, int32_t N, int32_t max_span) {
    if (!c || N < 0 || max_span < 0) return MSCKF_ERR_ARG;
    return sweep_mode_for(c, N, max_span) + 1;         // 0 merge tree, 1 k_sweep, 2 k_wsweep<4> (ring), 3 k_wsweep<6> (90-column tiles)
}

size_t msckf_group_record_doubles(const msckf_ctx* c) {
    return c ? rec_head(c) + (size_t)c->N * rec_slot(c) : 0;
}

int msckf_export_groups(msckf_ctx* c, void* dst, int device_ptr, int32_t* n_accepted) {
    if (!c || !dst) return MSCKF_ERR_ARG;
    if (!c->ran) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bytes = msckf_group_record_doubles(c) * 8;
    if (c->F == 0 || c->root < 0) {       // no tracks in this shard: all flags 0
        if (device_ptr) HIPCHK(c, hipMemsetAsync(dst, 0, bytes, c->stream));
        else std::memset(dst, 0, bytes);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (n_accepted) *n_accepted = 0;
        return MSCKF_OK;
    }
    if (!c->xchg_planned) return MSCKF_ERR_STATE;     // tree plan (wide tracks, N > 37): use msckf_export_block
    if (c->ran_gain) return MSCKF_ERR_STATE;           // records come out of msckf_run_compress (it also counts the accepted)
    // the shard's accepted count already sits in the record (double N, k_count_accepted): the merging rank sums them
    HIPCHK(c, hipMemcpyAsync(dst, c->dRbuf.p, bytes, device_ptr ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    if (n_accepted) {                                  // optional: costs a device-to-host copy of the gate results
        int counters[4] = {0, 0, 0, 0};
        if (int rc0 = gate_counts(c, counters, nullptr)) return rc0;       // syncs
        *n_accepted = counters[0];
    } else {
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return MSCKF_OK;
}

namespace {
int run_merge_groups(msckf_ctx* c, const void* records, int32_t n_rec, int device_ptr, int32_t total_accepted, const uint8_t* flags);
}

int msckf_run_merge_groups(msckf_ctx* c, const void* records, int32_t n_rec, int device_ptr, int32_t total_accepted) {
    return run_merge_groups(c, records, n_rec, device_ptr, total_accepted, nullptr);
}

int msckf_run_merge_groups_flags(msckf_ctx* c, const void* records, int32_t n_rec, int device_ptr, const uint8_t* flags) {
    return run_merge_groups(c, records, n_rec, device_ptr, -1, flags);
}

namespace {
// sharded update with msckf_set_exchange_mask: the gate bytes of all shards -> the result range (behind P_out)
int collect_masks(msckf_ctx* c, const double* recs, long long rec_stride, int n_rec) {
    if (c->x_bounds.empty() || c->xmask_doubles == 0) return MSCKF_OK;
    if ((int)c->x_bounds.size() != n_rec + 1) return MSCKF_ERR_ARG;
    MaskBounds mb{};
    mb.n = n_rec;
    for (int i = 0; i <= n_rec; ++i) mb.b[i] = c->x_bounds[i];
    hipLaunchKernelGGL(k_collect_masks, dim3(n_rec), dim3(256), 0, c->stream, recs, rec_stride, c->N + 1, mb,
                       static_cast<unsigned char*>(c->dResArena.p) + c->res_mask_off, ptr<int>(c->dStatus));
    HIPCHK(c, hipGetLastError());
    return MSCKF_OK;
}

int run_merge_groups(msckf_ctx* c, const void* records, int32_t n_rec, int device_ptr, int32_t total_accepted, const uint8_t* flags) {
    if (!c || !records || n_rec < 1) return MSCKF_ERR_ARG;
    MergeScope merge_scope(c);
    if (!c->have_state) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    const int N = c->N, dc = c->dc;
    const int xmode = xchg_mode(c, N);
    if (xmode < 0) return MSCKF_ERR_ARG;
    const size_t XCHG_SLOT = xchg_slot(xmode);
    const int XW = xchg_w(xmode);
    const size_t rec = msckf_group_record_doubles(c);
    // workspace behind the local plan: records | merged group triangles | root block | zero words
    // (60-column plans: merged triangles with rows of 64 doubles, streamable to the root inside its launch -- k_sweep.h)
    const bool xs = xmode == 0 && c->stream_enabled;
    const size_t MSLOT = xs ? (size_t)SWEEP_MAX_W * 64 : XCHG_SLOT;
    const size_t o_rec = c->gather_off, o_mrg = o_rec + (size_t)n_rec * rec, o_root = o_mrg + (size_t)N * MSLOT;
    const size_t o_zero = o_root + (size_t)dc * (dc + 1), o_end = o_zero + 16;
    const size_t need = (o_end + 16) * 8;
    if (c->dRbuf.bytes < need) {          // grow, keeping the local plan's blocks
        void* np = nullptr;
        HIPCHK(c, hipMalloc(&np, need));
        HIPCHK(c, hipMemset(np, 0, need));
        if (c->dRbuf.p) {
            HIPCHK(c, hipMemcpy(np, c->dRbuf.p, std::min(c->dRbuf.bytes, c->gather_off * 8), hipMemcpyDeviceToDevice));
            HIPCHK(c, hipFree(c->dRbuf.p));
        }
        c->dRbuf.p = np; c->dRbuf.bytes = need;
        c->x_plan_valid = false;
    }
    double* rb = ptr<double>(c->dRbuf);
    // records already in HBM are folded where they lie (sources are offsets from the workspace base, which may
    // point outside it); host records are staged behind the local plan
    const double* recs = rb + o_rec;
    if (device_ptr) recs = static_cast<const double*>(records);
    else HIPCHK(c, hipMemcpyAsync(rb + o_rec, records, (size_t)n_rec * rec * 8, hipMemcpyHostToDevice, c->stream));
    const long long rec_base = (long long)(recs - rb);
    // which groups does each record carry?  (N flags and the accepted count at the head of every record)
    std::vector<double> key((size_t)n_rec * N);
    bool count_on_device = false;
    if (flags) {
        // the caller knows the groups of every shard (it partitioned the batch): nothing is read back; the accepted
        // counts of the records are summed on the device into status word 2
        for (size_t i = 0; i < key.size(); ++i) key[i] = flags[i] ? 1.0 : 0.0;
        hipLaunchKernelGGL(k_sum_record_counts, dim3(1), dim3(64), 0, c->stream, recs, (long long)rec, N, n_rec, ptr<int>(c->dStatus) + 2);
        HIPCHK(c, hipGetLastError());
        count_on_device = true;
        total_accepted = -1;
    } else {
        std::vector<double> head((size_t)n_rec * (N + 1));
        HIPCHK(c, hipMemcpy2DAsync(head.data(), (size_t)(N + 1) * 8, recs, rec * 8, (size_t)(N + 1) * 8, n_rec,
                                   hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        double acc_sum = 0.0;
        for (int r = 0; r < n_rec; ++r) {
            std::copy(head.begin() + (size_t)r * (N + 1), head.begin() + (size_t)r * (N + 1) + N, key.begin() + (size_t)r * N);
            acc_sum += head[(size_t)r * (N + 1) + N];
        }
        if (total_accepted < 0) total_accepted = (int32_t)(acc_sum + 0.5);    // the counts the shards wrote into their records
        hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, c->stream, ptr<int>(c->dStatus) + 2, (int)total_accepted);
        HIPCHK(c, hipGetLastError());
    }
    const bool reuse = c->x_plan_valid && c->x_nrec == n_rec && c->x_root_off == o_root && c->x_rec_base == rec_base &&
                       c->x_key == key;
    if (!reuse) {
        c->x_snodes.clear(); c->x_sfolds.clear();
        const int fold_base = (int)c->sfolds.size();      // the tables sit behind the local plan's
        struct Tri { long long src; int lo, w; int prod = 0, ld = 0; };
        std::vector<Tri> groups;
        std::vector<SweepFold>& fl = c->x_sfolds;
        for (int s0 = 0; s0 < N; ++s0) {
            const int w = 6 * (std::min(s0 + XW / 6, N) - s0);
            std::vector<long long> src;
            for (int r = 0; r < n_rec; ++r)
                if (key[(size_t)r * N + s0] != 0.0) src.push_back(rec_base + (long long)((size_t)r * rec + rec_head(c) + (size_t)s0 * XCHG_SLOT));
            if (src.empty()) continue;
            if (src.size() == 1) { groups.push_back({src[0], s0, w}); continue; }
            SweepNode m{};
            m.fold_begin = fold_base + (int)fl.size();
            const int b = (int)fl.size();
            for (long long so : src) { SweepFold sf{}; sf.src_off = so; sf.off = 0; sf.w = w; sf.ew = w; fl.push_back(sf); }
            m.fold_end = fold_base + (int)fl.size();
            m.wtot = w;
            sweep_schedule(fl, b, (int)fl.size(), &m.nsteps);
            m.out_off = (long long)(o_mrg + (size_t)s0 * MSLOT);
            m.ldo = xs ? 64 : 0;
            c->x_snodes.push_back(m);
            groups.push_back({m.out_off, s0, w, (int)c->x_snodes.size(), m.ldo});
        }
        c->x_n_merges = (int)c->x_snodes.size();
        // the merge level rides in the root's launch and streams its rows to the root (as the local plan's last level does)
        c->x_streamed = xs && c->x_n_merges >= 1 && c->x_n_merges <= 64 && groups.size() > 1 && root_gain_ok(c, XW) &&
                        2 * (2 + (dc + 15) / 16 + c->x_n_merges) <= c->n_cu;
        if (!groups.empty()) {
            SweepNode r{};
            r.fold_begin = fold_base + (int)fl.size();
            const int b = (int)fl.size();
            int env = 0;
            for (const Tri& g : groups) {
                env = std::max(env, 6 * g.lo + g.w);
                SweepFold sf{}; sf.src_off = g.src; sf.off = 6 * g.lo; sf.w = g.w; sf.ew = env - 6 * g.lo; sf.ld = g.ld;
                if (c->x_streamed) sf.prod = g.prod;
                fl.push_back(sf);
            }
            r.fold_end = fold_base + (int)fl.size();
            r.wtot = dc;
            sweep_schedule(fl, b, (int)fl.size(), &r.nsteps, SWEEP_NW, !c->x_streamed);
            r.out_off = (long long)o_root;
            c->x_snodes.push_back(r);
            c->x_root_flush.clear();
            if (xmode == 0) sweep_flush_table(fl, b, (int)fl.size(), r.nsteps, r.wtot, 1 << 29, c->x_root_flush);
            c->x_root_n_gate = -1;
            if (c->x_streamed) {
                c->x_root_n_gate = sweep_gate_table(fl, b, (int)fl.size(), r.nsteps, SWEEP_NW, c->x_root_flush);
                c->x_mflush_at = (int)c->x_root_flush.size();
                const int nm = c->x_n_merges;
                c->x_root_flush.resize(c->x_root_flush.size() + nm, 0);
                for (int i = 0; i < nm; ++i) {
                    const SweepNode& m = c->x_snodes[i];
                    c->x_root_flush[c->x_mflush_at + i] = (int)c->x_root_flush.size() - c->x_mflush_at - nm;
                    sweep_flush_table(fl, m.fold_begin - fold_base, m.fold_end - fold_base, m.nsteps, m.wtot, 1 << 29, c->x_root_flush);
                }
            }
        }
        // tables: local plan first (its launches may follow this call), the merge plan behind it
        std::vector<SweepNode> all_n(c->snodes);
        all_n.insert(all_n.end(), c->x_snodes.begin(), c->x_snodes.end());
        std::vector<SweepFold> all_f(c->sfolds);
        all_f.insert(all_f.end(), fl.begin(), fl.end());
        if (xmode > 0) {
            // k_wsweep: which rows of R leave the ring at the head of every macro step (tables behind the local plan's)
            const int rc = 1 << (xmode == 1 ? WS_RC_LOG2_4 : WS_RC_LOG2_6);
            std::vector<int> fl_tab(c->h_flush), fl_off(c->h_flush_off);
            fl_off.resize(c->snodes.size(), 0);
            for (const SweepNode& nd : c->x_snodes) {
                const size_t o = fl_tab.size();
                fl_off.push_back((int)o);
                if (!sweep_flush_table(fl, nd.fold_begin - fold_base, nd.fold_end - fold_base, nd.nsteps, nd.wtot, rc, fl_tab)) {
                    c->last_error = "merge plan: the band does not fit the ring of k_wsweep";
                    return MSCKF_ERR_ARG;
                }
                sweep_publish_table(fl_tab, o, nd.nsteps, nd.wtot);
            }
            if (int rc2 = ensure(c, c->dFlush, fl_tab.size() * 4)) return rc2;
            if (int rc2 = ensure(c, c->dFlushOff, fl_off.size() * 4)) return rc2;
            HIPCHK(c, hipMemcpyAsync(c->dFlush.p, fl_tab.data(), fl_tab.size() * 4, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->dFlushOff.p, fl_off.data(), fl_off.size() * 4, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));                              // fl_tab / fl_off are locals
        }
        if (int rc = ensure(c, c->dSweepNodes, std::max<size_t>(all_n.size(), 1) * sizeof(SweepNode))) return rc;
        if (int rc = ensure(c, c->dSweepFolds, std::max<size_t>(all_f.size(), 1) * sizeof(SweepFold))) return rc;
        if (!all_n.empty()) HIPCHK(c, hipMemcpyAsync(c->dSweepNodes.p, all_n.data(), all_n.size() * sizeof(SweepNode), hipMemcpyHostToDevice, c->stream));
        if (!all_f.empty()) HIPCHK(c, hipMemcpyAsync(c->dSweepFolds.p, all_f.data(), all_f.size() * sizeof(SweepFold), hipMemcpyHostToDevice, c->stream));
        if (!c->x_root_flush.empty()) {
            if (int rc = ensure(c, c->dXRootFlush, c->x_root_flush.size() * 4)) return rc;
            HIPCHK(c, hipMemcpyAsync(c->dXRootFlush.p, c->x_root_flush.data(), c->x_root_flush.size() * 4, hipMemcpyHostToDevice, c->stream));
        }
        HIPCHK(c, hipMemsetAsync(rb + o_mrg, 0, (o_end - o_mrg) * 8, c->stream));   // merged triangles, root block, zero words
        HIPCHK(c, hipStreamSynchronize(c->stream));                                  // all_n / all_f are locals
        c->x_key = key; c->x_nrec = n_rec; c->x_root_off = o_root; c->x_zero_off = o_zero; c->x_rec_base = rec_base;
        c->x_plan_valid = true;
    }
    if (c->x_snodes.empty()) {            // no shard has a track: nothing to update
        if (!count_on_device) {
            hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, c->stream, ptr<int>(c->dStatus) + 2, 0);
            HIPCHK(c, hipGetLastError());
        }
        if (int rcm = collect_masks(c, recs, (long long)rec, n_rec)) return rcm;
        c->ran = true; c->ran_gain = false; c->acc_override = 0; c->acc_from_dev = false;
        ++c->run_serial; c->run_pending = true;
        return MSCKF_OK;
    }
    const int nb = (int)c->snodes.size();
    if (xmode > 0) {
        // ring-buffered sweeps (N > 37 or tracks of 11-15 slots): cross-rank group merges in one launch, then the root
        int ms = 0;
        for (int i = 0; i < c->x_n_merges; ++i) ms = std::max(ms, c->x_snodes[i].nsteps);
        const int rs = c->x_snodes.back().nsteps;
        const bool fused = root_gain_w_ok(c, XW);               // the root sweep and K6-K7 in one launch (k_root_gain_w)
        if (xmode == 1) {
            if (c->x_n_merges > 0) launch_wsweep<4>(c, nb, c->x_n_merges, WS_RC_LOG2_4, ms, rb + c->x_zero_off);
            if (!fused) launch_wsweep<4>(c, nb + c->x_n_merges, 1, WS_RC_LOG2_4, rs, rb + c->x_zero_off);
        } else {
            if (c->x_n_merges > 0) launch_wsweep<6>(c, nb, c->x_n_merges, WS_RC_LOG2_6, ms, rb + c->x_zero_off);
            if (!fused) launch_wsweep<6>(c, nb + c->x_n_merges, 1, WS_RC_LOG2_6, rs, rb + c->x_zero_off);
        }
        HIPCHK(c, hipGetLastError());
        int rcg;
        if (fused && xmode == 1) rcg = launch_root_and_gain_w<4>(c, nb + c->x_n_merges, rs, WS_RC_LOG2_4, rb + c->x_zero_off, rb + c->x_root_off, XW);
        else if (fused) rcg = launch_root_and_gain_w<6>(c, nb + c->x_n_merges, rs, WS_RC_LOG2_6, rb + c->x_zero_off, rb + c->x_root_off, XW);
        else if (gstream_ok(c, XW)) rcg = launch_gain_stream(c, rb + c->x_root_off, XW);
        else rcg = launch_gain(c, rb + c->x_root_off);
        if (rcg != MSCKF_OK) return rcg;
        if (int rcm = collect_masks(c, recs, (long long)rec, n_rec)) return rcm;
        c->ran = true; c->ran_gain = true;
        c->acc_override = total_accepted;
        c->acc_from_dev = count_on_device;
        ++c->run_serial; c->run_pending = true;
        return MSCKF_OK;
    }
    SweepArgs a{};
    a.nodes = ptr<SweepNode>(c->dSweepNodes);
    a.folds = ptr<SweepFold>(c->dSweepFolds);
    a.rbuf = rb;
    a.stamps = nullptr;
    a.zero = rb + c->x_zero_off;
    const dim3 block(64 * SWEEP_NW * SWEEP_WPF);
    const bool fused = root_gain_ok(c, XW) && !c->x_root_flush.empty();
    const bool ride_on = fused && c->x_streamed;
    if (c->x_n_merges > 0 && !ride_on) {
        a.node_base = nb;
        hipLaunchKernelGGL((k_sweep<SWEEP_NW, SWEEP_WPF>), dim3(c->x_n_merges), block,
                           sweep_lds_bytes(SWEEP_MAX_W, SWEEP_NW, SWEEP_WPF), c->stream, a);
    }
    a.node_base = nb + c->x_n_merges;
    int rc;
    if (fused) {
        MergeRide ride{};
        if (ride_on) {
            ride = MergeRide{nb, c->x_n_merges, SWEEP_NW, c->x_root_n_gate, ptr<int>(c->dXRootFlush) + c->x_mflush_at, 0};
            for (int i = 0; i < c->x_n_merges; ++i)
                ride.lds = std::max(ride.lds, sweep_lds_bytes_fl(c->x_snodes[i].wtot, SWEEP_NW, c->x_snodes[i].nsteps));
        }
        rc = launch_root_and_gain(c, a, dc, c->x_snodes.back().nsteps, ptr<int>(c->dXRootFlush), rb + c->x_root_off, XW, ride_on ? &ride : nullptr);
    } else {
        hipLaunchKernelGGL((k_sweep<SWEEP_NW, SWEEP_WPF, SWEEP_P2P>), dim3(1), block, sweep_lds_bytes(dc, SWEEP_NW, SWEEP_WPF), c->stream, a);
        HIPCHK(c, hipGetLastError());
        if (gstream_ok(c, XW)) rc = launch_gain_stream(c, rb + c->x_root_off, XW);
        else rc = launch_gain(c, rb + c->x_root_off);
    }
    if (rc != MSCKF_OK) return rc;
    if (int rcm = collect_masks(c, recs, (long long)rec, n_rec)) return rcm;
    c->ran = true; c->ran_gain = true;
    c->acc_override = total_accepted;
    c->acc_from_dev = count_on_device;
    ++c->run_serial; c->run_pending = true;
    return MSCKF_OK;
}
}  // namespace

// ---- RCCL exchange ---------------------------------------------------------------------
int msckf_comm_unique_id(void* id_out) {
    if (!id_out) return MSCKF_ERR_ARG;
    RcclApi& a = rccl();
    if (!a.lib) return MSCKF_ERR_COMM;
    static_assert(sizeof(ncclUniqueId) == MSCKF_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    if (a.GetUniqueId(&id) != ncclSuccess) return MSCKF_ERR_COMM;
    std::memcpy(id_out, &id, sizeof(id));
    return MSCKF_OK;
}

int msckf_comm_init(msckf_ctx* c, int32_t rank, int32_t world, const void* id) {
    if (!c || !id || world < 1 || rank < 0 || rank >= world) return MSCKF_ERR_ARG;
    if (c->comm) return MSCKF_ERR_STATE;
    RcclApi& a = rccl();
    if (!a.lib) { c->last_error = a.err; return MSCKF_ERR_COMM; }
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    NCCLCHK(c, a.CommInitRank(&c->comm, world, uid, rank));
    c->comm_rank = rank; c->comm_world = world;
    return MSCKF_OK;
}

int msckf_comm_destroy(msckf_ctx* c) {
    if (!c) return MSCKF_ERR_ARG;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)rccl().CommDestroy(c->comm);
        c->comm = nullptr;
    }
    return MSCKF_OK;
}

int msckf_comm_gather(msckf_ctx* c, const void* send, void* recv, size_t count, int32_t root) {
    if (!c || !send || (c->comm_rank == root && !recv)) return MSCKF_ERR_ARG;
    if (!c->comm) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    NCCLCHK(c, rccl().Gather(send, recv, count, ncclDouble, root, c->comm, c->stream));
    return MSCKF_OK;
}

int msckf_comm_broadcast(msckf_ctx* c, void* buf, size_t count, int32_t root) {
    if (!c || !buf) return MSCKF_ERR_ARG;
    if (!c->comm) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    NCCLCHK(c, rccl().Broadcast(buf, buf, count, ncclDouble, root, c->comm, c->stream));
    return MSCKF_OK;
}

int msckf_comm_allreduce(msckf_ctx* c, void* buf, size_t count, int32_t op) {
    if (!c || !buf || (op != 0 && op != 1)) return MSCKF_ERR_ARG;
    if (!c->comm) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    NCCLCHK(c, rccl().AllReduce(buf, buf, count, ncclDouble, op == 0 ? ncclSum : ncclMax, c->comm, c->stream));
    return MSCKF_OK;
}

void* msckf_comm_buffer(msckf_ctx* c, size_t bytes) {
    if (!c) return nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    if (c->dCommBuf.bytes < bytes) (void)hipStreamSynchronize(c->stream);       // a collective may still use the old buffer
    if (ensure(c, c->dCommBuf, bytes) != MSCKF_OK) return nullptr;
    return c->dCommBuf.p;
}

int msckf_comm_put(msckf_ctx* c, void* dst_device, const void* src_host, size_t bytes) {
    if (!c || !dst_device || !src_host) return MSCKF_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MSCKF_OK;
}

int msckf_comm_get(msckf_ctx* c, void* dst_host, const void* src_device, size_t bytes) {
    if (!c || !dst_host || !src_device) return MSCKF_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MSCKF_OK;
}

int msckf_export_result(msckf_ctx* c, void* dx_dst, void* P_dst, int device_ptr) {
    if (!c) return MSCKF_ERR_ARG;
    if (!c->ran || !c->ran_gain) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t d = c->d;
    const hipMemcpyKind kind = device_ptr ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (dx_dst) HIPCHK(c, hipMemcpyAsync(dx_dst, c->dDx.p, d * 8, kind, c->stream));
    if (P_dst) HIPCHK(c, hipMemcpyAsync(P_dst, c->dPout.p, d * d * 8, kind, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MSCKF_OK;
}

int msckf_import_covariance(msckf_ctx* c, const void* P, int device_ptr) {
    if (!c || !P) return MSCKF_ERR_ARG;
    if (!c->have_state) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t d = c->d;
    HIPCHK(c, hipMemcpyAsync(c->dP.p, P, d * d * 8, device_ptr ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MSCKF_OK;
}

int msckf_set_exchange_mask(msckf_ctx* c, int32_t n_shards, const int32_t* bounds) {
    if (!c || n_shards < 0 || n_shards > 64 || (n_shards > 0 && !bounds)) return MSCKF_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int cap = 0;
    for (int r = 0; r < n_shards; ++r) {
        if (bounds[r + 1] < bounds[r] || bounds[0] != 0) return MSCKF_ERR_ARG;
        cap = std::max(cap, bounds[r + 1] - bounds[r]);
    }
    c->x_bounds.clear();
    if (n_shards > 0) c->x_bounds.assign(bounds, bounds + n_shards + 1);
    c->xmask_doubles = (cap + 7) / 8;
    c->have_features = false;             // the record layout changed: the next msckf_set_features plans afresh
    c->plan_valid = false;
    c->x_plan_valid = false;
    c->ran = false;
    const size_t total = n_shards > 0 ? (size_t)bounds[n_shards] : 0;
    if (total > c->res_mask_cap) {        // the result range grows: nothing may be in flight on it
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const size_t dmax = 15 + 6 * (size_t)c->maxN;
        const size_t mcap = (total + 7) & ~(size_t)7;
        const size_t cap_bytes = 64 + dmax * 8 + dmax * dmax * 8 + mcap;
        void* np = nullptr;
        void* nh = nullptr;
        HIPCHK(c, hipMalloc(&np, cap_bytes));
        HIPCHK(c, hipMemset(np, 0, cap_bytes));
        if (hipHostMalloc(&nh, cap_bytes) != hipSuccess) { (void)hipFree(np); return MSCKF_ERR_HIP; }
        (void)hipFree(c->dResArena.p);
        (void)hipHostFree(c->hRes);
        c->dResArena.p = np; c->dResArena.bytes = cap_bytes;
        c->hRes = nh;
        c->res_mask_cap = mcap; c->res_cap = cap_bytes;
        seat_result_views(c);
    }
    return MSCKF_OK;
}

size_t msckf_result_range_doubles(const msckf_ctx* c) {
    if (!c) return 0;
    const size_t total = c->x_bounds.empty() ? 0 : (size_t)c->x_bounds.back();
    return 8 + (size_t)c->d + (size_t)c->d * c->d + (total + 7) / 8;
}

int msckf_get_shared_result(msckf_ctx* c, double* dx, double* P_out, uint8_t* accepted, msckf_stats* st) {
    if (!c) return MSCKF_ERR_ARG;
    if (!c->have_state) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t d = c->d;
    const size_t total = c->x_bounds.empty() ? 0 : (size_t)c->x_bounds.back();
    HIPCHK(c, hipMemcpyAsync(c->hRes, c->dResArena.p, c->res_mask_off + total, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int status[4];
    std::memcpy(status, c->hRes, 16);
    const int dc = c->dc;
    // (as launch_gain decides; k_gain_stream -- dtype f64 -- uses the first status word only)
    const bool blocked = !(c->gs_enabled && (dc + 15) / 16 + 1 <= GS_MAX_NS) &&
                         dc > 4 * CHOL_TILE_MAX_NT && dc <= 2 * GAIN_BLK && dc - GAIN_BLK >= 4;
    const int n_acc = status[2];
    int rc = (n_acc <= 0) ? MSCKF_NOOP : MSCKF_OK;
    if (rc == MSCKF_OK && (status[0] != 0 || (blocked && status[1] != 0))) rc = status[0] == 2 ? MSCKF_ERR_HIP : MSCKF_ERR_NOT_SPD;
    const char* hres = static_cast<const char*>(c->hRes);
    if (dx) {
        if (rc == MSCKF_OK) std::memcpy(dx, hres + c->res_dx_off, d * 8);
        else std::memset(dx, 0, d * 8);
    }
    if (P_out) {
        if (rc == MSCKF_OK) std::memcpy(P_out, hres + c->res_p_off, d * d * 8);
        else HIPCHK(c, hipMemcpy(P_out, c->dP.p, d * d * 8, hipMemcpyDeviceToHost));       // untouched prior (MSCKF.py:584-585)
    }
    int n_rej = 0, n_nspd = 0, n_unsel = 0;
    const unsigned char* mk = reinterpret_cast<const unsigned char*>(hres + c->res_mask_off);
    for (size_t i = 0; i < total; ++i) {
        const unsigned char a = mk[i];
        if (accepted) accepted[i] = a == 1 ? 1 : 0;
        n_rej += a == 0; n_nspd += a == 2; n_unsel += a == 3;
    }
    if (st) {
        std::memset(st, 0, sizeof(*st));
        st->n_features = (int)total - n_unsel; st->n_accepted = std::max(n_acc, 0);
        st->n_rejected = n_rej; st->not_spd = n_nspd;
    }
    return rc;
}

int msckf_debug_gate(msckf_ctx* c, double* gamma, int32_t* qdim) {
    if (!c || !c->ran) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<double> g(c->F);
    std::vector<int> rk(c->F);
    if (c->F == 0) return MSCKF_OK;
    HIPCHK(c, hipMemcpy(g.data(), c->dGamma.p, (size_t)c->F * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(rk.data(), c->dRank.p, (size_t)c->F * 4, hipMemcpyDeviceToHost));
    for (int s = 0; s < c->F; ++s) {
        const int f = c->perm[s];
        if (gamma) gamma[f] = g[s];
        if (qdim) qdim[f] = 2 * (c->h_view_sorted[s + 1] - c->h_view_sorted[s]) - rk[s];
    }
    return MSCKF_OK;
}

int msckf_debug_compressed(msckf_ctx* c, double* T, double* rn) {
    if (!c || !c->ran) return MSCKF_ERR_STATE;
    // split long tracks: K6-K7 took two sources of rows -- the band root and the remainder blocks' rows (as they are, or the root
    // of their own tree).  For this diagnostic ONE [T | r_n]: the Householder QR of both stacked, on the host.
    const bool tops = c->wide_active && !c->rem_direct && !c->rtops.empty();
    const bool second = c->wide_active && (c->rem_direct || c->rroot >= 0 || tops);
    if (c->root < 0 && !second) return MSCKF_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream2));
    const int dc = c->dc, n = dc + 1;
    std::vector<double> blk((size_t)dc * n, 0.0);
    if (c->root >= 0) HIPCHK(c, hipMemcpy(blk.data(), root_block(c), blk.size() * 8, hipMemcpyDeviceToHost));
    if (second) {
        int m2 = dc;
        if (c->rem_direct) {                                   // (the rows k_rem_scatter laid down, rounded up to whole blocks)
            int nr[2] = {0, 0};
            HIPCHK(c, hipMemcpy(nr, ptr<double>(c->dRem) + (size_t)16 * GS_MAX_NB2 * (6 * c->maxN + 1), 8, hipMemcpyDeviceToHost));
            m2 = 16 * nr[0];
        }
        if (tops) m2 = (c->rtop_rows + 15) / 16 * 16;            // (the triangles a cut tree ended with, as k_tri_gather laid them down)
        const double* src2 = (c->rem_direct || tops) ? ptr<double>(c->dRem) : ptr<double>(c->dRbuf) + c->rroot_off;
        std::vector<double> A((size_t)(dc + m2) * n, 0.0);
        std::memcpy(A.data(), blk.data(), blk.size() * 8);
        HIPCHK(c, hipMemcpy(A.data() + (size_t)dc * n, src2, (size_t)m2 * n * 8, hipMemcpyDeviceToHost));
        for (int k = 0; k < dc; ++k) {                          // row k of the triangle and the rows behind it hold column k
            double nrm2 = A[(size_t)k * n + k] * A[(size_t)k * n + k];
            for (int i = dc; i < dc + m2; ++i) nrm2 += A[(size_t)i * n + k] * A[(size_t)i * n + k];
            if (nrm2 == 0.0) continue;
            const double xk = A[(size_t)k * n + k], nrm = std::sqrt(nrm2), alpha = xk > 0.0 ? -nrm : nrm;
            const double vk = xk - alpha, beta = 1.0 / (nrm * (nrm + std::fabs(xk)));
            for (int j = k + 1; j < n; ++j) {
                double dot = vk * A[(size_t)k * n + j];
                for (int i = dc; i < dc + m2; ++i) dot += A[(size_t)i * n + k] * A[(size_t)i * n + j];
                const double w = beta * dot;
                A[(size_t)k * n + j] -= w * vk;
                for (int i = dc; i < dc + m2; ++i) A[(size_t)i * n + j] -= w * A[(size_t)i * n + k];
            }
            A[(size_t)k * n + k] = alpha;
            for (int i = dc; i < dc + m2; ++i) A[(size_t)i * n + k] = 0.0;
        }
        std::memcpy(blk.data(), A.data(), blk.size() * 8);
    }
    for (int i = 0; i < dc; ++i) {
        if (T) for (int j = 0; j < dc; ++j) T[(size_t)i * dc + j] = (j >= i) ? blk[(size_t)i * (dc + 1) + j] : 0.0;
        if (rn) rn[i] = blk[(size_t)i * (dc + 1) + dc];
    }
    return MSCKF_OK;
}

int msckf_debug_split(msckf_ctx* c, int32_t out[8]) {
    if (!c || !out) return MSCKF_ERR_ARG;
    if (!c->have_features) return MSCKF_ERR_STATE;
    out[0] = c->split_on ? c->Fw : 0;
    out[1] = c->nNarrow;
    out[2] = c->split_on ? c->rem_cap : 0;
    out[3] = c->wide_active ? (c->rem_direct ? 1 : ((c->rroot >= 0 || !c->rtops.empty()) ? 2 : 0)) : 0;
    out[4] = (int)c->rlevels.size();
    out[5] = c->Fs;
    out[6] = c->band_plan ? 1 : 0;
    out[7] = c->band_plan ? c->sweep_mode : -1;
    return MSCKF_OK;
}

int msckf_debug_set_rem_direct_rows(msckf_ctx* c, int32_t rows) {
    if (!c) return MSCKF_ERR_ARG;
    c->rem_direct_max = rows < 0 ? msckf_ctx::REM_DIRECT_DEFAULT : std::min(rows, 16 * GS_MAX_NB2);
    c->rem_direct_max_wide = rows < 0 ? 16 * GS_MAX_NB2 : c->rem_direct_max;
    c->plan_valid = false;
    return MSCKF_OK;
}

int msckf_debug_fold_stamps(msckf_ctx* c, long long* out, int32_t max_nodes) {
    // out == NULL: enable stamping for the following runs; else copy 8 values per node
    if (!c) return MSCKF_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!out) { return ensure(c, c->dStamps, (size_t)65536 * 8 * 8, true); }
    if (!c->dStamps.p) return MSCKF_ERR_STATE;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (max_nodes == -1000000) {   // per-wave cycle sums of the last fold launch's first node
        HIPCHK(c, hipMemcpy(out, ptr<long long>(c->dStamps) + 16 * 8192, 64 * 8, hipMemcpyDeviceToHost));
        return 64;
    }
    if (max_nodes < 0) {   // feature-kernel stamps: -max_nodes features
        const size_t nf = std::min<size_t>((size_t)(-max_nodes), (size_t)c->F);
        HIPCHK(c, hipMemcpy(out, ptr<long long>(c->dStamps) + 8 * 8192, nf * 64, hipMemcpyDeviceToHost));
        return (int)nf;
    }
    const size_t n = std::min<size_t>((size_t)max_nodes, c->nodes.size() + c->snodes.size());
    HIPCHK(c, hipMemcpy(out, c->dStamps.p, n * 64, hipMemcpyDeviceToHost));
    return (int)n;
}

uint64_t msckf_device_pointer(msckf_ctx* c, int which) {
    if (!c) return 0;
    switch (which) {
        case 0: return (uint64_t)c->dDx.p;
        case 1: return (uint64_t)c->dPout.p;
        case 2: return (c->root >= 0) ? (uint64_t)root_block(c) : 0;
        case 3: return c->xchg_planned ? (uint64_t)c->dRbuf.p : 0;
        case 4: return (uint64_t)c->dP.p;
        case 5: return (uint64_t)c->dResArena.p;                                   // status | dx | P_out | gate bytes
        case 6: return (uint64_t)(static_cast<char*>(c->dResArena.p) + c->res_mask_off);
        default: return 0;
    }
}

void* msckf_stream(msckf_ctx* c) { return c ? (void*)c->stream : nullptr; }

}  // extern "C"
