// The tracks of a batch, brought into the pipeline's order ON THE DEVICE.
//   reference MSCKF.py:570-582 : `for feature in features.values()` -- the reference walks its dict in insertion order;
// the engine's kernels want the tracks sorted by (class, first slot, last slot) (k_lsweep's leaves, the per-class launches
// of k_feature).  Until round 3 the host gathered every array into that order before the upload (29 us of the call at 2000
// tracks, 155 us at 10000, all of it in front of K1-K4).  Now the caller's arrays stay AS THEY ARE: the observations (60 % of
// the bytes) cross PCIe by DMA while the host is still validating and sorting; the small arrays follow by DMA (large batches)
// or through k_stage, a copy kernel that reads the pinned image in whole cache lines (small ones: no copy command, and none
// of the ~9 us a copy command waits behind its predecessor); the host's sort contributes one 24-byte record per track, which
// this kernel reads where it lies (pinned host memory for small batches).  It writes the sorted image the other kernels
// read: an HBM-bound permutation, one 32-lane group per track.
#pragma once
#include <hip/hip_runtime.h>
#include "k_lsweep.h"

namespace msckf {

struct __attribute__((aligned(8))) GatherRec {
    int f;                        // input index of the track at this sorted position
    int a;                        // its first view in the caller's arrays
    int M;                        // views
    int o;                        // its first view in the sorted arrays
    long long blk;                // offset of its K4 block
};

struct GatherArgs {
    // the caller's arrays, input order (uv: HBM, uploaded by DMA; the others: the pinned host image)
    const double* uv_in;          // [sumM][2]
    const int* slot_in;           // [sumM]
    const double* base_in;        // [F][3]
    const double* m_in;           // [F][3]
    const double* rho_in;         // [F]
    const GatherRec* rec;         // [F] from the host's sort (pinned host image)
    // the sorted image
    double* uv; double* base; double* m; double* rho;
    int* slot; int* fmin; int* view; int* perm;
    long long* blk;
    FeatInfo* info;
    int F, sumM;
};

constexpr int GATHER_THREADS = 256;

// The small arrays of a small batch, pinned host image -> HBM in whole cache lines (a few thousand tracks: cheaper than a copy
// command and the ~9 us it waits behind its predecessor, and 5x fewer PCIe requests than k_gather's scattered reads of them).
__global__ __launch_bounds__(256) void k_stage(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

__global__ __launch_bounds__(GATHER_THREADS) void k_gather(GatherArgs p) {
    __shared__ unsigned long long scol[GATHER_THREADS / 32][2];
    const int t = threadIdx.x, grp = t >> 5, v = t & 31;
    const int s = blockIdx.x * (GATHER_THREADS / 32) + grp;
    const bool live = s < p.F;
    // (the records may sit in pinned host memory: lanes 0 - 2 of a group fetch one 8-byte word each, so that the two groups of
    //  a wavefront make ONE request for their 48 contiguous bytes instead of 64 lanes asking for the same 24)
    GatherRec r{0, 0, 0, 0, 0};
    {
        long long w = 0;
        if (live && v < 3) w = reinterpret_cast<const long long*>(p.rec + s)[v];
        const long long w0 = __shfl(w, 0, 32), w1 = __shfl(w, 1, 32), w2 = __shfl(w, 2, 32);
        r.f = (int)w0; r.a = (int)(w0 >> 32); r.M = (int)w1; r.o = (int)(w1 >> 32); r.blk = w2;
    }
    const int f = r.f, a = r.a, M = r.M, o = r.o;
    int sl = 1 << 30;
    double hv = 0.0;
    if (live) {                   // (the host reads first, all of them in flight together)
        if (v < M) sl = p.slot_in[a + v];
        if (v < 3) hv = p.base_in[3 * (size_t)f + v];
        else if (v < 6) hv = p.m_in[3 * (size_t)f + v - 3];
        else if (v == 6) hv = p.rho_in[f];
    }
    if (v < M) {
        const double2 q = *reinterpret_cast<const double2*>(p.uv_in + 2 * (size_t)(a + v));
        *reinterpret_cast<double2*>(p.uv + 2 * (size_t)(o + v)) = q;
        p.slot[o + v] = sl;
    }
    int lo = sl;
#pragma unroll
    for (int k = 16; k >= 1; k >>= 1) lo = min(lo, __shfl_xor(lo, k, 32));
    if (v < 2) scol[grp][v] = ~0ull;
    __syncthreads();
    if (v < M && sl - lo < 16) reinterpret_cast<unsigned char*>(scol[grp])[sl - lo] = (unsigned char)v;
    __syncthreads();
    if (!live) return;
    if (v < 3) p.base[3 * (size_t)s + v] = hv;
    else if (v < 6) p.m[3 * (size_t)s + v - 3] = hv;
    else if (v == 6) p.rho[s] = hv;
    else if (v == 7) p.fmin[s] = lo;
    else if (v == 8) {
        FeatInfo fi;
        fi.blk_off = r.blk; fi.M = M; fi.pad = 0;
        *reinterpret_cast<unsigned long long*>(fi.col) = scol[grp][0];
        *reinterpret_cast<unsigned long long*>(fi.col + 8) = scol[grp][1];
        p.info[s] = fi;
    } else if (v == 9) {
        p.view[s] = o; p.perm[s] = f; p.blk[s] = r.blk;
        if (s == p.F - 1) p.view[p.F] = p.sumM;
    }
}

}  // namespace msckf
