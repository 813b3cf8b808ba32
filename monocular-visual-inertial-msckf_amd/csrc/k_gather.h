// The tracks of a batch, brought into the pipeline's order ON THE DEVICE.
//   reference MSCKF.py:570-582 : `for feature in features.values()` -- the reference walks its dict in insertion order;
// the engine's kernels want the tracks sorted by (class, first slot, last slot) (k_lsweep's leaves, the per-class launches
// of k_feature).  Until round 3 the host gathered every array into that order before the upload (29 us of the call at 2000
// tracks, 155 us at 10000, all of it in front of K1-K4).  Now the caller's arrays go up AS THEY ARE while the host is still
// validating and sorting; the host sends three short tables behind them (sorted position -> input index, the sorted CSR
// offsets, the offsets of the K4 blocks) and this kernel writes the sorted image the other kernels read: an HBM-bound
// permutation (5.3 MB read + written at 10000 tracks), one 32-lane group per track.
#pragma once
#include <hip/hip_runtime.h>
#include "k_lsweep.h"

namespace msckf {

struct GatherArgs {
    // the caller's arrays, input order
    const int* view_in;           // [F + 1] CSR offsets
    const int* slot_in;           // [sumM]
    const double* uv_in;          // [sumM][2]
    const double* base_in;        // [F][3]
    const double* m_in;           // [F][3]
    const double* rho_in;         // [F]
    // from the host's sort
    const int* perm;              // [F] sorted position -> input index
    const int* view_s;            // [F + 1] CSR offsets of the sorted order
    const long long* blk;         // [F] offset of the track's K4 block
    // the sorted image
    double* uv; double* base; double* m; double* rho;
    int* slot; int* fmin;
    FeatInfo* info;
    int F;
};

constexpr int GATHER_THREADS = 256;

__global__ __launch_bounds__(GATHER_THREADS) void k_gather(GatherArgs p) {
    __shared__ unsigned long long scol[GATHER_THREADS / 32][2];
    const int t = threadIdx.x, grp = t >> 5, v = t & 31;
    const int s = blockIdx.x * (GATHER_THREADS / 32) + grp;
    const bool live = s < p.F;
    int f = 0, a = 0, M = 0, o = 0;
    if (live) { f = p.perm[s]; a = p.view_in[f]; M = p.view_in[f + 1] - a; o = p.view_s[s]; }
    int sl = 1 << 30;
    if (v < M) {
        sl = p.slot_in[a + v];
        p.slot[o + v] = sl;
        const double2 q = *reinterpret_cast<const double2*>(p.uv_in + 2 * (size_t)(a + v));
        *reinterpret_cast<double2*>(p.uv + 2 * (size_t)(o + v)) = q;
    }
    int lo = sl;
#pragma unroll
    for (int k = 16; k >= 1; k >>= 1) lo = min(lo, __shfl_xor(lo, k, 32));
    if (v < 2) scol[grp][v] = ~0ull;
    __syncthreads();
    if (v < M && sl - lo < 16) reinterpret_cast<unsigned char*>(scol[grp])[sl - lo] = (unsigned char)v;
    __syncthreads();
    if (!live) return;
    if (v < 3) p.base[3 * (size_t)s + v] = p.base_in[3 * (size_t)f + v];
    else if (v < 6) p.m[3 * (size_t)s + v - 3] = p.m_in[3 * (size_t)f + v - 3];
    else if (v == 6) p.rho[s] = p.rho_in[f];
    else if (v == 7) p.fmin[s] = lo;
    else if (v == 8) {
        FeatInfo fi;
        fi.blk_off = p.blk[s]; fi.M = M; fi.pad = 0;
        *reinterpret_cast<unsigned long long*>(fi.col) = scol[grp][0];
        *reinterpret_cast<unsigned long long*>(fi.col + 8) = scol[grp][1];
        p.info[s] = fi;
    }
}

}  // namespace msckf
