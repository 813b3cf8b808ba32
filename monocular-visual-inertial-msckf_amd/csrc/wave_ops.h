// Wavefront (64-lane) primitives for gfx950.  DPP-based all-reduce for doubles:
// the 16-lane row stages use quad_perm / row_half_mirror / row_mirror moves
// (no LDS round trip), the 4 rows are combined through v_readlane.
#pragma once
#include <hip/hip_runtime.h>

namespace msckf {

template <int CTRL>
__device__ __forceinline__ double dpp_move(double x) {
    int lo = __double2loint(x);
    int hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_d(double x, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// Sum over the 16 lanes of each DPP row; every lane of the row gets the row sum.
__device__ __forceinline__ double row16_sum(double x) {
    x += dpp_move<0xB1>(x);    // quad_perm [1,0,3,2]  : lane ^ 1
    x += dpp_move<0x4E>(x);    // quad_perm [2,3,0,1]  : lane ^ 2
    x += dpp_move<0x141>(x);   // row_half_mirror      : pairs the two quads of each 8
    x += dpp_move<0x140>(x);   // row_mirror           : pairs the two halves of the row
    return x;
}

// Sum over all 64 lanes; the result is wave-uniform.
__device__ __forceinline__ double wave_sum(double x) {
    x = row16_sum(x);
    return (readlane_d(x, 0) + readlane_d(x, 16)) + (readlane_d(x, 32) + readlane_d(x, 48));
}

__device__ __forceinline__ double lane_bcast(double x, int lane) { return readlane_d(x, lane); }

}  // namespace msckf
