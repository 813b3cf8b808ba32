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

// Sum over the 8 lanes of each half DPP row; every lane of the group gets the sum.
__device__ __forceinline__ double row8_sum(double x) {
    x += dpp_move<0xB1>(x);    // lane ^ 1
    x += dpp_move<0x4E>(x);    // lane ^ 2
    x += dpp_move<0x141>(x);   // row_half_mirror: pairs the two quads of each 8
    return x;
}

// Sum over all 64 lanes; the result is wave-uniform.
__device__ __forceinline__ double wave_sum(double x) {
    x = row16_sum(x);
    return (readlane_d(x, 0) + readlane_d(x, 16)) + (readlane_d(x, 32) + readlane_d(x, 48));
}

// Sum over RL (8 or 16) adjacent lanes.
template <int RL>
__device__ __forceinline__ double rowN_sum(double x) {
    if constexpr (RL == 16) return row16_sum(x);
    else return row8_sum(x);
}

// 1/sqrt(s) and 1/x from the hardware seeds (v_rsq_f64 / v_rcp_f64, about 2^-26 accurate) plus
// ONE Newton step each (error ~1.5 e0^2, a few 1e-16): a short dependent chain instead of the
// IEEE sqrt / division expansions.  Valid for normal, well-scaled arguments (callers fall back
// to sqrt / division otherwise).  fast_norm() additionally corrects s * rsqrt(s) with the exact
// residual so the pivot magnitude itself is good to the last bit or two.
__device__ __forceinline__ double fast_rsqrt(double s) {
    const double y = __builtin_amdgcn_rsq(s);
    return fma(y, fma(-0.5 * s * y, y, 0.5), y);
}
__device__ __forceinline__ double fast_rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(r, fma(-x, r, 1.0), r);
}
// sqrt(s) given y ~ 1/sqrt(s)
__device__ __forceinline__ double fast_norm(double s, double y) {
    const double n = s * y;
    return fma(fma(-n, n, s), 0.5 * y, n);
}

// d += (lane P of the 16-lane row of src) * w: one DP instruction with the row broadcast folded in (64-bit DPP knows
// row_newbcast only, which is exactly this).  The s_nop covers the VALU-write -> DPP-read hazard, which the compiler's
// hazard recogniser does not see inside inline assembly.
template <int P>
__device__ __forceinline__ void fmac_row_bcast16(double& d, double src, double w) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(src), "v"(w), "i"(P));
}
__device__ __forceinline__ double lane_bcast(double x, int lane) { return readlane_d(x, lane); }

}  // namespace msckf
