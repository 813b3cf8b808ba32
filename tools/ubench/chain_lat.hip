// Dependent-chain latencies of the instructions on k_chol16's pivot chain (one wavefront, gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
template <int MODE>
__global__ void k(double* out, long long* cyc, double seed) {
    double x = seed + threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-9;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N; ++it) {
        if (MODE == 0) { asm volatile("v_rcp_f64 %0, %0" : "+v"(x)); asm volatile("v_rcp_f64 %0, %0" : "+v"(x)); asm volatile("v_rcp_f64 %0, %0" : "+v"(x)); asm volatile("v_rcp_f64 %0, %0" : "+v"(x)); }
        if (MODE == 1) { asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(y)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(y)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(y)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(y)); }
        if (MODE == 2) {  // readlane -> valu -> readlane
            for (int q = 0; q < 4; ++q) { int lo, hi; asm volatile("v_readlane_b32 %0, %2, 5\n\tv_readlane_b32 %1, %3, 5\n\ts_nop 3\n\tv_mov_b32 %2, %0\n\tv_mov_b32 %3, %1" : "=s"(lo), "=s"(hi), "+v"(((int*)&x)[0]), "+v"(((int*)&x)[1])); }
        }
        if (MODE == 3) { for (int q = 0; q < 4; ++q) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y)); }
        if (MODE == 4) { for (int q = 0; q < 4; ++q) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y)); }
        if (MODE == 5) {  // fma -> readlane -> fma (sgpr operand)
            for (int q = 0; q < 4; ++q) {
                int lo = __builtin_amdgcn_readlane(((int*)&x)[0], 5), hi = __builtin_amdgcn_readlane(((int*)&x)[1], 5);
                double sv = __hiloint2double(hi, lo);
                asm volatile("v_fma_f64 %0, %1, %2, %2" : "=v"(x) : "s"(sv), "v"(y));
            }
        }
        if (MODE == 6) { for (int q = 0; q < 4; ++q) { int lo = ((int*)&x)[0], hi = ((int*)&x)[1]; int a = (threadIdx.x & 15) * 4; asm volatile("ds_bpermute_b32 %0, %2, %0 offset:64\n\tds_bpermute_b32 %1, %2, %1 offset:64\n\ts_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi) : "v"(a)); ((int*)&x)[0] = lo; ((int*)&x)[1] = hi; } }
        if (MODE == 7) { for (int q = 0; q < 4; ++q) { int lo = ((int*)&x)[0], hi = ((int*)&x)[1]; asm volatile("v_cndmask_b32 %0, 0, %0, vcc\n\tv_cndmask_b32 %1, 0, %1, vcc" : "+v"(lo), "+v"(hi)); ((int*)&x)[0] = lo; ((int*)&x)[1] = hi; } }
    }
    if (MODE == 8 || MODE == 9) {   // the pivot body of k_chol16 (MODE 9: without the LDS column write / publish)
        __shared__ double sL[256]; __shared__ double sRp[16];
        const int lane = threadIdx.x, cc = lane & 15, g = lane >> 4, cc4 = cc * 4;
        double d[4] = {x, x + 1, x + 2, x + 3}; double pivs = 0;
        t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int it = 0; it < N; ++it) {
            int ccl = cc; asm volatile("" : "+v"(ccl));
            int lo = __double2loint(d[1]), hi = __double2hiint(d[1]);
            asm volatile("ds_bpermute_b32 %0, %2, %0 offset:%3\n\tds_bpermute_b32 %1, %2, %1 offset:%3" : "+v"(lo), "+v"(hi) : "v"(cc4), "i"(64) : "memory");
            double lc = __hiloint2double(hi, lo);
            int plo = __builtin_amdgcn_readlane(__double2loint(d[1]), 21), phi = __builtin_amdgcn_readlane(__double2hiint(d[1]), 21);
            double piv = __hiloint2double(phi, plo);
            if (MODE == 8 && ccl == 5) { pivs = piv; for (int i = 0; i < 4; ++i) sL[5 * 16 + g + 4 * i] = d[i]; }
            double r = __builtin_amdgcn_rcp(piv); double rp = fma(r, fma(-piv, r, 1.0), r);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lc), "+v"(rp)::"memory");
            const double lcm = (ccl > 5) ? -lc : 0.0; const double w = lcm * rp * 1e-3;
            asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "v"(w));
            if (MODE == 8 && lane == 0) sRp[5] = rp;
        }
        x = d[0] + d[1] + d[2] + d[3] + pivs + sL[lane] + sRp[lane & 15];
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 64 * 8 * 4); hipMalloc(&cyc, 64 * 8);
    const char* names[] = {"v_rcp_f64", "v_fma_f64", "readlane x2 + nop3 + v_mov x2", "v_fmac_f64_dpp (+s_nop 1)", "v_mul_f64", "readlane x2 + nop3 + v_fma_f64", "ds_bpermute x2 + wait", "v_cndmask x2"};
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5);
        hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5);
        hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5);
        hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); hipLaunchKernelGGL(k<7>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5);
        hipLaunchKernelGGL(k<8>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); hipLaunchKernelGGL(k<9>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5);
        hipDeviceSynchronize();
    }
    long long h[10]; hipMemcpy(h, cyc, 80, hipMemcpyDeviceToHost);
    for (int m = 0; m < 8; ++m) printf("%-34s %.1f cycles per dependent step\n", names[m], h[m] / (4.0 * N));
    printf("pivot body: %.1f cycles, without LDS column write/publish: %.1f cycles\n", h[8] / (double)N, h[9] / (double)N);
    return 0;
}
