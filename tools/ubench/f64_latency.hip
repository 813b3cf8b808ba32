// Dependent-chain latencies of the f64 building blocks on gfx950 (cycles via s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../monocular-visual-inertial-msckf_amd/csrc/wave_ops.h"
using namespace msckf;

template <int MODE>
__global__ void k(double* out, long long* cyc, int n) {
    __shared__ double lds[1024];
    const int t = threadIdx.x;
    lds[t & 1023] = 1.0 + t * 1e-6;
    __syncthreads();
    double x = 1.0 + t * 1e-9, y = 0.999999, z = 1e-9;
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) { x = fma(x, y, z); }                                   // dependent FMA
        if (MODE == 1) { a0 = fma(a0, y, z); a1 = fma(a1, y, z); a2 = fma(a2, y, z); a3 = fma(a3, y, z); }  // 4 independent
        if (MODE == 2) { x += dpp_move<0xB1>(x); }                             // one DPP reduce stage
        if (MODE == 3) { x = row16_sum(x) * 0.0625; }                          // 4-stage reduce
        if (MODE == 4) { x = lds[((int)x + i) & 1023] + 1e-9; }                // dependent LDS read
        if (MODE == 5) { x = fast_rsqrt(x + 1.0); }                            // rsq + 1 Newton
        if (MODE == 6) { x = fast_rcp(x + 1.0); }
        if (MODE == 7) { x = __builtin_amdgcn_rsq(x + 1.0); }
        if (MODE == 8) { x = sqrt(x + 1.0); }
        if (MODE == 9) { x = 1.0 / (x + 1.0); }
        if (MODE == 10) { x = wave_sum(x) * (1.0 / 64); }
        if (MODE == 11) { x = readlane_d(x, i & 63) + 1e-9; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + t] = x + a0 + a1 + a2 + a3;
    if (t == 0) cyc[blockIdx.x] = (long long)(t1 - t0);
}

template <int MODE> void run(const char* name, int threads) {
    double* out; long long* cyc;
    hipMalloc(&out, 8 * 1024 * 8); hipMalloc(&cyc, 64);
    const int n = 2000;
    hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(threads), 0, 0, out, cyc, n);
    hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(threads), 0, 0, out, cyc, n);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s threads=%4d : %7.1f cycles/iter\n", name, threads, (double)h / n);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int th : {64, 256, 512, 1024}) {
        run<0>("dependent v_fma_f64", th);
        run<1>("4 independent v_fma_f64", th);
        run<2>("1 DPP stage (2 mov_dpp + add_f64)", th);
        run<3>("row16_sum (4 stages)", th);
        run<10>("wave_sum (64 lanes)", th);
        run<11>("readlane_d + add", th);
        run<4>("dependent ds_read_b64", th);
        run<7>("v_rsq_f64 + add", th);
        run<5>("fast_rsqrt (1 Newton) + add", th);
        run<6>("fast_rcp (1 Newton) + add", th);
        run<8>("sqrt() + add", th);
        run<9>("1.0 / x", th);
    }
    return 0;
}
