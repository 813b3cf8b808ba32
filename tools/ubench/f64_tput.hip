// Chip-wide f64 throughput on gfx950 by waves per SIMD: v_fma_f64 (8 independent chains per wave) and
// v_mfma_f64_16x16x4_f64 (4 accumulators per wave), wall time via HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void k(double* out, int n) {
    const int t = threadIdx.x;
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + t * 1e-9 + i;
    const double y = 0.999999, z = 1e-9;
    d4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    double pa = 1.0 + t * 1e-3, pb = 1.0 - t * 1e-3;
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c) a[c] = fma(a[c], y, z);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, pb, acc[c], 0, 0, 0);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}

template <int MODE> void run(const char* name, int threads, int blocks) {
    double* out;
    (void)hipMalloc(&out, 64);
    const int n = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, out, n);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, out, n);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64;
    const double flops = MODE == 0 ? waves * n * 8.0 * 128.0 : waves * n * 4.0 * 2048.0;
    printf("%-22s threads=%4d blocks=%5d : %8.3f ms  %8.2f TFLOP/s\n", name, threads, blocks, ms, flops / ms * 1e-9);
    (void)hipFree(out);
}

int main() {
    for (int th : {256, 512, 1024}) {
        for (int b : {256, 512, 1024}) {
            run<0>("v_fma_f64 x8", th, b);
            run<1>("v_mfma_f64_16x16x4 x4", th, b);
        }
    }
    return 0;
}
