// Issue rates on gfx950 that decide how the QR / Cholesky kernels are shaped:
//  - v_fma_f64 throughput of ONE workgroup on one CU with 1, 2, 4 waves per SIMD and 1..8 independent chains,
//  - v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64: back-to-back issue with 1 / 4 accumulators,
//  - v_mfma_f32_16x16x4_f32 for comparison,
//  - DPP row_newbcast on a 64-bit move, v_permlane32_swap.
// cycles via s_memtime (shader clock).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int CH>
__global__ void k(double* out, long long* cyc, int n) {
    const int t = threadIdx.x;
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + t * 1e-9 + i;
    const double y = 0.999999, z = 1e-9;
    d4 acc[4];
    f4 facc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i] = d4{0, 0, 0, 0}; facc[i] = f4{0, 0, 0, 0}; }
    double pa = 1.0 + t * 1e-3, pb = 1.0 - t * 1e-3;
    float fa = 1.0f + t * 1e-3f, fb = 1.0f - t * 1e-3f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < CH; ++c) a[c] = fma(a[c], y, z);
        }
        if (MODE == 1) {
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, pb, acc[c], 0, 0, 0);
        }
        if (MODE == 2) {
#pragma unroll
            for (int c = 0; c < CH; ++c) a[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(pa, pb, a[c], 0, 0, 0);
        }
        if (MODE == 3) {
#pragma unroll
            for (int c = 0; c < CH; ++c) facc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, facc[c], 0, 0, 0);
        }
        if (MODE == 4) {   // mixed: 1 MFMA + CH independent FMAs (co-issue?)
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, pb, acc[0], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CH; ++c) a[c] = fma(a[c], y, z);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + facc[i][0] + facc[i][3];
    out[blockIdx.x * blockDim.x + t] = s;
    if (t == 0) cyc[blockIdx.x] = (long long)(t1 - t0);
}

template <int MODE, int CH> void run(const char* name, int threads) {
    double* out; long long* cyc;
    hipMalloc(&out, 8 * 1024 * 8); hipMalloc(&cyc, 64);
    const int n = 2000;
    hipLaunchKernelGGL((k<MODE, CH>), dim3(1), dim3(threads), 0, 0, out, cyc, n);
    hipLaunchKernelGGL((k<MODE, CH>), dim3(1), dim3(threads), 0, 0, out, cyc, n);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s chains=%d threads=%4d : %7.1f cycles/iter  %6.1f cycles/instr/wave\n", name, CH, threads,
           (double)h / n, (double)h / n / CH);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int th : {64, 256, 512, 1024}) {
        run<0, 1>("v_fma_f64", th);
        run<0, 2>("v_fma_f64", th);
        run<0, 4>("v_fma_f64", th);
        run<0, 8>("v_fma_f64", th);
        run<1, 1>("v_mfma_f64_16x16x4", th);
        run<1, 2>("v_mfma_f64_16x16x4", th);
        run<1, 4>("v_mfma_f64_16x16x4", th);
        run<2, 1>("v_mfma_f64_4x4x4 (4 blocks)", th);
        run<2, 4>("v_mfma_f64_4x4x4 (4 blocks)", th);
        run<3, 1>("v_mfma_f32_16x16x4", th);
        run<3, 4>("v_mfma_f32_16x16x4", th);
        run<4, 2>("1 mfma_f64 + CH fma_f64", th);
        run<4, 4>("1 mfma_f64 + CH fma_f64", th);
        run<4, 8>("1 mfma_f64 + CH fma_f64", th);
    }
    return 0;
}
