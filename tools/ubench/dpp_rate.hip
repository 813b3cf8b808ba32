// Issue rate of v_fmac_f64 with a DPP row_newbcast operand against the plain v_fma_f64, and of the lane swaps, with
// 1 / 2 / 4 wavefronts per SIMD (one workgroup on one CU).  cycles via s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int CH>
__global__ void k(double* out, long long* cyc, int n) {
    const int t = threadIdx.x;
    double a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = 1.0 + t * 1e-9 + i; b[i] = 1e-3 * i + t * 1e-6; }
    const double y = 0.999999, z = 1e-9;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < CH; ++c) a[c] = fma(a[c], y, z);
        }
        if (MODE == 1) {   // accumulate: d += bcast(src) * w, src not written in the loop
#pragma unroll
            for (int c = 0; c < CH; ++c) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(b[c]), "v"(z));
        }
        if (MODE == 2) {   // plain v_fmac_f64 (VOP2) through asm, same shape
#pragma unroll
            for (int c = 0; c < CH; ++c) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(z));
        }
        if (MODE == 3) {   // lane swaps (32-bit)
            unsigned* u = reinterpret_cast<unsigned*>(a);
#pragma unroll
            for (int c = 0; c < CH; ++c) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[2 * c]), "+v"(u[2 * c + 1]));
        }
        if (MODE == 4) {   // v_mov_b64 dpp row_newbcast
#pragma unroll
            for (int c = 0; c < CH; ++c) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(a[c]) : "v"(b[c]));
        }
        if (MODE == 5) {   // v_add_f64
#pragma unroll
            for (int c = 0; c < CH; ++c) a[c] = a[c] + z;
        }
        if (MODE == 6) {   // v_cndmask_b32 pairs
            unsigned* u = reinterpret_cast<unsigned*>(a);
#pragma unroll
            for (int c = 0; c < CH; ++c) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[2 * c]) : "v"(u[2 * c + 1]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + t] = s;
    if (t == 0) cyc[blockIdx.x] = (long long)(t1 - t0);
}
template <int MODE, int CH> void run(const char* name, int threads) {
    double* out; long long* cyc;
    (void)hipMalloc(&out, 8 * 1024 * 8); (void)hipMalloc(&cyc, 64);
    const int n = 2000;
    hipLaunchKernelGGL((k<MODE, CH>), dim3(1), dim3(threads), 0, 0, out, cyc, n);
    hipLaunchKernelGGL((k<MODE, CH>), dim3(1), dim3(threads), 0, 0, out, cyc, n);
    (void)hipDeviceSynchronize();
    long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-36s chains=%d waves/SIMD=%d : %6.1f cycles/instr/wave  %6.2f cycles/instr/SIMD\n", name, CH, threads / 256,
           (double)h / n / CH, (double)h / n / CH / (threads / 256));
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int th : {256, 512, 1024}) {
        run<0, 8>("v_fma_f64", th);
        run<2, 8>("v_fmac_f64 (VOP2)", th);
        run<1, 8>("v_fmac_f64_dpp row_newbcast", th);
        run<1, 4>("v_fmac_f64_dpp row_newbcast", th);
        run<4, 8>("v_mov_b64_dpp row_newbcast", th);
        run<5, 8>("v_add_f64", th);
        run<3, 4>("v_permlane32_swap_b32", th);
        run<6, 4>("v_cndmask_b32", th);
    }
    return 0;
}
