// Microbenchmark: cost of the per-column building blocks of the fold kernel on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../monocular-visual-inertial-msckf_amd/csrc/wave_ops.h"
using namespace msckf;

template <int MODE, int T>
__global__ __launch_bounds__(T) void k(double* out, int iters, long long* cyc) {
    __shared__ double buf[2][512];
    const int t = threadIdx.x;
    double x = 1.0 + t * 1e-3, acc = 0.0;
    double a[20];
    for (int i = 0; i < 20; ++i) a[i] = x + i;
    long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        double* b = buf[it & 1];
        if (MODE >= 1) { if ((t >> 3) == (it & 31)) b[t & 7] = x; }
        __syncthreads();
        if (MODE >= 1) x += b[it & 7];
        if (MODE >= 2) {              // 20 LDS reads + 20 FMA in 4 chains + 8-lane reduce
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            for (int i = 0; i < 20; i += 4) {
                s0 = fma(b[(t & 7) + 8 * i], a[i], s0); s1 = fma(b[(t & 7) + 8 * (i + 1)], a[i + 1], s1);
                s2 = fma(b[(t & 7) + 8 * (i + 2)], a[i + 2], s2); s3 = fma(b[(t & 7) + 8 * (i + 3)], a[i + 3], s3);
            }
            x += row8_sum((s0 + s1) + (s2 + s3));
        }
        if (MODE >= 3) {              // rsqrt / rcp chain
            const double ss = fma(x, x, 2.0);
            const double nrm = ss * fast_rsqrt(ss);
            x = fast_rcp(nrm * (nrm + fabs(x)));
        }
        if (MODE >= 4) {              // 20 dependent-free FMAs (update)
            for (int i = 0; i < 20; ++i) a[i] = fma(-x, a[i], 1.0);
        }
        acc += x;
    }
    long long t1 = wall_clock64();
    for (int i = 0; i < 20; ++i) acc += a[i];
    out[blockIdx.x * T + t] = acc;
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int T>
void run(const char* name, int blocks) {
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * T * blocks); hipMalloc(&cyc, 8 * blocks);
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE, T>), dim3(blocks), dim3(T), 0, 0, out, iters, cyc);
    hipLaunchKernelGGL((k<MODE, T>), dim3(blocks), dim3(T), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    long long h[4]; hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s T=%4d blocks=%3d : %7.1f ns/iter\n", name, T, blocks, h[0] * 10.0 / iters);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {1, 256}) {
        run<0, 256>("barrier only", blocks);
        run<0, 512>("barrier only", blocks);
        run<0, 1024>("barrier only", blocks);
        run<1, 512>("+ LDS publish/read", blocks);
        run<2, 512>("+ 20 LDS reads, 20 FMA, row8 reduce", blocks);
        run<3, 512>("+ rsqrt/rcp chain", blocks);
        run<4, 512>("+ 20 FMA update", blocks);
        run<4, 256>("+ 20 FMA update", blocks);
    }
    return 0;
}
