// Debug harness: runs k_chol16 on a random SPD matrix with per-step cycle stamps (build: see tools/ubench/README or
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCHOL16_STAMPS -I monocular-visual-inertial-msckf_amd/csrc ...).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "wave_ops.h"
#include "k_gain.h"
using namespace msckf;
#define WV CHOL16_W
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 180;
    std::vector<double> A(n * n), S(n * n);
    srand(1);
    for (auto& x : A) x = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
        double s = 0; for (int k = 0; k < n; ++k) s += A[i * n + k] * A[j * n + k];
        S[i * n + j] = s + (i == j ? 1.0 : 0.0);
    }
    double *dS, *dL, *dU, *dinv, *dwork; int* dst; long long* dstamp;
    hipMalloc(&dS, n * n * 8); hipMalloc(&dL, n * n * 8); hipMalloc(&dU, n * n * 8); hipMalloc(&dinv, n * 8);
    hipMalloc(&dwork, n * (n + 1) / 2 * 8); hipMalloc(&dst, 16); hipMalloc(&dstamp, (12 * WV * 4 + WV * 8) * 8);
    hipMemcpy(dS, S.data(), n * n * 8, hipMemcpyHostToDevice);
    hipMemset(dL, 0, n * n * 8); hipMemset(dstamp, 0, (12 * WV * 4 + WV * 8) * 8);
    CholArgs a{}; a.S = dS; a.lds_ = n; a.L = dL; a.U = dU; a.invd = dinv; a.n = n; a.work = dwork; a.status = dst; a.stamps = dstamp;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_chol16, dim3(1), dim3(64 * WV), 0, 0, a);
    hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(k_chol16, dim3(1), dim3(64 * WV), 0, 0, a);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<double> L(n * n); std::vector<long long> st(12 * WV * 4 + WV * 8);
    hipMemcpy(L.data(), dL, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(st.data(), dstamp, st.size() * 8, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
        double s = 0; for (int k = 0; k <= j; ++k) s += L[i * n + k] * L[j * n + k];
        err = fmax(err, fabs(s - S[i * n + j]));
    }
    printf("n=%d  %.1f us/launch  max |LL^T - S| = %.3g\n", n, ms * 1000 / 20, err);
    const int nb = (n + 15) / 16;
    long long t0 = st[0];
    for (int w = 0; w < WV; ++w) if (st[w * 4] && st[w * 4] < t0) t0 = st[w * 4];
    for (int k = 0; k < nb; ++k) {
        printf("step %2d:", k);
        for (int w = 0; w < WV; ++w) {
            long long* s = &st[(k * WV + w) * 4];
            printf(" | w%d %6lld D%5lld P%5lld B%5lld", w, s[0] - t0, s[1] - s[0], s[2] - s[1], s[3] - s[2]);
        }
        printf("\n");
    }
    printf("followers of step 0 (cycles from t0): start, after chunk 0..3, X published\n");
    for (int w = 0; w < WV; ++w) { long long* q = &st[12 * WV * 4 + w * 8]; if (q[0]) printf("  w%-2d %6lld %6lld %6lld %6lld %6lld %6lld\n", w, q[0] - t0, q[1] - t0, q[2] - t0, q[3] - t0, q[4] - t0, q[5] - t0); }
    return 0;
}
