#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <algorithm>
#include "../../monocular-visual-inertial-msckf_amd/csrc/k_gain.h"
using namespace msckf;
template <int TT = 512>
float run_tile(CholArgs a) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_chol_tile<TT>), dim3(1), dim3(TT), 0, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_chol_tile<TT>), dim3(1), dim3(TT), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 100.0f;
}
double check(const std::vector<double>& S, double* dL, int n) {
    std::vector<double> L(n * n);
    hipMemcpy(L.data(), dL, n * n * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
        double s = 0; for (int k = 0; k <= j; ++k) s += L[i * n + k] * L[j * n + k];
        worst = std::max(worst, std::abs(s - S[i * n + j]));
    }
    return worst;
}
int main() {
    const int n = 186;
    std::vector<double> S(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) S[i * n + j] = (i == j) ? n + 1.0 : 1.0 / (1 + std::abs(i - j));
    CholArgs a{};
    double *dS, *dL, *dU, *dI; int* st;
    hipMalloc(&dS, n * n * 8); hipMalloc(&dL, n * n * 8); hipMalloc(&dU, n * n * 8); hipMalloc(&dI, n * 8); hipMalloc(&st, 64);
    hipMemcpy(dS, S.data(), n * n * 8, hipMemcpyHostToDevice); hipMemset(st, 0, 64);
    a.S = dS; a.lds_ = n; a.L = dL; a.U = dU; a.invd = dI; a.n = n; a.status = st;
    printf("tile        : %.1f us\n", run_tile(a));
    hipDeviceSynchronize();
    printf("tile max |L L^T - S| = %.2e\n", check(S, dL, n));
    for (int nn : {60, 186, 6}) {
        a.n = nn; a.lds_ = n; hipMemset(dL, 0, n * n * 8);
        float us = run_tile(a); hipDeviceSynchronize();
        std::vector<double> L(nn * nn); hipMemcpy(L.data(), dL, nn * nn * 8, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int i = 0; i < nn; ++i) for (int j = 0; j <= i; ++j) { double s2 = 0; for (int k = 0; k <= j; ++k) s2 += L[i * nn + k] * L[j * nn + k]; worst = std::max(worst, std::abs(s2 - ((i < n && j < n) ? S[i * n + j] : 0))); }
        printf("tile n=%d : %.1f us, err %.2e\n", nn, us, worst);
    }
    return 0;
}
