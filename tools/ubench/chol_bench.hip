#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../../monocular-visual-inertial-msckf_amd/csrc/k_gain.h"
using namespace msckf;
template <int DBG> float run(CholArgs a, size_t lds) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_blk<512, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_chol_blk<512, DBG>), dim3(1), dim3(512), lds, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_chol_blk<512, DBG>), dim3(1), dim3(512), lds, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 100.0f;
}
int main() {
    const int n = 180;
    std::vector<double> S(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) S[i * n + j] = (i == j) ? n + 1.0 : 1.0 / (1 + std::abs(i - j));
    CholArgs a{};
    double *dS, *dL, *dU, *dI; int* st;
    hipMalloc(&dS, n * n * 8); hipMalloc(&dL, n * n * 8); hipMalloc(&dU, n * n * 8); hipMalloc(&dI, n * 8); hipMalloc(&st, 64);
    hipMemcpy(dS, S.data(), n * n * 8, hipMemcpyHostToDevice); hipMemset(st, 0, 64);
    a.S = dS; a.lds_ = n; a.L = dL; a.U = dU; a.invd = dI; a.n = n; a.status = st;
    const size_t lds = (size_t)n * (n + 1) / 2 * 8;
    printf("full        : %.1f us\n", run<0>(a, lds));
    printf("no trailing : %.1f us\n", run<1>(a, lds));
    printf("diag only   : %.1f us\n", run<2>(a, lds));
    return 0;
}
