// Standalone check of k_gain_stream (csrc/k_gstream.h): the sequential block update against a dense CPU evaluation of
// the reference's formulas (MSCKF.py:604-614), alone and beside a producer kernel that hands out the rows of T the way
// the root sweep's flusher does (write-through stores, progress word), at a given pace.
//   hipcc -O3 --offload-arch=gfx950 -I monocular-visual-inertial-msckf_amd/csrc -o build/gstream_test tools/ubench/gstream_test.hip
//   build/gstream_test [N=30] [band=60] [us_per_row=0.6] [reps=20] [nb2=0]
// nb2 > 0: ONLY a dense second source of 16 nb2 rows (the remainder rows of split long tracks), standalone, against the same formulas.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "k_gstream.h"
#include "k_gdense.h"

using namespace msckf;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// one wavefront: rows of src -> dst (write-through), `gap` wall-clock ticks (10 ns) apart, progress after each drained row
__global__ void k_producer(const double* src, double* dst, int dc, int ldt, int band, unsigned long long* progress, unsigned epoch, int gap) {
    const int lane = threadIdx.x;
    long long t_next = wall_clock64() + 2000;          // 20 us head start of the consumer
    for (int c = 0; c < dc; ++c) {
        while (wall_clock64() < t_next) __builtin_amdgcn_s_sleep(1);
        t_next += gap;
        for (int col = c + lane; col < dc && col < c + band; col += 64) gs_std(dst + (size_t)c * ldt + col, src[(size_t)c * ldt + col]);
        if (lane == 0) gs_std(dst + (size_t)c * ldt + dc, src[(size_t)c * ldt + dc]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) gs_st(progress, ((unsigned long long)epoch << 32) | (unsigned)(c + 1));
    }
}

#ifdef GS_TEST_WV
// the update's strips as k_root_gain runs them: GS_TEST_WV wavefronts per workgroup, GS_TEST_TPW tiles per wavefront
__global__ __launch_bounds__(64 * GS_TEST_WV) void k_gain_test(GStreamArgs p) { gain_stream_body<GS_TEST_WV, GS_TEST_WV - 1, GS_TEST_TPW>(p, blockIdx.x); }
#endif

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 30;
    const int band = argc > 2 ? atoi(argv[2]) : 60;
    const double us_row = argc > 3 ? atof(argv[3]) : 0.6;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    const int nb2 = argc > 5 ? atoi(argv[5]) : 0;
    const int pairs = argc > 6 ? atoi(argv[6]) : 0;         // 1: k_gain_dense (two row blocks per exchange) on the dense source
    const int dc = 6 * N, d = 15 + dc, ldt = dc + 1;
    const int nb = (dc + 15) / 16, ns = nb + 1;
    const int ncb = gstream_ncb(dc, band);
    const double sigma2 = 0.04;
    std::mt19937_64 rng(1234);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<double> T((size_t)dc * ldt, 0.0), P((size_t)d * d), A((size_t)d * d);
    for (int c = 0; c < dc; ++c) {
        for (int col = c; col < dc && col < c + band; ++col) T[(size_t)c * ldt + col] = 3.0 * nd(rng);
        T[(size_t)c * ldt + dc] = nd(rng);
    }
    for (auto& x : A) x = nd(rng);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            double s = 0.0;
            for (int k = 0; k < d; ++k) s += A[(size_t)i * d + k] * A[(size_t)j * d + k];
            P[(size_t)i * d + j] = 1e-4 * s / d + (i == j ? 1e-3 : 0.0);
        }
    // ---- CPU: S = T Pcc T^T + s2 I, K = P[:,15:] T^T S^-1, dx = K r, P+ = P - K (P[:,15:] T^T)^T (exact-arithmetic Joseph) ----
    std::vector<double> Y((size_t)d * dc), S((size_t)dc * dc), L((size_t)dc * dc, 0.0), X((size_t)d * dc), dx(d), Pn((size_t)d * d);
    for (int i = 0; i < d; ++i)
        for (int c = 0; c < dc; ++c) {
            double s = 0.0;
            for (int k = c; k < dc && k < c + band; ++k) s += P[(size_t)i * d + 15 + k] * T[(size_t)c * ldt + k];
            Y[(size_t)i * dc + c] = s;
        }
    for (int a = 0; a < dc; ++a)
        for (int b = 0; b < dc; ++b) {
            double s = (a == b) ? sigma2 : 0.0;
            for (int k = a; k < dc && k < a + band; ++k) s += T[(size_t)a * ldt + k] * Y[(size_t)(15 + k) * dc + b];
            S[(size_t)a * dc + b] = s;
        }
    for (int j = 0; j < dc; ++j) {
        double s = S[(size_t)j * dc + j];
        for (int k = 0; k < j; ++k) s -= L[(size_t)j * dc + k] * L[(size_t)j * dc + k];
        L[(size_t)j * dc + j] = std::sqrt(s);
        for (int i = j + 1; i < dc; ++i) {
            double v = S[(size_t)i * dc + j];
            for (int k = 0; k < j; ++k) v -= L[(size_t)i * dc + k] * L[(size_t)j * dc + k];
            L[(size_t)i * dc + j] = v / L[(size_t)j * dc + j];
        }
    }
    std::vector<double> w(dc);
    for (int j = 0; j < dc; ++j) {
        double v = T[(size_t)j * ldt + dc];
        for (int k = 0; k < j; ++k) v -= L[(size_t)j * dc + k] * w[k];
        w[j] = v / L[(size_t)j * dc + j];
    }
    for (int i = 0; i < d; ++i) {
        for (int j = 0; j < dc; ++j) {
            double v = Y[(size_t)i * dc + j];
            for (int k = 0; k < j; ++k) v -= X[(size_t)i * dc + k] * L[(size_t)j * dc + k];
            X[(size_t)i * dc + j] = v / L[(size_t)j * dc + j];
        }
        double s = 0.0;
        for (int j = 0; j < dc; ++j) s += X[(size_t)i * dc + j] * w[j];
        dx[i] = s;
    }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            double s = P[(size_t)i * d + j];
            for (int k = 0; k < dc; ++k) s -= X[(size_t)i * dc + k] * X[(size_t)j * dc + k];
            Pn[(size_t)i * d + j] = s;
        }
    // ---- device ----
    double *dP, *dT, *dTsrc, *dEx, *dDx, *dPout;
    unsigned long long *dFlag, *dProg;
    int* dStatus;
    CK(hipMalloc(&dP, P.size() * 8)); CK(hipMalloc(&dT, T.size() * 8)); CK(hipMalloc(&dTsrc, T.size() * 8));
    CK(hipMalloc(&dEx, (size_t)nb * ns * 256 * 16)); CK(hipMemset(dEx, 0, (size_t)nb * ns * 256 * 16)); CK(hipMalloc(&dFlag, (size_t)nb * ns * 8 + 64)); CK(hipMalloc(&dProg, 64));
    CK(hipMalloc(&dDx, d * 8)); CK(hipMalloc(&dPout, P.size() * 8)); CK(hipMalloc(&dStatus, 64));
    CK(hipMemcpy(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dTsrc, T.data(), T.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dFlag, 0, (size_t)nb * ns * 8 + 64)); CK(hipMemset(dProg, 0, 64));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const size_t lds = std::max<size_t>(gstream_lds_doubles(ns, ncb) * 8, 100 * 1024);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gain_stream<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gain_stream<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
#ifdef GS_TEST_WV
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gain_test), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
#endif
    GStreamArgs a{};
    a.P = dP; a.ldp = d; a.T = dT; a.ldt = ldt; a.ex = dEx; a.exflag = dFlag; a.dx = dDx; a.Pout = dPout; a.ldo = d;
    a.status = dStatus; a.sigma2 = sigma2; a.d = d; a.dc = dc; a.nb = nb; a.ns = ns; a.ncb = ncb; a.nb1 = nb;
    long long* dStamps = nullptr;
    CK(hipMalloc(&dStamps, 80 * 8 * 8)); CK(hipMemset(dStamps, 0, 80 * 8 * 8));
    a.stamps = dStamps;
    unsigned epoch = 0;
    auto launch = [&](hipStream_t st) {
#ifdef GS_TEST_WV
        hipLaunchKernelGGL(k_gain_test, dim3(ns), dim3(64 * GS_TEST_WV), lds, st, a);
        return;
#endif

        if (ns <= 16) hipLaunchKernelGGL(k_gain_stream<1>, dim3(ns), dim3(64 * GS_WAVES), lds, st, a);
        else hipLaunchKernelGGL(k_gain_stream<2>, dim3(ns), dim3(64 * GS_WAVES), lds, st, a);
    };
    auto check = [&](const char* what) -> int {
        std::vector<double> hdx(d), hP((size_t)d * d);
        int st = -1;
        if (hipMemcpy(hdx.data(), dDx, d * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        if (hipMemcpy(hP.data(), dPout, hP.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        if (hipMemcpy(&st, dStatus, 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        double n1 = 0, e1 = 0, n2 = 0, e2 = 0, asym = 0;
        for (int i = 0; i < d; ++i) { n1 += dx[i] * dx[i]; e1 += (hdx[i] - dx[i]) * (hdx[i] - dx[i]); }
        for (size_t i = 0; i < hP.size(); ++i) { n2 += Pn[i] * Pn[i]; e2 += (hP[i] - Pn[i]) * (hP[i] - Pn[i]); }
        for (int i = 0; i < d; ++i) for (int j = 0; j < i; ++j) asym = std::max(asym, std::fabs(hP[(size_t)i * d + j] - hP[(size_t)j * d + i]));
        std::printf("%s: status %d  rel err dx %.3e  P+ %.3e  max |P+ - P+^T| %.3e\n", what, st, std::sqrt(e1 / n1), std::sqrt(e2 / n2), asym);
        return (st == 0 && std::sqrt(e1 / n1) < 1e-9 && std::sqrt(e2 / n2) < 1e-9) ? 0 : 2;
    };

    if (nb2 > 0) {
        // dense rows R: m x (dc + 1); Y = P[:, 15:] R^T; S = R Y[15:] + s2 I = L L^T; X = Y L^-T; dx = X L^-1 r; P+ = P - X X^T
        const int m = 16 * nb2;
        std::vector<double> R((size_t)m * ldt), Y2((size_t)d * m), S2((size_t)m * m), L2((size_t)m * m, 0.0), X2((size_t)d * m), dx2(d), Pn2((size_t)d * d), w2(m);
        for (auto& x : R) x = nd(rng);
        for (int i = 0; i < d; ++i)
            for (int c = 0; c < m; ++c) {
                double s = 0.0;
                for (int k = 0; k < dc; ++k) s += P[(size_t)i * d + 15 + k] * R[(size_t)c * ldt + k];
                Y2[(size_t)i * m + c] = s;
            }
        for (int a2 = 0; a2 < m; ++a2)
            for (int b = 0; b < m; ++b) {
                double s = (a2 == b) ? sigma2 : 0.0;
                for (int k = 0; k < dc; ++k) s += R[(size_t)a2 * ldt + k] * Y2[(size_t)(15 + k) * m + b];
                S2[(size_t)a2 * m + b] = s;
            }
        for (int j = 0; j < m; ++j) {
            double s = S2[(size_t)j * m + j];
            for (int k = 0; k < j; ++k) s -= L2[(size_t)j * m + k] * L2[(size_t)j * m + k];
            L2[(size_t)j * m + j] = std::sqrt(s);
            for (int i = j + 1; i < m; ++i) {
                double v = S2[(size_t)i * m + j];
                for (int k = 0; k < j; ++k) v -= L2[(size_t)i * m + k] * L2[(size_t)j * m + k];
                L2[(size_t)i * m + j] = v / L2[(size_t)j * m + j];
            }
        }
        for (int j = 0; j < m; ++j) {
            double v = R[(size_t)j * ldt + dc];
            for (int k = 0; k < j; ++k) v -= L2[(size_t)j * m + k] * w2[k];
            w2[j] = v / L2[(size_t)j * m + j];
        }
        for (int i = 0; i < d; ++i) {
            for (int j = 0; j < m; ++j) {
                double v = Y2[(size_t)i * m + j];
                for (int k = 0; k < j; ++k) v -= X2[(size_t)i * m + k] * L2[(size_t)j * m + k];
                X2[(size_t)i * m + j] = v / L2[(size_t)j * m + j];
            }
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += X2[(size_t)i * m + j] * w2[j];
            dx2[i] = s;
        }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                double s = P[(size_t)i * d + j];
                for (int k = 0; k < m; ++k) s -= X2[(size_t)i * m + k] * X2[(size_t)j * m + k];
                Pn2[(size_t)i * d + j] = s;
            }
        dx = dx2; Pn = Pn2;
        double* dR; double* dEx2; unsigned long long* dFlag2; long long* dSt2;
        CK(hipMalloc(&dR, R.size() * 8)); CK(hipMemcpy(dR, R.data(), R.size() * 8, hipMemcpyHostToDevice));
        CK(hipMalloc(&dEx2, (size_t)(nb2 + nb) * ns * 256 * 16)); CK(hipMemset(dEx2, 0, (size_t)(nb2 + nb) * ns * 256 * 16)); CK(hipMalloc(&dFlag2, (size_t)(nb2 + nb) * ns * 8 + 64));
        CK(hipMemset(dFlag2, 0, (size_t)(nb2 + nb) * ns * 8 + 64));
        CK(hipMalloc(&dSt2, (size_t)(nb2 + 80) * 8 * 8)); CK(hipMemset(dSt2, 0, (size_t)(nb2 + 80) * 8 * 8));
        a.ex = dEx2; a.exflag = dFlag2; a.T = nullptr; a.nb1 = 0; a.T2 = dR; a.ldt2 = ldt; a.nb2 = nb2; a.progress = nullptr; a.stamps = dSt2;
        const size_t lds2 = pairs ? gdense_lds_doubles(ns, nb) * 8 : gstream_lds_doubles(ns, nb) * 8;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gain_dense), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        if (pairs && (ns > 14 || lds2 > 160 * 1024 - 1024)) { std::printf("k_gain_dense: window too large\n"); return 1; }
        auto launch2 = [&](hipStream_t st) {
            if (pairs) hipLaunchKernelGGL(k_gain_dense, dim3(ns), dim3(64 * GS_WAVES), lds2, st, a);
            else if (ns <= 16) hipLaunchKernelGGL(k_gain_stream<1>, dim3(ns), dim3(64 * GS_WAVES), lds2, st, a);
            else hipLaunchKernelGGL(k_gain_stream<2>, dim3(ns), dim3(64 * GS_WAVES), lds2, st, a);
        };
        int rc2 = 0;
        a.epoch = ++epoch; launch2(sb); CK(hipStreamSynchronize(sb));
        rc2 |= check("dense second source");
        float best = 1e9f, sum = 0.f;
        for (int it = 0; it < reps; ++it) {
            a.epoch = ++epoch;
            CK(hipEventRecord(e0, sb)); launch2(sb); CK(hipEventRecord(e1, sb)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); sum += ms;
        }
        std::printf("dense second source, %d row blocks: best %.1f us, mean %.1f us = %.2f us per block (N = %d, %d strips)\n", nb2, best * 1e3, sum / reps * 1e3,
                    best * 1e3 / nb2, N, ns);
        rc2 |= check("dense second source (last rep)");
#ifdef GS_STAMPS
        {
            std::vector<long long> st((size_t)(nb2 + 80) * 8);
            CK(hipMemcpy(st.data(), dSt2, st.size() * 8, hipMemcpyDeviceToHost));
            std::printf("workgroup 0, 10 ns ticks per row block: top -> partials | -> published | -> tiles fetched | barrier | eliminated | wave 0 followed | LDS write + barrier | to next block's top\n");
            for (int I = 0; I < nb2 && I < 60; ++I) {
                if (I == 5) continue;
                const long long* q = &st[I * 8];
                std::printf("  block %2d: %5lld | %5lld | %5lld | %4lld | %5lld | %5lld | %5lld | %5lld\n", I, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4],
                            q[7] - q[5], q[6] - q[7], I + 1 < nb2 ? st[(I + 1) * 8] - q[6] : 0LL);
            }
        }
#endif
        std::printf(rc2 == 0 ? "OK\n" : "FAILED\n");
        return rc2;
    }
    int rc = 0;
    // (1) alone: T complete, no progress word
    CK(hipMemcpy(dT, T.data(), T.size() * 8, hipMemcpyHostToDevice));
    a.progress = nullptr; a.epoch = ++epoch;
    launch(sb);
    CK(hipStreamSynchronize(sb));
    rc |= check("standalone");
    {
        float best = 1e9f, sum = 0.f;
        for (int it = 0; it < reps; ++it) {
            a.epoch = ++epoch;
            CK(hipEventRecord(e0, sb)); launch(sb); CK(hipEventRecord(e1, sb)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); sum += ms;
        }
        std::printf("standalone kernel: best %.1f us, mean %.1f us (N = %d, %d row blocks, %d strips, %d column blocks per row block)\n",
                    best * 1e3, sum / reps * 1e3, N, nb, ns, ncb);
        rc |= check("standalone (last rep)");
#ifdef GS_STAMPS
        std::vector<long long> st(80 * 8);
        CK(hipMemcpy(st.data(), dStamps, st.size() * 8, hipMemcpyDeviceToHost));
        std::printf("workgroup 0, 10 ns ticks per row block: T seen -> partials | -> published | -> tiles fetched | barrier | eliminated | wave 0 followed | LDS write + barrier | to next block's T seen\n");
        for (int I = 0; I < nb; ++I) {
            const long long* q = &st[I * 8];
            std::printf("  block %2d: %5lld | %5lld | %5lld | %4lld | %5lld | %5lld | %5lld | %5lld\n", I, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4],
                        q[7] - q[5], q[6] - q[7], I + 1 < nb ? st[(I + 1) * 8] - q[6] : 0LL);
        }
        for (int k = 0; k < 2; ++k) { std::printf("timeout record %d:", k); for (int q = 0; q < 8; ++q) std::printf(" %llx", (unsigned long long)st[70 * 8 + 8 * k + q]); std::printf("\n"); }
        std::printf("block 5, arrival at the last barrier by wavefront, ticks after the elimination: ");
        for (int w = 0; w < 16; ++w) std::printf("%lld ", st[64 * 8 + w] - st[5 * 8 + 5]);
        std::printf("\n");
#endif
    }
    // (2) beside the producer
    const int gap = (int)std::lround(us_row * 100.0);
    float tail_best = 1e9f, tail_sum = 0.f;
    for (int it = 0; it < reps; ++it) {
        CK(hipMemsetAsync(dT, 0, T.size() * 8, sa));
        CK(hipStreamSynchronize(sa));
        a.progress = dProg; a.epoch = ++epoch;
        hipLaunchKernelGGL(k_producer, dim3(1), dim3(64), 0, sa, dTsrc, dT, dc, ldt, band, dProg, epoch, gap);
        CK(hipEventRecord(e1, sa));
        launch(sb);
        CK(hipEventRecord(e2, sb));
        CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
        float ms = 0; CK(hipEventElapsedTime(&ms, e1, e2));
        tail_best = std::min(tail_best, ms); tail_sum += ms;
        if (it == 0 || it == reps - 1) rc |= check("beside the producer");
    }
    std::printf("producer at %.2f us per row (%.0f us): consumer ends %.1f us (best), %.1f us (mean) after the producer\n",
                us_row, us_row * dc, tail_best * 1e3, tail_sum / reps * 1e3);
    std::printf(rc == 0 ? "OK\n" : "FAILED\n");
    return rc;
}
