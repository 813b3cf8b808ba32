// Which part of k_chol16's pivot body costs what: the body in a warm loop with parts knocked out (one wavefront).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
// KO bits: 1 = no ds_bpermute, 2 = no readlane (pivot from a register), 4 = no rcp/Newton (constant), 8 = no column/rp publish,
//          16 = no lcm select, 32 = only one fmac_dpp
template <int KO>
__global__ void k(double* out, long long* cyc, double seed, int slot) {
    __shared__ double sL[256 + 64 + 256]; __shared__ double sRp[16 + 64];
    const int lane = threadIdx.x, cc = lane & 15, g = lane >> 4, cc4 = cc * 4;
    double d[4] = {seed + lane, seed + 1, seed + 2, seed + 3}; double pivs = 0;
    const unsigned sl_g = (unsigned)(size_t)&sL[g], dump = (unsigned)(size_t)&sL[256 + lane], rpo = (unsigned)(size_t)&sRp[0], rdump = (unsigned)(size_t)&sRp[16 + lane];
    int rlo = 0, rhi = 0x3ff00000; double piv = seed;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N; ++it) {
        int ccl = cc; asm volatile("" : "+v"(ccl));
        double r = (KO & 4) ? 0.5 : __builtin_amdgcn_rcp(piv);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rlo), "+v"(rhi), "+v"(r)::"memory");
        const double lc = __hiloint2double(rhi, rlo);
        const double lcm = (KO & 16) ? -lc : ((ccl > 5) ? -lc : 0.0);
        if (!(KO & 8)) {
            pivs = (ccl == 5) ? piv : pivs;
            const unsigned ca = (ccl == 5) ? sl_g : dump;
            asm volatile("ds_write2_b64 %0, %1, %2 offset0:16 offset1:20\n\tds_write2_b64 %0, %3, %4 offset0:24 offset1:28" ::"v"(ca), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]) : "memory");
        }
        const double e = (KO & 4) ? 1e-9 : fma(-piv, r, 1.0);
        const double w = lcm * r * 1e-3;
        const double w2 = fma(w, e, w);
        if (!(KO & 8)) {
            const double rp = fma(r, e, r);
            const unsigned ra = (lane == 0) ? rpo : rdump;
            asm volatile("ds_write_b64 %0, %1 offset:8" ::"v"(ra), "v"(rp) : "memory");
        }
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(d[1]) : "v"(w2));
        if (!(KO & 1)) asm volatile("ds_bpermute_b32 %0, %2, %3 offset:64\n\tds_bpermute_b32 %1, %2, %4 offset:64" : "=&v"(rlo), "=&v"(rhi) : "v"(cc4), "v"(__double2loint(d[1])), "v"(__double2hiint(d[1])) : "memory");
        else { rlo = __double2loint(d[1]); rhi = __double2hiint(d[1]); }
        if (!(KO & 2)) piv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(d[1]), 21), __builtin_amdgcn_readlane(__double2loint(d[1]), 21));
        else piv = d[1];
        if (!(KO & 32)) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(d[2]), "+v"(d[3]), "+v"(d[0]) : "v"(w2));
    }
    long long t1 = __builtin_readcyclecounter();
    out[lane] = d[0] + d[1] + d[2] + d[3] + pivs + sL[lane] + sRp[lane & 15];
    if (lane == 0) cyc[slot] = t1 - t0;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 64 * 8);
    for (int rep = 0; rep < 2; ++rep) {
#define RUN(KO, S) hipLaunchKernelGGL(k<KO>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5, S)
        RUN(0, 0); RUN(1, 1); RUN(2, 2); RUN(4, 3); RUN(8, 4); RUN(16, 5); RUN(32, 6); RUN(1 | 2, 7); RUN(1 | 2 | 4, 8); RUN(1 | 2 | 4 | 8, 9); RUN(1 | 2 | 4 | 8 | 16 | 32, 10);
        hipDeviceSynchronize();
    }
    long long h[16]; hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
    const char* nm[] = {"full body", "no bpermute", "no readlane", "no rcp/Newton", "no publish (column, 1/a_pp)", "no lcm select", "one fmac_dpp", "no bpermute, no readlane", "  .. and no rcp", "  .. and no publish", "  .. bare: mul, fma, one fmac_dpp"};
    for (int i = 0; i < 11; ++i) printf("%-36s %.1f cycles / pivot\n", nm[i], h[i] / (double)N);
    return 0;
}
