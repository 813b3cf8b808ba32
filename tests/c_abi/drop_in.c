/* A plain-C caller of the drop-in boundary (include/msckf_mi355x.h): reads one update problem from a flat binary file,
 * calls msckf_create / msckf_update / msckf_destroy, writes dx | P_out | accepted back.  Built with gcc and linked against
 * libmsckf_mi355x.so by tests/test_gpu_c_abi.py -- no Python, no PyTorch, no HIP headers on the caller's side.
 *
 * file layout (little endian): int32 N, F, sumM, n_crit; then doubles P[d*d], cam_R[9N], cam_t[3N], cam_R0[9N], cam_t0[3N],
 * g[3], Kinv[9], sigma[1], obs_uv[2 sumM], idp_base[3F], idp_m[3F], idp_rho[F], chi2[n_crit]; then int32 view_ptr[F+1],
 * obs_slot[sumM]. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "msckf_mi355x.h"

static void* rd(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 8);
    if (bytes && fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: drop_in <problem.bin> <result.bin>\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    int32_t h[4];
    if (fread(h, 4, 4, f) != 4) return 2;
    const int32_t N = h[0], F = h[1], sumM = h[2], n_crit = h[3];
    const size_t d = 15 + 6 * (size_t)N;
    double* P = rd(f, d * d * 8);
    double* cam_R = rd(f, (size_t)N * 72); double* cam_t = rd(f, (size_t)N * 24);
    double* cam_R0 = rd(f, (size_t)N * 72); double* cam_t0 = rd(f, (size_t)N * 24);
    double* g = rd(f, 24); double* Kinv = rd(f, 72); double* sigma = rd(f, 8);
    double* obs_uv = rd(f, (size_t)sumM * 16);
    double* base = rd(f, (size_t)F * 24); double* m = rd(f, (size_t)F * 24); double* rho = rd(f, (size_t)F * 8);
    double* chi2 = rd(f, (size_t)n_crit * 8);
    int32_t* view_ptr = rd(f, ((size_t)F + 1) * 4); int32_t* obs_slot = rd(f, (size_t)sumM * 4);
    fclose(f);

    int max_track = 1;
    for (int32_t j = 0; j < F; ++j) if (view_ptr[j + 1] - view_ptr[j] > max_track) max_track = view_ptr[j + 1] - view_ptr[j];
    msckf_config cfg = {MSCKF_ABI_VERSION, 0, N, F > 0 ? F : 1, max_track, 0, 0, 0, MSCKF_DTYPE_F64, 0};
    msckf_ctx* ctx = NULL;
    int rc = msckf_create(&ctx, &cfg);
    if (rc != MSCKF_OK) { fprintf(stderr, "msckf_create: %s\n", msckf_strerror(rc)); return 3; }
    double* dx = malloc(d * 8); double* P_out = malloc(d * d * 8);
    uint8_t* acc = calloc(F > 0 ? F : 1, 1);
    msckf_stats st;
    rc = msckf_update(ctx, N, P, cam_R, cam_t, cam_R0, cam_t0, g, Kinv, sigma[0], F, view_ptr, obs_uv, obs_slot, base, m, rho,
                      chi2, n_crit, dx, P_out, acc, &st);
    if (rc < 0) { fprintf(stderr, "msckf_update: %s (%s)\n", msckf_strerror(rc), msckf_last_error(ctx)); return 4; }
    printf("status %d accepted %d rejected %d rows %d device_us %.0f\n", rc, st.n_accepted, st.n_rejected, st.stacked_rows, st.us_total);
    FILE* o = fopen(argv[2], "wb");
    int32_t s32 = rc;
    fwrite(&s32, 4, 1, o); fwrite(dx, 8, d, o); fwrite(P_out, 8, d * d, o); fwrite(acc, 1, F, o);
    fclose(o);
    msckf_destroy(ctx);
    return 0;
}
