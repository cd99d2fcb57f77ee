/* A pure-C caller of include/bas.h (no Python, no torch): hipMalloc'd buffers, the render entry points of
 * libbas_hip.so, checked against the oracle's plain-C restatement (oracle/bas_oracle_fir.c).
 * Built by tests/cabi/Makefile (gcc), run by tests/test_gpu_parity.py::test_c_caller_of_the_abi.
 *   cabi_check [scale]          the stored-IR entry points: bas_render_mix_f32 + bas_scale_by_peak_f32
 *   cabi_check fused [scale]    the DEFAULT path of the Python layer, from C: bas_table_pack_f32 ->
 *                               bas_traj_params_branch_f64 -> bas_interp2d_plan_f32 -> bas_render_mix_fused_f32 (peak rule
 *                               in the kernel tail) -> bas_render_status, on a scene the plan gives the split-role kernel;
 *                               then the same window as one block of a stream (bas_render_stream_block_f32);
 *                               the checker is fed the chunk IRs of bas_interp2d_f32 (itself pinned to the reference's
 *                               goldens by the Python tests)
 * Exit code 0 = parity within 1e-5 norm-relative. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <hip/hip_runtime_api.h>

#include "../../include/bas.h"
#include "../../oracle/bas_oracle_fir.h"

#define CHECK_HIP(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d hip error %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_BAS(e) do { int r_ = (e); if (r_ != 0) { fprintf(stderr, "%s:%d bas error %d: %s\n", __FILE__, __LINE__, r_, bas_last_error()); return 3; } } while (0)

static unsigned lcg_state = 2024u;
static double lcg(void) {                     /* uniform in [-1, 1) */
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (double)(lcg_state >> 8) / 8388608.0 - 1.0;
}

/* ---- the default path ------------------------------------------------------------------------------------------- */
static int check_default_path(double scale) {
    enum { NDIR = 187, U = 8, L = 128, M = L * U, K = 512, S = 32, N_SRC = 12 };
    const long n = 300000;                                    /* 37 tiles of 8192 x 12 sources = 444 units > 256 CUs, and too many tiles of 2048 for the four-wave kernel */
    const long T_in = bas_oracle_in_length(n, K), T_out = T_in + L - 1;
    const int n_q = (int)(T_in / K) + 1;
    const long n_query = (long)N_SRC * n_q;
    /* the direction grid of sphere.py:124-319: rings at -45 .. 45 (24 azimuths), 60 (12), 75 (6), 90 (1); node
     * azimuths float32(deg) * float32(2 pi / 360) as sphere.py:318 builds them */
    double ring_elev[10];
    int32_t ring_start[10], ring_count[10];
    float node_az[NDIR];
    const float deg2rad32 = (float)(2.0 * 3.14159265358979323846 / 360.0);
    int at = 0;
    for (int r = 0; r < 10; ++r) {
        ring_elev[r] = (-45.0 + 15.0 * r) * 3.14159265358979323846 / 180.0;
        ring_count[r] = r < 7 ? 24 : (r == 7 ? 12 : (r == 8 ? 6 : 1));
        ring_start[r] = at;
        for (int i = 0; i < ring_count[r]; ++i) node_az[at++] = (float)(i * (360 / ring_count[r])) * deg2rad32;
    }
    if (at != NDIR) return 1;
    /* a synthetic table: decaying noise bursts with direction-dependent onsets, antisymmetric delay differences */
    float *irs = (float *)malloc(sizeof(float) * 2 * NDIR * M);
    double *diffs = (double *)malloc(sizeof(double) * 2 * NDIR * NDIR);
    double onset[2][NDIR];
    for (int e = 0; e < 2; ++e)
        for (int p = 0; p < NDIR; ++p) {
            onset[e][p] = 6.0 + 3.0 * lcg();
            for (int i = 0; i < M; ++i) {
                const double t = (double)i / U - onset[e][p];
                irs[((size_t)e * NDIR + p) * M + i] = (float)(t < 0 ? 0.0 : lcg() * exp(-t / 14.0) * 0.3);
            }
        }
    for (int e = 0; e < 2; ++e)
        for (int p = 0; p < NDIR; ++p)
            for (int q = 0; q < NDIR; ++q) diffs[((size_t)e * NDIR + p) * NDIR + q] = onset[e][q] - onset[e][p];
    /* audio and trajectories (radians at t = 0, K, .., T_in per source: apply_hrtf.py:429, :435) */
    float *x32 = (float *)calloc((size_t)N_SRC * T_in, sizeof(float));
    double *x64 = (double *)calloc((size_t)N_SRC * T_in, sizeof(double));
    double *elev = (double *)malloc(sizeof(double) * n_query), *azim = (double *)malloc(sizeof(double) * n_query);
    for (int s = 0; s < N_SRC; ++s) {
        for (long m = 0; m < n; ++m) {
            x32[s * T_in + m] = (float)(scale * lcg());
            x64[s * T_in + m] = (double)x32[s * T_in + m];
        }
        for (int c = 0; c < n_q; ++c) {
            const double t = (double)c * K;
            elev[(long)s * n_q + c] = -0.6 + 2.0 * t / (double)T_in + 0.05 * s;        /* through every ring, past the pole */
            azim[(long)s * n_q + c] = 0.7 * s + 2.0 * 3.14159265358979323846 * t / (44100.0 * (1.5 + 0.2 * s));
        }
    }
    float *d_irs, *d_packed, *d_node, *d_x, *d_y, *d_peak, *d_H;
    double *d_diffs, *d_elev, *d_azim, *d_w;
    int32_t *d_idx;
    void *d_plans, *d_ws, *d_wsH;
    const size_t packed_floats = bas_table_packed_floats(NDIR, M, U);
    const size_t plan_bytes = bas_interp2d_workspace_bytes((int)n_query);
    const size_t ws_bytes = bas_render_fused_workspace_bytes(N_SRC, T_in, K, S, L);
    if (!bas_render_fused_supported(N_SRC, T_in, K, S, L)) { fprintf(stderr, "scene not served by the fused kernel\n"); return 1; }
    CHECK_HIP(hipMalloc((void **)&d_irs, sizeof(float) * 2 * NDIR * M));
    CHECK_HIP(hipMalloc((void **)&d_packed, sizeof(float) * packed_floats));
    CHECK_HIP(hipMalloc((void **)&d_diffs, sizeof(double) * 2 * NDIR * NDIR));
    CHECK_HIP(hipMalloc((void **)&d_node, sizeof(float) * NDIR));
    CHECK_HIP(hipMalloc((void **)&d_elev, sizeof(double) * n_query));
    CHECK_HIP(hipMalloc((void **)&d_azim, sizeof(double) * n_query));
    CHECK_HIP(hipMalloc((void **)&d_idx, sizeof(int32_t) * 4 * n_query));
    CHECK_HIP(hipMalloc((void **)&d_w, sizeof(double) * 3 * n_query));
    CHECK_HIP(hipMalloc(&d_plans, plan_bytes));
    CHECK_HIP(hipMalloc(&d_wsH, plan_bytes));
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_HIP(hipMalloc((void **)&d_x, sizeof(float) * N_SRC * T_in));
    CHECK_HIP(hipMalloc((void **)&d_y, sizeof(float) * 2 * T_out));
    CHECK_HIP(hipMalloc((void **)&d_peak, sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_H, sizeof(float) * (size_t)n_query * 2 * L));
    CHECK_HIP(hipMemset(d_ws, 0, BAS_WS_CONTROL_BYTES));      /* the control block: once per workspace (include/bas.h) */
    CHECK_HIP(hipMemcpy(d_irs, irs, sizeof(float) * 2 * NDIR * M, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_diffs, diffs, sizeof(double) * 2 * NDIR * NDIR, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_node, node_az, sizeof(float) * NDIR, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_elev, elev, sizeof(double) * n_query, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_azim, azim, sizeof(double) * n_query, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_x, x32, sizeof(float) * N_SRC * T_in, hipMemcpyHostToDevice));

    CHECK_BAS(bas_table_pack_f32(d_irs, NDIR, M, U, d_packed, NULL));                                   /* a1 */
    CHECK_BAS(bas_traj_params_branch_f64(d_elev, d_azim, n_query, ring_elev, ring_start, ring_count, d_node, d_idx, d_w,
                                         BAS_BRANCH_F64, NULL));                                        /* a3 */
    CHECK_BAS(bas_interp2d_plan_f32(d_diffs, d_idx, d_w, (int)n_query, NDIR, L, U, d_plans, plan_bytes, NULL));   /* a6, plans */
    CHECK_BAS(bas_render_mix_fused_f32(d_x, T_in, d_packed, d_plans, N_SRC, T_in, K, S, L, U, NDIR, d_y, 0, d_peak, 1,
                                       d_ws, ws_bytes, NULL));                                          /* a6 + a7 + a8 + rule */
    const int status = bas_render_status(d_ws, ws_bytes, NULL);
    /* the checker's chunk IRs: the library's stored-IR interpolate_2d */
    CHECK_BAS(bas_interp2d_f32(d_packed, d_diffs, d_idx, d_w, (int)n_query, NDIR, L, U, d_H, d_wsH, plan_bytes, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float *y = (float *)malloc(sizeof(float) * 2 * T_out), *H32 = (float *)malloc(sizeof(float) * (size_t)n_query * 2 * L);
    double *H64 = (double *)malloc(sizeof(double) * (size_t)n_query * 2 * L);
    float peak = 0.f;
    CHECK_HIP(hipMemcpy(y, d_y, sizeof(float) * 2 * T_out, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(H32, d_H, sizeof(float) * (size_t)n_query * 2 * L, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&peak, d_peak, sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < (size_t)n_query * 2 * L; ++i) H64[i] = (double)H32[i];
    double *acc = (double *)calloc((size_t)2 * T_out, sizeof(double));
    float *want = (float *)malloc(sizeof(float) * 2 * T_out);
    for (int s = 0; s < N_SRC; ++s)
        bas_oracle_render_accumulate(x64 + s * T_in, n, K, S, H64 + (size_t)s * n_q * 2 * L, L, acc);
    bas_oracle_finish(acc, T_out, 1, want);
    double err = 0, ref = 0;
    for (long i = 0; i < T_out; ++i)
        for (int e = 0; e < 2; ++e) {
            const double w = want[2 * i + e], g = y[e * T_out + i];
            if (!(fabs(g - w) <= err)) err = fabs(g - w);     /* (NaN counts as an error) */
            if (fabs(w) > ref) ref = fabs(w);
        }
    printf("cabi_check fused: kernel %s, peak before the rule %.4f, status %d, rel err %.3e\n",
           bas_render_fused_kernel_name(N_SRC, T_in, K, S, L), peak, status, err / ref);
    /* ---- the same window as ONE BLOCK OF A STREAM (bas_render_stream_block_f32): halo = one chunk, the block = the rest.
     * Its output must be the un-normalised render bit for bit; the carried state must have moved as bas.h says. */
    int stream_ok = 1;
    {
        const int halo = K, nh = 1, nb = n_q - 1;
        const long B = T_in - halo;
        float *d_x2, *d_y2, *d_y3, *d_rpeak;
        double *d_e2, *d_a2, *d_last;
        CHECK_HIP(hipMalloc((void **)&d_x2, sizeof(float) * N_SRC * T_in));
        CHECK_HIP(hipMalloc((void **)&d_y2, sizeof(float) * 2 * T_out));
        CHECK_HIP(hipMalloc((void **)&d_y3, sizeof(float) * 2 * T_out));
        CHECK_HIP(hipMalloc((void **)&d_rpeak, sizeof(float)));
        CHECK_HIP(hipMalloc((void **)&d_e2, sizeof(double) * n_query));
        CHECK_HIP(hipMalloc((void **)&d_a2, sizeof(double) * n_query));
        CHECK_HIP(hipMalloc((void **)&d_last, sizeof(double) * 2 * N_SRC));
        CHECK_HIP(hipMemcpy(d_x2, d_x, sizeof(float) * N_SRC * T_in, hipMemcpyDeviceToDevice));
        CHECK_HIP(hipMemcpy(d_e2, d_elev, sizeof(double) * n_query, hipMemcpyDeviceToDevice));
        CHECK_HIP(hipMemcpy(d_a2, d_azim, sizeof(double) * n_query, hipMemcpyDeviceToDevice));
        CHECK_HIP(hipMemset(d_rpeak, 0, sizeof(float)));
        CHECK_BAS(bas_render_mix_fused_f32(d_x, T_in, d_packed, d_plans, N_SRC, T_in, K, S, L, U, NDIR, d_y3, 0, NULL, 0, d_ws,
                                           ws_bytes, NULL));
        CHECK_BAS(bas_render_stream_block_f32(d_x2, T_in, d_packed, d_plans, N_SRC, T_in, K, S, L, U, NDIR, d_y2, d_ws, ws_bytes,
                                              halo, d_e2, d_a2, n_q, nh, nb, d_last, d_rpeak, NULL));
        CHECK_HIP(hipDeviceSynchronize());
        float *y2 = (float *)malloc(sizeof(float) * 2 * T_out), *y3 = (float *)malloc(sizeof(float) * 2 * T_out);
        float *x2 = (float *)malloc(sizeof(float) * N_SRC * T_in), rpeak = -1.f;
        double *e2 = (double *)malloc(sizeof(double) * n_query), *last = (double *)malloc(sizeof(double) * 2 * N_SRC);
        CHECK_HIP(hipMemcpy(y2, d_y2, sizeof(float) * 2 * T_out, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(y3, d_y3, sizeof(float) * 2 * T_out, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(x2, d_x2, sizeof(float) * N_SRC * T_in, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(e2, d_e2, sizeof(double) * n_query, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(last, d_last, sizeof(double) * 2 * N_SRC, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(&rpeak, d_rpeak, sizeof(float), hipMemcpyDeviceToHost));
        float emitted = 0.f;
        for (long i = 0; i < 2 * T_out; ++i) stream_ok &= y2[i] == y3[i];
        for (int e = 0; e < 2; ++e)
            for (long i = halo; i < halo + B; ++i) emitted = fmaxf(emitted, fabsf(y3[e * T_out + i]));
        stream_ok &= rpeak == emitted;
        for (int s = 0; s < N_SRC; ++s) {
            for (long i = 0; i < halo; ++i) stream_ok &= x2[s * T_in + i] == x32[s * T_in + B + i];
            stream_ok &= last[s] == elev[(long)s * n_q + n_q - 1] && last[N_SRC + s] == azim[(long)s * n_q + n_q - 1];
            stream_ok &= e2[(long)s * n_q] == elev[(long)s * n_q + nb - 1];
        }
        printf("cabi_check fused: the window as one stream block: %s (running peak %.4f)\n", stream_ok ? "ok" : "MISMATCH", rpeak);
        hipFree(d_x2); hipFree(d_y2); hipFree(d_y3); hipFree(d_rpeak); hipFree(d_e2); hipFree(d_a2); hipFree(d_last);
    }
    hipFree(d_irs); hipFree(d_packed); hipFree(d_diffs); hipFree(d_node); hipFree(d_elev); hipFree(d_azim); hipFree(d_idx);
    hipFree(d_w); hipFree(d_plans); hipFree(d_wsH); hipFree(d_ws); hipFree(d_x); hipFree(d_y); hipFree(d_peak); hipFree(d_H);
    return (err / ref <= 1e-5 && status == 0 && stream_ok) ? 0 : 1;
}

int main(int argc, char **argv) {
    if (argc > 1 && argv[1][0] == 'f') {
        if (bas_version() != BAS_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
        return check_default_path(argc > 2 ? atof(argv[2]) : 0.02);
    }
    const int n_src = 3, K = 512, S = 32, L = 128;
    const long n = 5000;
    const double scale = argc > 1 ? atof(argv[1]) : 0.1;     /* > 1.5 or so makes the peak rule fire */
    const long T_in = bas_oracle_in_length(n, K), T_out = T_in + L - 1;
    const int n_q = (int)(T_in / K) + 1;
    if (bas_version() != BAS_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }

    /* inputs: noise, and chunk IRs that drift from chunk to chunk (a decaying burst) */
    float *x32 = (float *)calloc((size_t)n_src * T_in, sizeof(float));
    double *x64 = (double *)calloc((size_t)n_src * T_in, sizeof(double));
    float *H32 = (float *)malloc(sizeof(float) * (size_t)n_src * n_q * 2 * L);
    double *H64 = (double *)malloc(sizeof(double) * (size_t)n_src * n_q * 2 * L);
    for (int s = 0; s < n_src; ++s)
        for (long m = 0; m < n; ++m) {
            x32[s * T_in + m] = (float)(scale * lcg());
            x64[s * T_in + m] = (double)x32[s * T_in + m];
        }
    for (size_t i = 0; i < (size_t)n_src * n_q * 2 * L; ++i) {
        const int k = (int)(i % L);
        H32[i] = (float)(lcg() * exp(-k / 24.0) * 0.2);
        H64[i] = (double)H32[i];
    }

    float *dx, *dH, *dy, *dpeak;
    void *dws;
    const size_t ws_bytes = bas_render_workspace_bytes(n_src, T_in, K, S, L);
    CHECK_HIP(hipMalloc((void **)&dx, sizeof(float) * n_src * T_in));
    CHECK_HIP(hipMalloc((void **)&dH, sizeof(float) * (size_t)n_src * n_q * 2 * L));
    CHECK_HIP(hipMalloc((void **)&dy, sizeof(float) * 2 * T_out));
    CHECK_HIP(hipMalloc((void **)&dpeak, sizeof(float)));
    CHECK_HIP(hipMalloc(&dws, ws_bytes));
    CHECK_HIP(hipMemset(dws, 0, BAS_WS_CONTROL_BYTES));
    CHECK_HIP(hipMemcpy(dx, x32, sizeof(float) * n_src * T_in, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dH, H32, sizeof(float) * (size_t)n_src * n_q * 2 * L, hipMemcpyHostToDevice));

    CHECK_BAS(bas_render_mix_f32(dx, T_in, dH, n_src, T_in, K, S, L, dy, 0, dpeak, dws, ws_bytes, NULL));
    CHECK_BAS(bas_scale_by_peak_f32(dy, 2 * T_out, dpeak, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float *y = (float *)malloc(sizeof(float) * 2 * T_out);
    float peak = 0.f;
    CHECK_HIP(hipMemcpy(y, dy, sizeof(float) * 2 * T_out, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&peak, dpeak, sizeof(float), hipMemcpyDeviceToHost));

    /* the checker: mix = sum of the un-normalised binary64 renders, float32, peak rule once */
    double *acc = (double *)calloc((size_t)2 * T_out, sizeof(double));
    float *want = (float *)malloc(sizeof(float) * 2 * T_out);
    for (int s = 0; s < n_src; ++s)
        bas_oracle_render_accumulate(x64 + s * T_in, n, K, S, H64 + (size_t)s * n_q * 2 * L, L, acc);
    bas_oracle_finish(acc, T_out, 1, want);

    double err = 0, ref = 0;
    for (long i = 0; i < T_out; ++i)
        for (int e = 0; e < 2; ++e) {
            const double w = want[2 * i + e], g = y[e * T_out + i];        /* library: [2][T_out]; oracle: [T_out][2] */
            if (fabs(g - w) > err) err = fabs(g - w);
            if (fabs(w) > ref) ref = fabs(w);
        }
    /* an argument error must come back as a code + text, nothing enqueued */
    const int rc = bas_render_mix_f32(dx, T_in, dH, n_src, T_in, K, 48, L, dy, 0, dpeak, dws, ws_bytes, NULL);
    printf("cabi_check: kernel %s, peak before the rule %.4f, rel err %.3e, bad-argument rc %d (%s)\n",
           bas_render_kernel_name(n_src, T_in, K, S, L), peak, err / ref, rc, bas_last_error());
    hipFree(dx); hipFree(dH); hipFree(dy); hipFree(dpeak); hipFree(dws);
    return (err / ref <= 1e-5 && rc == BAS_E_SHAPE) ? 0 : 1;
}
