/* A pure-C caller of include/bas.h (no Python, no torch): hipMalloc'd buffers, the render entry points of
 * libbas_hip.so, checked against the oracle's plain-C restatement (oracle/bas_oracle_fir.c).
 * Built by tests/cabi/Makefile (gcc), run by tests/test_gpu_parity.py::test_c_caller_of_the_abi.
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

int main(int argc, char **argv) {
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
