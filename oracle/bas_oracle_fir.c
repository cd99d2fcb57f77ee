/* TEST INFRASTRUCTURE ONLY: plain-C restatement of the loops of make_signal_move_2d
 * (/root/reference apply_hrtf.py:431-464) for given chunk IRs.  Only tests/ may link or load it
 * (tests/test_oracle_golden.py pins it to the reference-generated goldens through the numpy oracle's chunk
 * IRs; tests/cabi/cabi_check.c uses it as the checker of a pure-C caller of include/bas.h).
 * The product (libbas_hip.so) never links this file.
 *
 * Parity: PINNED to tests/golden/render_*.npz within 1e-6 norm-relative (not bit-exact: numpy's convolve sums
 * its dot products in a different order than the plain loops below; the numpy oracle is the bit-exact one). */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "bas_oracle_fir.h"

long bas_oracle_in_length(long n, int K) { return (n + K - 1) / K * K; }      /* apply_hrtf.py:405 */

void bas_oracle_render_accumulate(const double *x, long n, int K, int S, const double *irs, int L, double *acc) {
    const long in_length = bas_oracle_in_length(n, K);
    const long out_length = in_length + L - 1;                                /* :410 */
    double *h = (double *)malloc(sizeof(double) * 2 * (size_t)L);
    for (long i = 0, c = 0; i < in_length; i += K, ++c) {                     /* :431 */
        const double *h0 = irs + (size_t)c * 2 * L, *h1 = h0 + 2 * (size_t)L; /* :434-435 */
        for (int j = 0; j < K; j += S) {                                      /* :438 */
            const double alpha = (double)j / (double)K;                       /* :442 */
            for (int t = 0; t < 2 * L; ++t) h[t] = (1 - alpha) * h0[t] + alpha * h1[t];   /* :443 */
            for (int a = 0; a < S; ++a) {                                     /* direct convolution, :445-446 */
                const long m = i + j + a;
                const double xm = m < n ? x[m] : 0.0;                         /* zero padding, :406 */
                if (xm == 0.0) continue;
                for (int k = 0; k < L; ++k) {                                 /* overlap-add, :450-453 */
                    acc[m + k] += xm * h[k];
                    acc[out_length + m + k] += xm * h[L + k];
                }
            }
        }
    }
    free(h);
}

void bas_oracle_finish(const double *acc, long out_length, int normalize, float *out) {
    float m = 0.f;
    for (long i = 0; i < out_length; ++i) {                                   /* astype(float32).T, :459 */
        out[2 * i] = (float)acc[i];
        out[2 * i + 1] = (float)acc[out_length + i];
        if (fabsf(out[2 * i]) > m) m = fabsf(out[2 * i]);
        if (fabsf(out[2 * i + 1]) > m) m = fabsf(out[2 * i + 1]);
    }
    if (normalize && m > 1.f)                                                 /* :462-464 */
        for (long i = 0; i < 2 * out_length; ++i) out[i] /= m;
}
