/* TEST INFRASTRUCTURE ONLY - see bas_oracle_fir.c. */
#ifndef BAS_ORACLE_FIR_H
#define BAS_ORACLE_FIR_H
#ifdef __cplusplus
extern "C" {
#endif
long bas_oracle_in_length(long n, int K);
/* acc[2][in_length + L - 1] (binary64) += un-normalised render of ONE source. */
void bas_oracle_render_accumulate(const double *x, long n, int K, int S, const double *irs, int L, double *acc);
/* out[out_length][2] float32 = (float) acc, then the peak rule if normalize != 0. */
void bas_oracle_finish(const double *acc, long out_length, int normalize, float *out);
#ifdef __cplusplus
}
#endif
#endif
