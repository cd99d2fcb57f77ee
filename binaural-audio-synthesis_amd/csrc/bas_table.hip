// The table builder's heavy parts on the device (gfx950): upsample_irs.m of the reference, run once per HRIR
// database, spends its time in 17 391 pairs x 2 ears of (cross-correlation -> x U band-limited resampling -> arg-max ->
// parabola, upsample_irs.m:59-101) and resamples the 374 HRIRs themselves (:37-44).  Both are small dense float64 sums; the
// host restatement (upsample_irs.py, numpy + FFTs) needs ~10 s on one core, these kernels a few milliseconds.
//
// PARITY UNPINNED, like the host restatement they are tested against (tests/test_gpu_table.py): Octave and the IRCAM data are
// not available to the build, so nothing here was compared with the reference's output - see upsample_irs.py's header for
// what `resample` is restated from.  Arithmetic: float64 throughout, direct sums (the host path correlates through FFTs:
// the two differ by rounding, ~1e-13 of a sample in the delays).
//
// The resampling filter h (octave_resample_filter: a Kaiser-windowed sinc of half-length Lh, 2 Lh + 1 taps) is designed on
// the host and handed over.  With q = 1 Octave's resample(x, p, 1) is, for j < lx p,
//     y[j] = sum_k h[j + Lh - p k] x[k]        over the k with 0 <= j + Lh - p k <= 2 Lh, 0 <= k < lx
// (upfirdn with one zero in front of h and the group delay Lh + 1 trimmed: upsample_irs.py, octave_resample).
#include "bas_internal.h"

// one output of the resampler from LDS-resident filter and signal
__device__ __forceinline__ double bas_up_sample(const double *hs, int Lh, int p, const double *xs, int lx, long j) {
    long k0 = j - Lh;
    k0 = k0 <= 0 ? 0 : (k0 + p - 1) / p;
    long k1 = (j + Lh) / p;
    if (k1 > lx - 1) k1 = lx - 1;
    double acc = 0.0;
    for (long k = k0; k <= k1; ++k) acc = fma(hs[j + Lh - (long)p * k], xs[k], acc);
    return acc;
}

// rows of x resampled by p: one workgroup per row
__global__ __launch_bounds__(256) void bas_resample_up_kernel(const double *__restrict__ x, int lx, const double *__restrict__ h,
                                                                int Lh, int p, double *__restrict__ y) {
    extern __shared__ double sh[];
    double *hs = sh, *xs = sh + (2 * Lh + 1);
    const double *xr = x + (long)blockIdx.x * lx;
    for (int i = threadIdx.x; i < 2 * Lh + 1; i += 256) hs[i] = h[i];
    for (int i = threadIdx.x; i < lx; i += 256) xs[i] = xr[i];
    __syncthreads();
    const long ly = (long)lx * p;
    double *yr = y + (long)blockIdx.x * ly;
    for (long j = threadIdx.x; j < ly; j += 256) yr[j] = bas_up_sample(hs, Lh, p, xs, lx, j);
}

#define BAS_DD_EDGE 1        // cross-correlation peak at the edge of its support (upsample_irs.m:70 would index out of range)
#define BAS_DD_NOT_MAX 2     // the middle point is not the (first) maximum (upsample_irs.m:92-93) - NaNs in the input
#define BAS_DD_COLLINEAR 3   // three collinear points around the peak (upsample_irs.m:98)

// delaydifference(irs[i], irs[j]) for every pair i < j (upsample_irs.m:22-28, :58-77): workgroup (j, i) of the grid;
// diffs[i][j] = d, diffs[j][i] = -d (the antisymmetry of :31-32); the diagonal is the caller's (zero).
__global__ __launch_bounds__(256) void bas_delaydiff_kernel(const double *__restrict__ irs, int n_dir, int n,
                                                              const double *__restrict__ h, int Lh, int p,
                                                              double *__restrict__ diffs,
                                                              unsigned long long *__restrict__ status) {
    const int j2 = blockIdx.x, i2 = blockIdx.y;
    if (j2 <= i2) return;                                    // (uniform: the lower triangle and the diagonal have no work)
    extern __shared__ double sh[];
    const int nc = 2 * n - 1;                                // lags of the cross-correlation
    double *hs = sh, *a = hs + (2 * Lh + 1), *b = a + n, *xc = b + n;
    __shared__ double r_val[4];
    __shared__ long long r_idx[4];
    for (int i = threadIdx.x; i < 2 * Lh + 1; i += 256) hs[i] = h[i];
    for (int i = threadIdx.x; i < n; i += 256) {
        a[i] = irs[(long)i2 * n + i];
        b[i] = irs[(long)j2 * n + i];
    }
    __syncthreads();
    // xc = conv(a reversed, b) (:66): xc[m] = sum_t a[n-1-t] b[m-t]
    for (int m = threadIdx.x; m < nc; m += 256) {
        const int t0 = m - (n - 1) > 0 ? m - (n - 1) : 0, t1 = m < n - 1 ? m : n - 1;
        double acc = 0.0;
        for (int t = t0; t <= t1; ++t) acc = fma(a[n - 1 - t], b[m - t], acc);
        xc[m] = acc;
    }
    __syncthreads();
    // the FIRST maximum of the resampled correlation (:69; Octave's max and numpy's argmax agree on ties): every thread
    // walks its samples in ascending order and keeps a strictly larger one; equal values across threads: the smaller index
    const long ly = (long)nc * p;
    double best = -__builtin_huge_val();
    long long bk = ly;
    for (long j = threadIdx.x; j < ly; j += 256) {
        const double v = bas_up_sample(hs, Lh, p, xc, nc, j);
        if (v > best) {
            best = v;
            bk = j;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(best, o);
        const long long ok = __shfl_xor(bk, o);
        if (ov > best || (ov == best && ok < bk)) {
            best = ov;
            bk = ok;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        r_val[threadIdx.x >> 6] = best;
        r_idx[threadIdx.x >> 6] = bk;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int w = 1; w < 4; ++w)
        if (r_val[w] > best || (r_val[w] == best && r_idx[w] < bk)) {
            best = r_val[w];
            bk = r_idx[w];
        }
    int code = 0;
    double d = 0.0;
    if (bk < 1 || bk > ly - 2) {
        code = BAS_DD_EDGE;
    } else {
        const double lo = bas_up_sample(hs, Lh, p, xc, nc, bk - 1), mid = best, hi = bas_up_sample(hs, Lh, p, xc, nc, bk + 1);
        if (!(mid > lo && mid >= hi)) {
            code = BAS_DD_NOT_MAX;
        } else {
            const double pa = 0.5 * (lo + hi - 2.0 * mid), pb = 0.5 * (hi - lo);   // parabolic_interpolation (:88-101)
            if (pa == 0.0) {
                code = BAS_DD_COLLINEAR;
            } else {
                const double peak = (double)bk - pb / (2.0 * pa);
                d = peak / (double)p - (double)(n - 1);      // (:73-76) in 0-based indexing
            }
        }
    }
    if (code) {                                              // the failing pair with the smallest (i, j) is reported, whoever ran first
        const unsigned long long pair = (unsigned long long)i2 * n_dir + j2;
        atomicMax(status, ((((unsigned long long)n_dir * n_dir) - pair) << 2) | (unsigned long long)code);
        return;
    }
    diffs[(long)i2 * n_dir + j2] = d;
    diffs[(long)j2 * n_dir + i2] = -d;
}

static int table_args(const char *who, const void *x, int rows, int lx, const void *h, int Lh, int p) {
    BAS_REQUIRE(x && h, BAS_E_NULL, "%s: null pointer", who);
    BAS_REQUIRE(rows > 0 && lx > 0 && Lh > 0 && p > 0, BAS_E_SHAPE, "%s: need rows, length, Lh, p > 0 (rows=%d length=%d Lh=%d p=%d)", who,
                rows, lx, Lh, p);
    BAS_REQUIRE((long)lx * p < (1L << 30) && Lh < (1 << 20), BAS_E_SHAPE, "%s: sizes too large", who);
    return 0;
}

extern "C" int bas_resample_up_f64(const double *x, int rows, int lx, const double *h, int Lh, int p, double *y,
                                   bas_stream_t stream) {
    int rc = table_args("bas_resample_up_f64", x, rows, lx, h, Lh, p);
    if (rc) return rc;
    BAS_REQUIRE(y, BAS_E_NULL, "bas_resample_up_f64: y is null");
    const size_t lds = sizeof(double) * (size_t)(2 * Lh + 1 + lx);
    BAS_REQUIRE(lds <= 64 * 1024, BAS_E_SHAPE, "bas_resample_up_f64: filter + one row (%zu bytes) must fit 64 KB of LDS", lds);
    hipLaunchKernelGGL(bas_resample_up_kernel, dim3((unsigned)rows), dim3(256), lds, bas_stream(stream), x, lx, h, Lh, p, y);
    return bas_check_launch("bas_resample_up_f64");
}

extern "C" int bas_delaydiffs_f64(const double *irs, int n_dir, int n_taps, const double *h, int Lh, int p, double *diffs,
                                  unsigned long long *status, bas_stream_t stream) {
    int rc = table_args("bas_delaydiffs_f64", irs, n_dir, n_taps, h, Lh, p);
    if (rc) return rc;
    BAS_REQUIRE(diffs && status, BAS_E_NULL, "bas_delaydiffs_f64: diffs or status is null");
    BAS_REQUIRE(reinterpret_cast<uintptr_t>(status) % 8 == 0, BAS_E_ALIGN, "bas_delaydiffs_f64: status must be 8-byte aligned");
    BAS_REQUIRE(n_dir <= 65535, BAS_E_SHAPE, "bas_delaydiffs_f64: more than 65535 directions");
    const size_t lds = sizeof(double) * (size_t)(2 * Lh + 1 + 2 * n_taps + 2 * n_taps - 1);
    BAS_REQUIRE(lds <= 64 * 1024, BAS_E_SHAPE,
                "bas_delaydiffs_f64: filter, two signals and their correlation (%zu bytes) must fit 64 KB of LDS", lds);
    hipStream_t st = bas_stream(stream);
    hipError_t e = hipMemsetAsync(diffs, 0, sizeof(double) * (size_t)n_dir * n_dir, st);
    if (e == hipSuccess) e = hipMemsetAsync(status, 0, sizeof(unsigned long long), st);
    if (e != hipSuccess) return bas_fail((int)e, "bas_delaydiffs_f64: hipMemsetAsync: %s", hipGetErrorString(e));
    if (n_dir < 2) return 0;
    hipLaunchKernelGGL(bas_delaydiff_kernel, dim3((unsigned)n_dir, (unsigned)n_dir), dim3(256), lds, st, irs, n_dir, n_taps, h,
                       Lh, p, diffs, status);
    return bas_check_launch("bas_delaydiffs_f64");
}
