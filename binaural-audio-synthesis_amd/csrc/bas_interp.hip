// HRIR table kernels: layout packing (a1), fractional circular shift (a4), ring
// interpolation (a5) and batched 2-D delay-compensated interpolation (a6).
// Reference semantics: apply_hrtf.py:53-106, :127-165, :171-281.
//
// Closed forms.  With U = upsampling, M = L*U, T_e[p] row p of
// ear e, d_e the delay-difference matrix and
//     S(x,s)[n] = (1-f) x[(n - floor s) mod M] + f x[(n - floor s - 1) mod M],  f = s - floor s
// (identical to the reference's floor/ceil blend: for integer s, f = 0),
//     ring (p,q,al):  D = U d_e[p,q];  B = (1-al) T_e[p] + al S(T_e[q], -D);  R = S(B, al D)
//     2-D:  Dv = U (-al_t D_t/U + d_e[pt,pb] + al_b D_b/U)
//           C = (1-a) S(R_b, -Dv) + a R_t;   h_e[m] = S(C, (1-a) Dv)[m U]
// Every shift amount is evaluated in binary64 (|D| reaches a few hundred upsampled
// samples, where a binary32 shift would already cost ~1e-5 of the IR peak); table
// samples and blends are binary32.
//
// Evaluation is lazy: output tap m of one ear needs C at 2 adjacent upsampled
// positions -> R_b at 3, R_t at 2 -> B_b at 4, B_t at 3 -> 4+5+3+4 = 16 table
// samples.  All positions are (m*U + c) mod M with c uniform over the query, so in
// the phase-plane layout packed[e][p][c mod U][.] every one of the 16 reads is a
// contiguous run across the lanes (lane = tap m).
#include "bas_internal.h"

// ---------------------------------------------------------------------------
// a1: [2][ndir][M] -> [2][ndir][U][L]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bas_pack_kernel(const float *__restrict__ irs, long rows,
                                                         int M, int U, int L,
                                                         float *__restrict__ packed) {
    long total = rows * (long)M;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        long row = i / M;
        int j = (int)(i - row * M);            // destination index inside the row: ph*L + idx
        int ph = j / L, idx = j - ph * L;
        packed[i] = irs[row * M + (long)idx * U + ph];
    }
}

extern "C" int bas_table_pack_f32(const float *irs, int ndir, int M, int U, float *packed,
                                  bas_stream_t stream) {
    BAS_REQUIRE(irs && packed, BAS_E_NULL, "bas_table_pack_f32: null pointer");
    BAS_REQUIRE(ndir > 0 && U > 0 && M > 0 && M % U == 0, BAS_E_SHAPE,
                "bas_table_pack_f32: need ndir>0, U>0, M>0, M %% U == 0 (ndir=%d M=%d U=%d)", ndir, M, U);
    long rows = 2L * ndir;
    long total = rows * M;
    int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(bas_pack_kernel, dim3(grid), dim3(256), 0, bas_stream(stream), irs, rows, M, U,
                       M / U, packed);
    return bas_check_launch("bas_table_pack_f32");
}

// ---------------------------------------------------------------------------
// a4: delay_signal_float on plain rows
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bas_delay_kernel(const float *__restrict__ x,
                                                          const double *__restrict__ shifts, int n,
                                                          int M, int down, int Mout,
                                                          float *__restrict__ y) {
    long total = (long)n * Mout;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        int row = (int)(i / Mout);
        int j = (int)(i - (long)row * Mout);
        long long b;
        float f;
        bas_split_shift(shifts[row], b, f);
        int p0 = bas_pmod((long long)j * down - b, M);
        int p1 = p0 == 0 ? M - 1 : p0 - 1;
        const float *xr = x + (long)row * M;
        y[i] = (1.0f - f) * xr[p0] + f * xr[p1];
    }
}

extern "C" int bas_delay_signal_f32(const float *x, const double *shifts, int n, int M, int down,
                                    float *y, bas_stream_t stream) {
    BAS_REQUIRE(x && shifts && y, BAS_E_NULL, "bas_delay_signal_f32: null pointer");
    BAS_REQUIRE(n >= 0 && M > 0 && down >= 1, BAS_E_SHAPE,
                "bas_delay_signal_f32: need n>=0, M>0, down>=1 (n=%d M=%d down=%d)", n, M, down);
    if (n == 0) return 0;
    int Mout = (M + down - 1) / down;
    long total = (long)n * Mout;
    int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(bas_delay_kernel, dim3(grid), dim3(256), 0, bas_stream(stream), x, shifts, n, M,
                       down, Mout, y);
    return bas_check_launch("bas_delay_signal_f32");
}

// ---------------------------------------------------------------------------
// shared plan for one (query, ear): where to read and how to blend
// ---------------------------------------------------------------------------
struct RingPlan {
    int row_p, row_q;      // float offsets of packed rows T_e[p], T_e[q]
    int c_p, c_q;          // read j of the set starts at upsampled offset (c - j) mod M
    float f1, f2, al;      // fraction of S(T_q,-D), fraction of S(B, al D), ring weight
};

struct EarPlan {
    RingPlan top, bot;
    float f3, f4, a;
};

__device__ __forceinline__ int clamp_dir(int p, int ndir) { return p < 0 ? 0 : (p >= ndir ? ndir - 1 : p); }

// Table sample of a packed row at upsampled position (pos0 + c - j) mod M, where
// c in [0,M), 0 <= j <= 4 and pos0 = m*U is folded in through `m` (row index in a plane).
template <bool POW2>
__device__ __forceinline__ float tab_read(const float *__restrict__ packed, int row, int c, int j,
                                           int m, int L, int U, int ush, int M) {
    int cc = c - j;
    if (cc < 0) cc += M;
    int ph, o;
    if (POW2) {
        ph = cc & (U - 1);
        o = cc >> ush;
    } else {
        o = cc / U;
        ph = cc - o * U;
    }
    int idx = m + o;                 // m < L, o < L
    if (idx >= L) idx -= L;
    return packed[row + ph * L + idx];
}

// R = S(B, al D) at the positions (c_p - j), j = 0..NR-1, relative to the lane's tap
template <bool POW2, int NR>
__device__ __forceinline__ void ring_eval(const float *__restrict__ packed, const RingPlan &rp, int m,
                                           int L, int U, int ush, int M, float (&r)[NR]) {
    float tq[NR + 2], b[NR + 1];
#pragma unroll
    for (int j = 0; j < NR + 2; ++j) tq[j] = tab_read<POW2>(packed, rp.row_q, rp.c_q, j, m, L, U, ush, M);
#pragma unroll
    for (int j = 0; j < NR + 1; ++j) {
        float tp = tab_read<POW2>(packed, rp.row_p, rp.c_p, j, m, L, U, ush, M);
        float sq = (1.0f - rp.f1) * tq[j] + rp.f1 * tq[j + 1];       // S(T_q, -D)
        b[j] = (1.0f - rp.al) * tp + rp.al * sq;                     // apply_hrtf.py:90-91
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) r[j] = (1.0f - rp.f2) * b[j] + rp.f2 * b[j + 1];   // :98-99
}

// Fills the ring part of a plan.  `c_out` is the offset (mod M) at which R itself is
// wanted for j = 0.  Returns al*D (upsampled samples) through s2.
__device__ __forceinline__ void ring_plan(RingPlan &rp, const double *__restrict__ d, int e, int ndir,
                                           int p, int q, double al, int L, int U, long long c_out,
                                           double &s2) {
    int M = L * U;
    double D = (double)U * d[(long)p * ndir + q];                   // apply_hrtf.py:82-83
    s2 = al * D;                                                    // :94-95
    long long b1, b2;
    float f1, f2;
    bas_split_shift(-D, b1, f1);                                    // :86-87
    bas_split_shift(s2, b2, f2);                                    // :98-99
    rp.row_p = (e * ndir + p) * M;
    rp.row_q = (e * ndir + q) * M;
    long long cB = c_out - b2;                                      // B is read at (c_out - j) - b2 (-1)
    rp.c_p = bas_pmod(cB, M);
    rp.c_q = bas_pmod(cB - b1, M);
    rp.f1 = f1;
    rp.f2 = f2;
    rp.al = (float)al;
}

// ---------------------------------------------------------------------------
// a6: batched interpolate_2d
// ---------------------------------------------------------------------------
#define BAS_QB 16      // queries per workgroup

template <bool POW2>
__global__ __launch_bounds__(256) void bas_interp2d_kernel(const float *__restrict__ packed,
                                                             const double *__restrict__ diffs,
                                                             const int32_t *__restrict__ idx,
                                                             const double *__restrict__ w, int n,
                                                             int ndir, int L, int U, int ush,
                                                             float *__restrict__ H) {
    __shared__ EarPlan plans[BAS_QB * 2];
    const int q0 = blockIdx.x * BAS_QB;
    const int nq = n - q0 < BAS_QB ? n - q0 : BAS_QB;
    const int M = L * U;

    if ((int)threadIdx.x < nq * 2) {
        const int ql = threadIdx.x >> 1, e = threadIdx.x & 1;
        const long q = q0 + ql;
        const int pt = clamp_dir(idx[4 * q + 0], ndir), qt = clamp_dir(idx[4 * q + 1], ndir);
        const int pb = clamp_dir(idx[4 * q + 2], ndir), qb = clamp_dir(idx[4 * q + 3], ndir);
        const double at = w[3 * q + 0], ab = w[3 * q + 1], a = w[3 * q + 2];
        const double *d = diffs + (long)e * ndir * ndir;
        // delays of the two ring interpolations in non-upsampled samples (apply_hrtf.py:106)
        const double dt = (at * ((double)U * d[(long)pt * ndir + qt])) / (double)U;
        const double db = (ab * ((double)U * d[(long)pb * ndir + qb])) / (double)U;
        const double dv = (double)U * (-dt + d[(long)pt * ndir + pb] + db);     // :246-252
        long long b3, b4;
        float f3, f4;
        bas_split_shift(-dv, b3, f3);                                           // :254-255
        bas_split_shift((1.0 - a) * dv, b4, f4);                                // :272-277
        EarPlan &pl = plans[threadIdx.x];
        double s2;
        // C is read at -b4 - j; R_t at the same offsets; R_b at (-b4 - b3) - j
        ring_plan(pl.top, d, e, ndir, pt, qt, at, L, U, -b4, s2);
        ring_plan(pl.bot, d, e, ndir, pb, qb, ab, L, U, -b4 - b3, s2);
        pl.f3 = f3;
        pl.f4 = f4;
        pl.a = (float)a;
    }
    __syncthreads();

    for (int i = threadIdx.x; i < 2 * L; i += 256) {
        const int e = i >= L ? 1 : 0;
        const int m = i - e * L;
        for (int ql = 0; ql < nq; ++ql) {
            const EarPlan &pl = plans[ql * 2 + e];
            float rb[3], rt[2];
            ring_eval<POW2, 3>(packed, pl.bot, m, L, U, ush, M, rb);
            ring_eval<POW2, 2>(packed, pl.top, m, L, U, ush, M, rt);
            float c0 = (1.0f - pl.a) * ((1.0f - pl.f3) * rb[0] + pl.f3 * rb[1]) + pl.a * rt[0];   // :268-269
            float c1 = (1.0f - pl.a) * ((1.0f - pl.f3) * rb[1] + pl.f3 * rb[2]) + pl.a * rt[1];
            H[(long)(q0 + ql) * 2 * L + i] = (1.0f - pl.f4) * c0 + pl.f4 * c1;                     // :276-277
        }
    }
}

static int pow2_shift(int U) {
    for (int s = 0; s < 30; ++s)
        if ((1 << s) == U) return s;
    return -1;
}

extern "C" int bas_interp2d_f32(const float *packed, const double *diffs, const int32_t *idx,
                                const double *w, int n, int ndir, int L, int U, float *H,
                                bas_stream_t stream) {
    BAS_REQUIRE(packed && diffs && idx && w && H, BAS_E_NULL, "bas_interp2d_f32: null pointer");
    BAS_REQUIRE(n >= 0 && ndir > 0 && L > 0 && U > 0, BAS_E_SHAPE,
                "bas_interp2d_f32: need n>=0, ndir>0, L>0, U>0 (n=%d ndir=%d L=%d U=%d)", n, ndir, L, U);
    BAS_REQUIRE((long)2 * ndir * L * U < (1L << 31), BAS_E_SHAPE, "bas_interp2d_f32: table too large");
    if (n == 0) return 0;
    int grid = (n + BAS_QB - 1) / BAS_QB;
    int ush = pow2_shift(U);
    if (ush >= 0)
        hipLaunchKernelGGL(bas_interp2d_kernel<true>, dim3(grid), dim3(256), 0, bas_stream(stream), packed,
                           diffs, idx, w, n, ndir, L, U, ush, H);
    else
        hipLaunchKernelGGL(bas_interp2d_kernel<false>, dim3(grid), dim3(256), 0, bas_stream(stream), packed,
                           diffs, idx, w, n, ndir, L, U, 0, H);
    return bas_check_launch("bas_interp2d_f32");
}

// ---------------------------------------------------------------------------
// a5: ring interpolation alone (both output rates)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bas_ring_kernel(const float *__restrict__ packed,
                                                         const double *__restrict__ diffs,
                                                         const int32_t *__restrict__ pq,
                                                         const double *__restrict__ alpha, int n,
                                                         int ndir, int L, int U, int step, int Mout,
                                                         float *__restrict__ out,
                                                         double *__restrict__ delays) {
    const int M = L * U;
    long total = (long)n * 2 * Mout;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        long qe = i / Mout;
        int j = (int)(i - qe * Mout);
        int q = (int)(qe >> 1), e = (int)(qe & 1);
        int p = clamp_dir(pq[2 * q], ndir), r = clamp_dir(pq[2 * q + 1], ndir);
        RingPlan rp;
        double s2;
        ring_plan(rp, diffs + (long)e * ndir * ndir, e, ndir, p, r, alpha[q], L, U, (long long)j * step, s2);
        // generic-position reads (the lane's own position is folded into c): m = 0
        float res[1];
        ring_eval<false, 1>(packed, rp, 0, L, U, 0, M, res);
        out[i] = res[0];
        if (delays && j == 0) delays[qe] = s2 / (double)U;             // apply_hrtf.py:106
    }
}

extern "C" int bas_ring_interp_f32(const float *packed, const double *diffs, const int32_t *pq,
                                   const double *alpha, int n, int ndir, int L, int U,
                                   int return_upsampled, float *out, double *delays,
                                   bas_stream_t stream) {
    BAS_REQUIRE(packed && diffs && pq && alpha && out, BAS_E_NULL, "bas_ring_interp_f32: null pointer");
    BAS_REQUIRE(n >= 0 && ndir > 0 && L > 0 && U > 0, BAS_E_SHAPE,
                "bas_ring_interp_f32: need n>=0, ndir>0, L>0, U>0 (n=%d ndir=%d L=%d U=%d)", n, ndir, L, U);
    BAS_REQUIRE((long)2 * ndir * L * U < (1L << 31), BAS_E_SHAPE, "bas_ring_interp_f32: table too large");
    if (n == 0) return 0;
    int step = return_upsampled ? 1 : U;
    int Mout = return_upsampled ? L * U : L;
    long total = (long)n * 2 * Mout;
    int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(bas_ring_kernel, dim3(grid), dim3(256), 0, bas_stream(stream), packed, diffs, pq,
                       alpha, n, ndir, L, U, step, Mout, out, delays);
    return bas_check_launch("bas_ring_interp_f32");
}
