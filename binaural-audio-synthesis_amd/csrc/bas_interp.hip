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
// contiguous run across the lanes (a lane owns two adjacent taps and reads them with one
// 8-byte load).  Each plane carries a guard float at either end (front: copy of its last
// sample, back: copy of its first), so "one plane-sample earlier" and "the next tap" never
// need a wrap test: the reads of one set share a single per-lane offset and differ only in
// a wave-uniform base.  All blends are linear with query-uniform coefficients, hence
//     h[m] = sum_k W_k * sample_k(m)
// with the 16 weights W_k folded once per (query, ear) in binary64.
#include "bas_internal.h"
#include "bas_plan.h"
#include <stdlib.h>
#include <string.h>

// ---------------------------------------------------------------------------
// a1: [2][ndir][M] -> [2][ndir][U][L + 4]   (plane = [last sample][L samples][first three samples])
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bas_pack_kernel(const float *__restrict__ irs, long rows,
                                                         int M, int U, int L,
                                                         float *__restrict__ packed) {
    const int LP = BAS_PLANE(L);
    long total = rows * (long)U * LP;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        long row = i / ((long)U * LP);
        int j = (int)(i - row * U * LP);
        int ph = j / LP, g = j - ph * LP;          // g = 0 and g > L are the guards
        int idx = g == 0 ? L - 1 : (g > L ? (g - L - 1) % L : g - 1);
        packed[i] = irs[row * M + (long)idx * U + ph];
    }
}

extern "C" size_t bas_table_packed_floats(int ndir, int M, int U) {
    if (ndir <= 0 || U <= 0 || M <= 0 || M % U) return 0;
    return (size_t)2 * ndir * U * BAS_PLANE(M / U);
}

extern "C" int bas_table_pack_f32(const float *irs, int ndir, int M, int U, float *packed,
                                  bas_stream_t stream) {
    BAS_REQUIRE(irs && packed, BAS_E_NULL, "bas_table_pack_f32: null pointer");
    BAS_REQUIRE(ndir > 0 && U > 0 && M > 0 && M % U == 0, BAS_E_SHAPE,
                "bas_table_pack_f32: need ndir>0, U>0, M>0, M %% U == 0 (ndir=%d M=%d U=%d)", ndir, M, U);
    long rows = 2L * ndir;
    long total = rows * U * BAS_PLANE(M / U);
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
    if (cc < 0) cc = bas_pmod(cc, M);                        // tables shorter than the five-sample reach (M < 5)
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
    return packed[row + ph * BAS_PLANE(L) + 1 + idx];
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
                                           int p, int q, double al, int L, int U, int c_out,
                                           double &s2) {
    int M = L * U;
    double D = (double)U * d[(long)p * ndir + q];                   // apply_hrtf.py:82-83
    s2 = al * D;                                                    // :94-95
    float f1, f2;
    const int b1 = bas_split_shift_mod(-D, M, f1);                  // :86-87
    const int b2 = bas_split_shift_mod(s2, M, f2);                  // :98-99
    rp.row_p = (e * ndir + p) * U * BAS_PLANE(L);
    rp.row_q = (e * ndir + q) * U * BAS_PLANE(L);
    rp.c_p = bas_submod(c_out, b2, M);                              // B is read at (c_out - j) - b2 (-1)
    rp.c_q = bas_submod(rp.c_p, b1, M);
    rp.f1 = f1;
    rp.f2 = f2;
    rp.al = (float)al;
}

// ---------------------------------------------------------------------------
// a6: batched interpolate_2d
// ---------------------------------------------------------------------------
// Phase 1 (32 threads): one thread per (query, ear) folds indices, weights and delay differences
// into a plan: for each of the 4 read sets (bottom T_q x5, bottom T_p x4, top T_q x4, top T_p x3)
// the first read's plane address and plane offset, plus the 16 weights.
// Phase 2: wave w handles ear w&1 of every second query of the workgroup.  The plan is
// wave-uniform and lives in SGPRs; per output tap a lane computes 4 wrapped offsets (one per set),
// 16 loads off scalar bases and 16 FMAs with scalar weights.
// ---------------------------------------------------------------------------
// a3 + elevation bracket on the device: (elev, azim) -> (idx, w)
// ---------------------------------------------------------------------------
// Float64 branch of sphere.azim_to_interpolation_params (sphere.py:78-121) and of interpolate_2d's
// bracket (apply_hrtf.py:199-215, :261-266), same arithmetic as the host's
// sphere.interpolation_params_batch: node angles are the float32 table values, comparisons run in
// binary64, the interval width is a float32 subtraction, weights are binary64 divisions.
// `rings` (host-computed, so both sides use identical constants):
//   ring_elev[10] f64 = deg2rad(-45..90), ring_start[10], ring_count[10] int32, node_az[187] f32.
struct RingTable {
    double ring_elev[10];
    int ring_start[10];
    int ring_count[10];
};

// PYF: the reference's branch for a PYTHON-FLOAT azimuth under NumPy >= 2 (sphere.py:98-105, :119): a Python float is a
// "weak" scalar, so `index_azim[:,1] <= azim` and `(azim - before_azim) / (after_azim - before_azim)` are evaluated in the
// table's float32 - the azimuth is rounded to binary32 first (after the binary64 modulo of :86), compared in binary32,
// and the weight is a binary32 quotient of binary32 differences.  That is what the reference's own trajectory presets
// (circle_horizontal, circle_askew, spiral: apply_hrtf.py:585-586, :593) feed it.
template <bool PYF>
__device__ __forceinline__ void ring_lookup(const RingTable &R, const float *__restrict__ node_az, int ring,
                                             double az, int &before, int &after, double &a) {
    if (ring == 9) {                                         // pole: sphere.py:92-93
        before = after = 186;
        a = 0.0;
        return;
    }
    const int start = R.ring_start[ring], count = R.ring_count[ring];
    const float az32 = (float)az;                            // round to nearest even, as numpy converts the weak scalar
    // the last node <= az (sphere.py:103; node 0 is azimuth 0).  The nodes of a ring ascend (sphere.py:124-319 lists them so)
    // and are evenly spaced but for float32 rounding: the guess az / (2 pi / count) is off by at most one, and the two loops
    // make it exact under this branch's comparison for ANY ascending ring.  (Round 3 scanned the whole ring: 23 reads one
    // after the other per lookup - most of the 8.9 us this kernel took for a single source's 863 chunk boundaries.)
    int j = (int)(az * ((double)count * (1.0 / (2.0 * 3.14159265358979323846))));
    j = j < 0 ? 0 : (j > count - 1 ? count - 1 : j);
    while (j > 0 && !(PYF ? node_az[start + j] <= az32 : (double)node_az[start + j] <= az)) --j;
    while (j + 1 < count && (PYF ? node_az[start + j + 1] <= az32 : (double)node_az[start + j + 1] <= az)) ++j;
    const bool wrap = j + 1 >= count;
    const float b32 = node_az[start + j];
    const float a32 = wrap ? (float)(2.0 * 3.14159265358979323846) : node_az[start + j + 1];
    const float den = __fsub_rn(a32, b32);                   // float32 subtraction, as sphere.py:119 evaluates it
    before = start + j;
    after = wrap ? start : start + j + 1;
    if (PYF) a = (double)__fdiv_rn(__fsub_rn(az32, b32), den);
    else a = (az - (double)b32) / (double)den;
}

// one (elev, azim) -> (top_before, top_after, bot_before, bot_after), (top_alpha, bot_alpha, a)
template <bool PYF>
__device__ __forceinline__ void traj_params_one(const RingTable &R, const float *__restrict__ node_az, double e, double az,
                                                 int (&ix)[4], double (&wt)[3]) {
    const double two_pi = 2.0 * 3.14159265358979323846;
    double z = fmod(az, two_pi);                             // numpy's % : result in [0, 2 pi)
    if (z != 0.0 && z < 0.0) z += two_pi;
    int hi = 0, lo = 9;                                      // first elevation >= e, last elevation <= e
    while (hi < 9 && R.ring_elev[hi] < e) ++hi;
    while (lo > 0 && R.ring_elev[lo] > e) --lo;
    ring_lookup<PYF>(R, node_az, hi, z, ix[0], ix[1], wt[0]);
    ring_lookup<PYF>(R, node_az, lo, z, ix[2], ix[3], wt[1]);
    const double span = R.ring_elev[hi] - R.ring_elev[lo];
    wt[2] = span > 0.0 ? (e - R.ring_elev[lo]) / span : 0.0;
}

#define BAS_QB 16      // queries per workgroup

// Where the plan kernel takes a query's parameters from: the (idx, w) arrays, or straight from the trajectory angles
// (ANG = 1: float64 branch of a3, 2: the Python-float branch) - for SMALL batches, where a separate a3 launch is pure
// launch latency (one source x 10 s = 863 queries: 5 us per launch); for large ones both ears' threads redoing the
// angle arithmetic cost more than the launch (221 k queries: 53 us merged against 9 + 15 us).
struct PlanAngles {
    const double *elev, *azim;
    const float *node_az;
    RingTable R;
};

#define BAS_PLAN_WAVE_STAGED_FROM 32768   // queries from which the plan kernel stages its records per wave (see its end)

// plan kernel: one thread per (query, ear)
template <int ANG, int WAVE_STAGED = 0>
__global__ __launch_bounds__(256) void bas_interp2d_plan_kernel(const double *__restrict__ diffs,
                                                                  const int32_t *__restrict__ idx,
                                                                  const double *__restrict__ w, int n,
                                                                  int ndir, int L, int U,
                                                                  EarPlanS *__restrict__ plans, PlanAngles PA) {
    long t = blockIdx.x * 256L + threadIdx.x;
    if (t >= 2L * n) t = 2L * n - 1;                         // (idle threads of the last block redo its last record: barrier below)
    const long q = t >> 1;
    const int e = (int)(t & 1);
    const int M = L * U;
    int ix[4];
    double wt[3];
    if constexpr (ANG == 0) {
        ix[0] = idx[4 * q + 0]; ix[1] = idx[4 * q + 1]; ix[2] = idx[4 * q + 2]; ix[3] = idx[4 * q + 3];
        wt[0] = w[3 * q + 0]; wt[1] = w[3 * q + 1]; wt[2] = w[3 * q + 2];
    } else {
        // a3 for the block's 128 queries by its first two waves (one query per lane: the angle arithmetic is done once per
        // query, as in the separate kernel), handed to the four waves' (query, ear) threads through LDS.  Round 3 had every
        // (query, ear) thread redo it: 53 us for 221 k queries against 9 + 15 for two kernels; this form: both in one launch
        // with no (idx, w) round trip through HBM.
        __shared__ int s_ix[128][4];
        __shared__ double s_wt[128][3];
        if (threadIdx.x < 128) {                              // (waves 0 and 1: uniform per wave)
            long qq = blockIdx.x * 128L + threadIdx.x;
            if (qq >= n) qq = n - 1;
            int jx[4];
            double wv[3];
            traj_params_one<ANG == 2>(PA.R, PA.node_az, PA.elev[qq], PA.azim[qq], jx, wv);
#pragma unroll
            for (int i = 0; i < 4; ++i) s_ix[threadIdx.x][i] = jx[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) s_wt[threadIdx.x][i] = wv[i];
        }
        __syncthreads();
        const int ql = (int)(q - blockIdx.x * 128L);          // (the clamped tail threads of the last block: its last query)
#pragma unroll
        for (int i = 0; i < 4; ++i) ix[i] = s_ix[ql][i];
#pragma unroll
        for (int i = 0; i < 3; ++i) wt[i] = s_wt[ql][i];
    }
    const int pt = clamp_dir(ix[0], ndir), qt = clamp_dir(ix[1], ndir);
    const int pb = clamp_dir(ix[2], ndir), qb = clamp_dir(ix[3], ndir);
    const double at = wt[0], ab = wt[1], a = wt[2];
    const double *d = diffs + (long)e * ndir * ndir;
    // delays of the two ring interpolations in non-upsampled samples (apply_hrtf.py:106)
    const double dt = (at * ((double)U * d[(long)pt * ndir + qt])) / (double)U;
    const double db = (ab * ((double)U * d[(long)pb * ndir + qb])) / (double)U;
    const double dv = (double)U * (-dt + d[(long)pt * ndir + pb] + db);         // :246-252
    float f3, f4;
    const int b3 = bas_split_shift_mod(-dv, M, f3);                             // :254-255
    const int b4 = bas_split_shift_mod((1.0 - a) * dv, M, f4);                  // :272-277
    RingPlan top, bot;
    double s2;
    // C is read at -b4 - j; R_t at the same offsets; R_b at (-b4 - b3) - j
    const int c_top = bas_submod(0, b4, M);
    ring_plan(top, d, e, ndir, pt, qt, at, L, U, c_top, s2);
    ring_plan(bot, d, e, ndir, pb, qb, ab, L, U, bas_submod(c_top, b3, M), s2);
    // the four read sets: (packed row, offset c of read 0 in upsampled samples); read j of a set sits at c - j
    const int set_row[4] = {bot.row_q, bot.row_p, top.row_q, top.row_p};
    const int set_c[4] = {bot.c_q, bot.c_p, top.c_q, top.c_p};
    const int nset[4] = {5, 4, 4, 3};
    EarPlanS ps;
    int k = 0;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int o = set_c[st] / U, ph0 = set_c[st] - o * U;                 // c in [0, M)
        const int base = set_row[st] + ph0 * BAS_PLANE(L) + 1;                // float index of plane ph0's sample 0
        ps.o4[st] = 4u * (unsigned)o;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (j < nset[st]) {
                // read j <= ph0: plane ph0 - j; beyond: plane ph0 - j + U one sample earlier (guards make -1 safe)
                const int pb = j <= ph0 ? base - j * BAS_PLANE(L) : base + (U - j) * BAS_PLANE(L) - 1;
                ps.off[k] = 4u * (unsigned)pb;
                ++k;
            }
        }
    }
    // fold the blend chain into 16 weights (apply_hrtf.py:90-91, :98-99, :268-269, :276-277)
    const double cC[2] = {1.0 - (double)f4, (double)f4};
    double wrb[3], wbb[4], wbt[3];
    {
        const double s0 = (1.0 - a) * cC[0], s1 = (1.0 - a) * cC[1], g = (double)f3;
        wrb[0] = s0 * (1.0 - g); wrb[1] = s0 * g + s1 * (1.0 - g); wrb[2] = s1 * g;
        const double h = (double)bot.f2;
        wbb[0] = wrb[0] * (1.0 - h); wbb[1] = wrb[0] * h + wrb[1] * (1.0 - h);
        wbb[2] = wrb[1] * h + wrb[2] * (1.0 - h); wbb[3] = wrb[2] * h;
        const double kk = (double)bot.f1;
        ps.w[0] = (float)(ab * wbb[0] * (1.0 - kk));
        ps.w[1] = (float)(ab * (wbb[0] * kk + wbb[1] * (1.0 - kk)));
        ps.w[2] = (float)(ab * (wbb[1] * kk + wbb[2] * (1.0 - kk)));
        ps.w[3] = (float)(ab * (wbb[2] * kk + wbb[3] * (1.0 - kk)));
        ps.w[4] = (float)(ab * wbb[3] * kk);
#pragma unroll
        for (int j = 0; j < 4; ++j) ps.w[5 + j] = (float)((1.0 - ab) * wbb[j]);
    }
    {
        const double r0 = a * cC[0], r1 = a * cC[1], h = (double)top.f2;
        wbt[0] = r0 * (1.0 - h); wbt[1] = r0 * h + r1 * (1.0 - h); wbt[2] = r1 * h;
        const double kk = (double)top.f1;
        ps.w[9] = (float)(at * wbt[0] * (1.0 - kk));
        ps.w[10] = (float)(at * (wbt[0] * kk + wbt[1] * (1.0 - kk)));
        ps.w[11] = (float)(at * (wbt[1] * kk + wbt[2] * (1.0 - kk)));
        ps.w[12] = (float)(at * wbt[2] * kk);
#pragma unroll
        for (int j = 0; j < 3; ++j) ps.w[13 + j] = (float)((1.0 - at) * wbt[j]);
    }
    // 144-byte records written lane by lane would touch every 64-byte segment two or three times: the block's records go
    // through LDS (stride 37 words: conflict-free) and leave as one contiguous run of 16-byte stores
    // 144-byte records written lane by lane would touch every 64-byte segment two or three times: they go through LDS
    // (stride 37 words: conflict-free) and leave as contiguous runs of 16-byte stores.
    // WAVE_STAGED = 0 (small batches, where one wave's chain of dependent round trips IS the kernel's time): the block's
    // 256 records in one staging area (37.9 KB: three workgroups per CU), one barrier.
    // WAVE_STAGED = 1 (large batches): the staging area sets how many workgroups a CU holds, and a kernel made of such
    // chains wants all the waves its 63 registers allow - each WAVE stages its 64 records in two passes through a region
    // of its own and writes them out itself (its records are one contiguous run of the output; no workgroup barrier):
    // 18.9 KB per workgroup, six per CU: 21.2 -> 18.3 us for 221 k queries (four passes, eight per CU: the same; for a
    // single source's 863 queries the passes lengthen the chain: 6.2 -> 7.6 us, which is why both forms exist).
    if constexpr (WAVE_STAGED == 0) {
        __shared__ unsigned stage[256 * (BAS_PLANS_WORDS + 1)];
        {
            const unsigned *src = reinterpret_cast<const unsigned *>(&ps);
            unsigned *row = stage + threadIdx.x * (BAS_PLANS_WORDS + 1);
#pragma unroll
            for (int i = 0; i < BAS_PLANS_WORDS; ++i) row[i] = src[i];
        }
        __syncthreads();
        const long t0 = blockIdx.x * 256L;
        const long live = 2L * n - t0 < 256 ? 2L * n - t0 : 256;            // records of this block
        u32x4 *dst = reinterpret_cast<u32x4 *>(plans + t0);
        for (int i = threadIdx.x; i < live * (BAS_PLANS_WORDS / 4); i += 256) {
            const int r = i / (BAS_PLANS_WORDS / 4), c = i - r * (BAS_PLANS_WORDS / 4);
            const unsigned *q = stage + r * (BAS_PLANS_WORDS + 1) + 4 * c;
            dst[i] = u32x4{q[0], q[1], q[2], q[3]};
        }
        return;
    }
    constexpr int PLAN_PASSES = 2, PER_PASS = 64 / PLAN_PASSES, Q = BAS_PLANS_WORDS / 4;
    __shared__ unsigned wstage[4 * PER_PASS * (BAS_PLANS_WORDS + 1)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned *region = wstage + wave * PER_PASS * (BAS_PLANS_WORDS + 1);
    const long w0 = blockIdx.x * 256L + 64L * wave;                      // first record of this wave
    long live_w = 2L * n - w0;                                           // records of this wave that exist
    live_w = live_w < 0 ? 0 : (live_w > 64 ? 64 : live_w);
    unsigned rec[BAS_PLANS_WORDS];                                       // the record as words (EarPlanS: off, w, o4), all in registers
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        rec[i] = ps.off[i];
        rec[16 + i] = __float_as_uint(ps.w[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) rec[32 + i] = ps.o4[i];
#pragma unroll
    for (int pass = 0; pass < PLAN_PASSES; ++pass) {
        if (lane / PER_PASS == pass) {
            unsigned *row = region + (lane - pass * PER_PASS) * (BAS_PLANS_WORDS + 1);
#pragma unroll
            for (int i = 0; i < BAS_PLANS_WORDS; ++i) row[i] = rec[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");           // (a wave's LDS accesses are in order; the compiler must keep them so)
        __builtin_amdgcn_wave_barrier();
        long live_p = live_w - pass * PER_PASS;                          // records of this pass that exist
        live_p = live_p < 0 ? 0 : (live_p > PER_PASS ? PER_PASS : live_p);
        u32x4 *dst = reinterpret_cast<u32x4 *>(plans + w0 + pass * PER_PASS);
        for (int i = lane; i < live_p * Q; i += 64) {
            const int r = i / Q, c = i - r * Q;
            const unsigned *q = region + r * (BAS_PLANS_WORDS + 1) + 4 * c;
            dst[i] = u32x4{q[0], q[1], q[2], q[3]};
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// eval kernel: a wave per query (both ears), grid-stride.  The wave copies the query's two plans (288 bytes) into its
// own LDS region; lanes 0-31 then evaluate four adjacent taps of the left ear, lanes 32-63 of the right: every plan
// value reaches its half-wave as a broadcast LDS read (bas_plan.h).  Sixteen 16-byte table reads per lane in two halves
// of 8, the second in flight while the first is folded; the next query's plans are requested before this one is evaluated.
__global__ __launch_bounds__(256) void bas_interp2d_eval_kernel(const float *__restrict__ packed, unsigned packed_bytes,
                                                                  const EarPlanS *__restrict__ plans, long n_queries,
                                                                  int L, float *__restrict__ H) {
    constexpr int PL4 = 2 * BAS_PLANS_WORDS / 4;             // float4 per query (18)
    __shared__ f32x4 pl_lds[4][PL4];
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int wv = rfl((int)(threadIdx.x >> 6));
    const __amdgpu_buffer_rsrc_t tab =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(packed), 0, (int)packed_bytes, 0x00020000);
    const unsigned L4 = 4u * (unsigned)L;
    const f32x4 *pl_src = reinterpret_cast<const f32x4 *>(plans);
    const long stride = gridDim.x * 4L;
    long q = blockIdx.x * 4L + wv;
    const int piece = lane < PL4 ? lane : 0;
    f32x4 next = q < n_queries ? pl_src[q * PL4 + piece] : f32x4{0.f, 0.f, 0.f, 0.f};
    for (; q < n_queries; q += stride) {
        if (lane < PL4) pl_lds[wv][lane] = next;
        __builtin_amdgcn_wave_barrier();                     // the other lanes of this wave read these words below
        const long qn = q + stride < n_queries ? q + stride : q;
        next = pl_src[qn * PL4 + piece];
        const f32x4 *pl = pl_lds[wv] + half * (BAS_PLANS_WORDS / 4);
        float *Hrow = H + (2 * q + half) * L;
        for (int m0 = 0; m0 < L; m0 += 128) {                // 32 lanes x 4 taps per sweep
            const int m = m0 + 4 * (lane & 31);
            const int m_c = m < L ? m : L - 1;               // idle lanes evaluate a valid tap and drop it
            const unsigned m4 = 4u * (unsigned)m_c;
            FzHalf ha, hb;
            fz_issue<0>(tab, pl, m4, L4, ha);
            fz_issue<1>(tab, pl, m4, L4, hb);
            f32x4 acc = fz_finish<0>(pl, ha, f32x4{0.f, 0.f, 0.f, 0.f});
            acc = fz_finish<1>(pl, hb, acc);
            if (m + 3 < L) *reinterpret_cast<f32x4_a4 *>(Hrow + m) = acc;
            else {
                if (m < L) Hrow[m] = acc.x;
                if (m + 1 < L) Hrow[m + 1] = acc.y;
                if (m + 2 < L) Hrow[m + 2] = acc.z;
            }
        }
        __builtin_amdgcn_wave_barrier();                     // (the next iteration overwrites the plan words)
    }
}

// Any upsampling factor: the planned evaluation above steps through up to five consecutive upsampled
// positions with "previous plane, or plane + U one sample earlier" (bas_plan.h), which needs U >= 4.  Tables
// with U = 1, 2, 3 (the reference accepts any factor, apply_hrtf.py:38) take this plain form instead: one
// thread per output tap, every table sample addressed through tab_read's general position arithmetic.
__global__ __launch_bounds__(256) void bas_interp2d_generic_kernel(const float *__restrict__ packed,
                                                                     const double *__restrict__ diffs,
                                                                     const int32_t *__restrict__ idx,
                                                                     const double *__restrict__ w, long n,
                                                                     int ndir, int L, int U,
                                                                     float *__restrict__ H) {
    const int M = L * U;
    const long total = n * 2 * L;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        const long qe = i / L;
        const int m = (int)(i - qe * L);
        const long q = qe >> 1;
        const int e = (int)(qe & 1);
        const int pt = clamp_dir(idx[4 * q + 0], ndir), qt = clamp_dir(idx[4 * q + 1], ndir);
        const int pb = clamp_dir(idx[4 * q + 2], ndir), qb = clamp_dir(idx[4 * q + 3], ndir);
        const double at = w[3 * q + 0], ab = w[3 * q + 1], a = w[3 * q + 2];
        const double *d = diffs + (long)e * ndir * ndir;
        const double dt = (at * ((double)U * d[(long)pt * ndir + qt])) / (double)U;   // apply_hrtf.py:106
        const double db = (ab * ((double)U * d[(long)pb * ndir + qb])) / (double)U;
        const double dv = (double)U * (-dt + d[(long)pt * ndir + pb] + db);           // :246-252
        float f3, f4;
        const int b3 = bas_split_shift_mod(-dv, M, f3);                               // :254-255
        const int b4 = bas_split_shift_mod((1.0 - a) * dv, M, f4);                    // :272-277
        RingPlan top, bot;
        double s2;
        const int c_top = bas_submod(0, b4, M);
        ring_plan(top, d, e, ndir, pt, qt, at, L, U, c_top, s2);
        ring_plan(bot, d, e, ndir, pb, qb, ab, L, U, bas_submod(c_top, b3, M), s2);
        float rb[3], rt[2];
        ring_eval<false, 3>(packed, bot, m, L, U, 0, M, rb);
        ring_eval<false, 2>(packed, top, m, L, U, 0, M, rt);
        const float af = (float)a;
        float c[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float sb = (1.0f - f3) * rb[j] + f3 * rb[j + 1];                    // S(R_b, -Dv)
            c[j] = (1.0f - af) * sb + af * rt[j];                                      // :268-269
        }
        H[i] = (1.0f - f4) * c[0] + f4 * c[1];                                         // :276-277
    }
}

static_assert(sizeof(EarPlanS) == 4 * BAS_PLANS_WORDS && sizeof(EarPlanS) % 16 == 0, "a plan = 9 pieces of 16 bytes");

extern "C" size_t bas_interp2d_workspace_bytes(int n) {
    return n > 0 ? (size_t)n * 2 * sizeof(EarPlanS) + 16 : 16;
}

extern "C" int bas_interp2d_plan_f32(const double *diffs, const int32_t *idx, const double *w, int n, int ndir,
                                     int L, int U, void *plans, size_t plans_bytes, bas_stream_t stream) {
    BAS_REQUIRE(diffs && ((idx && w) || n == 0), BAS_E_NULL, "bas_interp2d_plan_f32: null pointer");   // (no queries: idx, w may be null)
    BAS_REQUIRE(n >= 0 && ndir > 0 && L > 0 && U > 0, BAS_E_SHAPE,
                "bas_interp2d_plan_f32: need n>=0, ndir>0, L>0, U>0 (n=%d ndir=%d L=%d U=%d)", n, ndir, L, U);
    BAS_REQUIRE((long)2 * ndir * BAS_PLANE(L) * U < (1L << 31), BAS_E_SHAPE, "bas_interp2d_plan_f32: table too large");
    BAS_REQUIRE(U >= BAS_PLAN_MIN_U, BAS_E_SHAPE,
                "bas_interp2d_plan_f32: read plans need an upsampling factor >= %d (U=%d): use bas_interp2d_f32",
                BAS_PLAN_MIN_U, U);
    if (n == 0) return 0;
    BAS_REQUIRE(plans && plans_bytes >= bas_interp2d_workspace_bytes(n) &&
                    reinterpret_cast<uintptr_t>(plans) % 16 == 0,
                BAS_E_WORKSPACE, "bas_interp2d_plan_f32: 16-byte aligned buffer of %zu bytes needed, %zu given",
                bas_interp2d_workspace_bytes(n), plans_bytes);
    const long rows = 2L * n;
    if (n >= BAS_PLAN_WAVE_STAGED_FROM)
        hipLaunchKernelGGL((bas_interp2d_plan_kernel<0, 1>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                           bas_stream(stream), diffs, idx, w, n, ndir, L, U, reinterpret_cast<EarPlanS *>(plans), PlanAngles{});
    else
        hipLaunchKernelGGL((bas_interp2d_plan_kernel<0, 0>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                           bas_stream(stream), diffs, idx, w, n, ndir, L, U, reinterpret_cast<EarPlanS *>(plans), PlanAngles{});
    return bas_check_launch("bas_interp2d_plan_f32");
}

extern "C" int bas_interp2d_plan_angles_f32(const double *diffs, const double *elev, const double *azim, int n,
                                            const double *ring_elev, const int32_t *ring_start,
                                            const int32_t *ring_count, const float *node_az, int branch, int ndir,
                                            int L, int U, void *plans, size_t plans_bytes, bas_stream_t stream) {
    BAS_REQUIRE(diffs && ring_elev && ring_start && ring_count && node_az && ((elev && azim) || n == 0), BAS_E_NULL,
                "bas_interp2d_plan_angles_f32: null pointer");
    BAS_REQUIRE(n >= 0 && ndir > 0 && L > 0 && U >= BAS_PLAN_MIN_U, BAS_E_SHAPE,
                "bas_interp2d_plan_angles_f32: need n>=0, ndir>0, L>0, U>=%d (n=%d ndir=%d L=%d U=%d)", BAS_PLAN_MIN_U, n,
                ndir, L, U);
    BAS_REQUIRE(branch == BAS_BRANCH_F64 || branch == BAS_BRANCH_PYFLOAT, BAS_E_SHAPE,
                "bas_interp2d_plan_angles_f32: branch must be BAS_BRANCH_F64 (0) or BAS_BRANCH_PYFLOAT (1), got %d", branch);
    BAS_REQUIRE((long)2 * ndir * BAS_PLANE(L) * U < (1L << 31), BAS_E_SHAPE, "bas_interp2d_plan_angles_f32: table too large");
    if (n == 0) return 0;
    BAS_REQUIRE(plans && plans_bytes >= bas_interp2d_workspace_bytes(n) && reinterpret_cast<uintptr_t>(plans) % 16 == 0,
                BAS_E_WORKSPACE, "bas_interp2d_plan_angles_f32: 16-byte aligned buffer of %zu bytes needed, %zu given",
                bas_interp2d_workspace_bytes(n), plans_bytes);
    PlanAngles PA;
    PA.elev = elev; PA.azim = azim; PA.node_az = node_az;
    for (int i = 0; i < 10; ++i) {
        PA.R.ring_elev[i] = ring_elev[i];
        PA.R.ring_start[i] = ring_start[i];
        PA.R.ring_count[i] = ring_count[i];
        BAS_REQUIRE(PA.R.ring_count[i] > 0 && PA.R.ring_start[i] >= 0 && PA.R.ring_start[i] + PA.R.ring_count[i] <= ndir,
                    BAS_E_SHAPE, "bas_interp2d_plan_angles_f32: ring %d out of the %d-direction table", i, ndir);
    }
    const long rows = 2L * n;
    typedef void (*plan_fn)(const double *, const int32_t *, const double *, int, int, int, int, EarPlanS *, PlanAngles);
    const bool big = n >= BAS_PLAN_WAVE_STAGED_FROM;
    const plan_fn fn = branch == BAS_BRANCH_PYFLOAT ? (big ? bas_interp2d_plan_kernel<2, 1> : bas_interp2d_plan_kernel<2, 0>)
                                                    : (big ? bas_interp2d_plan_kernel<1, 1> : bas_interp2d_plan_kernel<1, 0>);
    hipLaunchKernelGGL(fn, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, bas_stream(stream), diffs, nullptr, nullptr, n,
                       ndir, L, U, reinterpret_cast<EarPlanS *>(plans), PA);
    return bas_check_launch("bas_interp2d_plan_angles_f32");
}

extern "C" int bas_interp2d_f32(const float *packed, const double *diffs, const int32_t *idx,
                                const double *w, int n, int ndir, int L, int U, float *H, void *ws,
                                size_t ws_bytes, bas_stream_t stream) {
    BAS_REQUIRE(packed && diffs && ((idx && w && H) || n == 0), BAS_E_NULL, "bas_interp2d_f32: null pointer");   // (no queries: idx, w, H may be null)
    BAS_REQUIRE(n >= 0 && ndir > 0 && L > 0 && U > 0, BAS_E_SHAPE,
                "bas_interp2d_f32: need n>=0, ndir>0, L>0, U>0 (n=%d ndir=%d L=%d U=%d)", n, ndir, L, U);
    BAS_REQUIRE((long)2 * ndir * BAS_PLANE(L) * U < (1L << 31), BAS_E_SHAPE, "bas_interp2d_f32: table too large");
    if (n == 0) return 0;
    BAS_REQUIRE(ws && ws_bytes >= bas_interp2d_workspace_bytes(n) && reinterpret_cast<uintptr_t>(ws) % 16 == 0,
                BAS_E_WORKSPACE, "bas_interp2d_f32: 16-byte aligned workspace of %zu bytes needed, %zu given",
                bas_interp2d_workspace_bytes(n), ws_bytes);
    EarPlanS *plans = reinterpret_cast<EarPlanS *>(ws);
    hipStream_t st = bas_stream(stream);
    const long rows = 2L * n;
    if (U < BAS_PLAN_MIN_U) {                                // small upsampling factors: plain evaluation
        long blocks = ((long)n * 2 * L + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(bas_interp2d_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, st, packed, diffs, idx,
                           w, (long)n, ndir, L, U, H);
        return bas_check_launch("bas_interp2d_f32(generic)");
    }
    if (n >= BAS_PLAN_WAVE_STAGED_FROM)
        hipLaunchKernelGGL((bas_interp2d_plan_kernel<0, 1>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, diffs, idx,
                           w, n, ndir, L, U, plans, PlanAngles{});
    else
        hipLaunchKernelGGL((bas_interp2d_plan_kernel<0, 0>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, diffs, idx,
                           w, n, ndir, L, U, plans, PlanAngles{});
    int rc = bas_check_launch("bas_interp2d_f32(plan)");
    if (rc) return rc;
    long blocks = ((long)n + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    const unsigned table_bytes = (unsigned)((size_t)2 * ndir * U * BAS_PLANE(L) * sizeof(float));
    hipLaunchKernelGGL(bas_interp2d_eval_kernel, dim3((unsigned)blocks), dim3(256), 0, st, packed, table_bytes, plans,
                       (long)n, L, H);
    return bas_check_launch("bas_interp2d_f32(eval)");
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
        ring_plan(rp, diffs + (long)e * ndir * ndir, e, ndir, p, r, alpha[q], L, U, j * step, s2);
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
    BAS_REQUIRE((long)2 * ndir * BAS_PLANE(L) * U < (1L << 31), BAS_E_SHAPE, "bas_ring_interp_f32: table too large");
    if (n == 0) return 0;
    int step = return_upsampled ? 1 : U;
    int Mout = return_upsampled ? L * U : L;
    long total = (long)n * 2 * Mout;
    int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(bas_ring_kernel, dim3(grid), dim3(256), 0, bas_stream(stream), packed, diffs, pq,
                       alpha, n, ndir, L, U, step, Mout, out, delays);
    return bas_check_launch("bas_ring_interp_f32");
}

template <bool PYF>
__global__ __launch_bounds__(256) void bas_traj_params_kernel(const double *__restrict__ elev,
                                                                const double *__restrict__ azim, long n,
                                                                RingTable R, const float *__restrict__ node_az,
                                                                int32_t *__restrict__ idx,
                                                                double *__restrict__ w) {
    // (the body of traj_params_one, written out: through the helper's array references hipcc copies the by-value ring
    // table into scratch - 168 bytes per lane, 19 us instead of 9 for 221 k boundaries)
    const double two_pi = 2.0 * 3.14159265358979323846;
    for (long q = blockIdx.x * 256L + threadIdx.x; q < n; q += (long)gridDim.x * 256L) {
        const double e = elev[q];
        double z = fmod(azim[q], two_pi);                    // numpy's % : result in [0, 2 pi)
        if (z != 0.0 && z < 0.0) z += two_pi;
        int hi = 0, lo = 9;                                  // first elevation >= e, last elevation <= e
        while (hi < 9 && R.ring_elev[hi] < e) ++hi;
        while (lo > 0 && R.ring_elev[lo] > e) --lo;
        int tb, taf, bb, baf;
        double ta, ba;
        ring_lookup<PYF>(R, node_az, hi, z, tb, taf, ta);
        ring_lookup<PYF>(R, node_az, lo, z, bb, baf, ba);
        const double span = R.ring_elev[hi] - R.ring_elev[lo];
        const double a = span > 0.0 ? (e - R.ring_elev[lo]) / span : 0.0;
        idx[4 * q + 0] = tb; idx[4 * q + 1] = taf; idx[4 * q + 2] = bb; idx[4 * q + 3] = baf;
        w[3 * q + 0] = ta; w[3 * q + 1] = ba; w[3 * q + 2] = a;
    }
}

static int traj_params_launch(const char *who, const double *elev, const double *azim, long n, const double *ring_elev,
                              const int32_t *ring_start, const int32_t *ring_count, const float *node_az,
                              int32_t *idx, double *w, int branch, bas_stream_t stream) {
    BAS_REQUIRE(ring_elev && ring_start && ring_count && node_az, BAS_E_NULL, "%s: null ring table", who);
    BAS_REQUIRE(n >= 0, BAS_E_SHAPE, "%s: n < 0", who);
    BAS_REQUIRE(branch == BAS_BRANCH_F64 || branch == BAS_BRANCH_PYFLOAT, BAS_E_SHAPE,
                "%s: branch must be BAS_BRANCH_F64 (0) or BAS_BRANCH_PYFLOAT (1), got %d", who, branch);
    if (n == 0) return 0;
    BAS_REQUIRE(elev && azim && idx && w, BAS_E_NULL, "%s: null pointer", who);
    RingTable R;
    for (int i = 0; i < 10; ++i) {
        R.ring_elev[i] = ring_elev[i];
        R.ring_start[i] = ring_start[i];
        R.ring_count[i] = ring_count[i];
        BAS_REQUIRE(R.ring_count[i] > 0 && R.ring_start[i] >= 0 && R.ring_start[i] + R.ring_count[i] <= 187,
                    BAS_E_SHAPE, "%s: ring %d out of the 187-direction table", who, i);
    }
    long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (branch == BAS_BRANCH_PYFLOAT)
        hipLaunchKernelGGL(bas_traj_params_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, bas_stream(stream), elev,
                           azim, n, R, node_az, idx, w);
    else
        hipLaunchKernelGGL(bas_traj_params_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, bas_stream(stream), elev,
                           azim, n, R, node_az, idx, w);
    return bas_check_launch(who);
}

extern "C" int bas_traj_params_f64(const double *elev, const double *azim, long n, const double *ring_elev,
                                   const int32_t *ring_start, const int32_t *ring_count, const float *node_az,
                                   int32_t *idx, double *w, bas_stream_t stream) {
    return traj_params_launch("bas_traj_params_f64", elev, azim, n, ring_elev, ring_start, ring_count, node_az, idx, w,
                              BAS_BRANCH_F64, stream);
}

extern "C" int bas_traj_params_branch_f64(const double *elev, const double *azim, long n, const double *ring_elev,
                                          const int32_t *ring_start, const int32_t *ring_count, const float *node_az,
                                          int32_t *idx, double *w, int branch, bas_stream_t stream) {
    return traj_params_launch("bas_traj_params_branch_f64", elev, azim, n, ring_elev, ring_start, ring_count, node_az,
                              idx, w, branch, stream);
}
