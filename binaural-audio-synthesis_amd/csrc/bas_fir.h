// Row step of the output-stationary FIR kernels (bas_render.hip: bas_render_hd_kernel; bas_fused.hip:
// bas_render_fz_kernel): one lane owns a row of 32 consecutive outputs (both ears: 64 accumulators) and meets
// one row of 32 inputs with the 64 taps that connect the two rows - a 32 x 32 Toeplitz block of packed FMAs
// (apply_hrtf.py:445-446 is the direct FIR these sums restate; :442-443 the crossfade g = h0 + al d formed here).
#pragma once
#include "bas_internal.h"

#define RT_SEG 128                            // taps per LDS pass
#define HD_HALO (RT_SEG / 32)                 // input rows above a tile (4)
#define HD_SLOT (RT_SEG * 4 + 4)              // floats per chunk slot (+16 B: slots on distinct banks)

__device__ __forceinline__ void fma2(f32x2 &acc, float xv, f32x2 g) {
    acc = __builtin_elementwise_fma(g, f32x2{xv, xv}, acc);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

#define HO_SLOT (RT_SEG * 2 + 2)             // floats per chunk slot of the h-only image (+8 B: slots on distinct banks)

// HONLY: the slot holds (h_L, h_R) only and d = H_{c+1} - H_c is taken here from the next slot (small chunks:
// twice as many chunk slots fit the LDS; one more packed op per tap)
template <bool HONLY>
__device__ __forceinline__ void hd_load_octet(f32x4 (&hv)[8], const float *__restrict__ hdrow, int i) {
    if (HONLY) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x2 h0 = *reinterpret_cast<const f32x2 *>(hdrow + (8 * i + j) * 2);
            const f32x2 h1 = *reinterpret_cast<const f32x2 *>(hdrow + HO_SLOT + (8 * i + j) * 2);
            const f32x2 d = h1 - h0;
            hv[j] = f32x4{h0.x, h0.y, d.x, d.y};
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[j] = *reinterpret_cast<const f32x4 *>(hdrow + (8 * i + j) * 4);
    }
}

// one octet of taps (delta = 8 i + j - 32 = output index - input index) against the whole input row.
// NSUB > 1: the subchunk is shorter than a row (S = 32 / NSUB), so the row's 32 inputs fall into NSUB
// groups with their own crossfade weight al[u]; each group gets its own formed taps.
template <int I, int NSUB>
__device__ __forceinline__ void hd_octet_fma(f32x2 (&acc)[32], const float (&xr)[32], const f32x4 (&hv)[8],
                                              const float (&al)[NSUB]) {
    constexpr int G = 32 / NSUB;                 // inputs per group (= the subchunk size when NSUB > 1)
    if constexpr (NSUB <= 4) {                   // few groups: all their taps formed up front
        f32x2 g[NSUB][8];
#pragma unroll
        for (int u = 0; u < NSUB; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                g[u][j] = __builtin_elementwise_fma(f32x2{hv[j].z, hv[j].w}, f32x2{al[u], al[u]},
                                                    f32x2{hv[j].x, hv[j].y});
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int delta = 8 * I + j - 32;
#pragma unroll
            for (int o = 0; o < 32; ++o) {
                const int a = o - delta;
                if (a >= 0 && a < 32) fma2(acc[o], xr[a], g[a / G][j]);
            }
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {             // many groups: one at a time, eight formed taps live
        f32x2 g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            g[j] = __builtin_elementwise_fma(f32x2{hv[j].z, hv[j].w}, f32x2{al[u], al[u]}, f32x2{hv[j].x, hv[j].y});
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int delta = 8 * I + j - 32;
#pragma unroll
            for (int o = 0; o < 32; ++o) {
                const int a = o - delta;
                if (a >= u * G && a < (u + 1) * G && a >= 0 && a < 32) fma2(acc[o], xr[a], g[j]);
            }
        }
    }
}

// Row step on an x row already in registers, with a run-time set of live octets.
template <int NSUB, bool HONLY>
__device__ __forceinline__ void hd_row_step_x(f32x2 (&acc)[32], const float (&xr)[32], const float *__restrict__ hdrow,
                                               const float (&al)[NSUB], unsigned live_mask) {
    f32x4 hv[8];
#define HD_MASKED_OCTET(I)                          \
    if (live_mask & (1u << I)) {                    \
        hd_load_octet<HONLY>(hv, hdrow, I);         \
        hd_octet_fma<I, NSUB>(acc, xr, hv, al);     \
    }
    HD_MASKED_OCTET(0) HD_MASKED_OCTET(1) HD_MASKED_OCTET(2) HD_MASKED_OCTET(3)
    HD_MASKED_OCTET(4) HD_MASKED_OCTET(5) HD_MASKED_OCTET(6) HD_MASKED_OCTET(7)
#undef HD_MASKED_OCTET
}

// ---------------------------------------------------------------------------
// The same row step as a 2-parallel fast FIR (polyphase split of outputs, inputs and taps into even / odd):
//   y[2p]   = A[p] + B[p-1]              A = g_e * x_e,  B = g_o * x_o,  P = (g_e + g_o) * (x_e + x_o)
//   y[2p+1] = P[p] - A[p] - B[p]         (three half-rate products instead of four: 784 packed FMAs per 32 x 32
//                                          block instead of 1024, + 32 adds forming g_e + g_o, + 16 forming x_e + x_o)
// A, B (17 entries: p = -1 .. 15) and P are accumulated over all row steps, sources and tap segments of a tile and
// only combined when the tile is flushed (ffa_combine).  Every product is linear in g, so the per-row crossfade
// g = h0 + al d is formed per octet exactly as in the direct form.  Rounding differs from the direct sum by a few
// 1e-8 of the output's norm (tests: 1e-5).
__device__ __forceinline__ void ffa_zero(f32x2 (&fa)[16], f32x2 (&fb)[17], f32x2 (&fp)[16]) {
#pragma unroll
    for (int p = 0; p < 16; ++p) fa[p] = fp[p] = f32x2{0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 17; ++p) fb[p] = f32x2{0.f, 0.f};
}

__device__ __forceinline__ void ffa_combine(f32x2 (&out)[32], const f32x2 (&fa)[16], const f32x2 (&fb)[17],
                                             const f32x2 (&fp)[16]) {
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        out[2 * p] = fa[p] + fb[p];                          // fb[p] holds B[p-1]
        out[2 * p + 1] = (fp[p] - fa[p]) - fb[p + 1];
    }
}

// one octet of taps: full-rate taps 8 I + j - 32 (relative to the row distance) = half-rate taps dk = 4 I + jj - 16
// of g_e (j even) and g_o (j odd)
template <int I>
__device__ __forceinline__ void ffa_octet_fma(f32x2 (&fa)[16], f32x2 (&fb)[17], f32x2 (&fp)[16], const float (&xr)[32],
                                               const float (&xs)[16], const f32x4 (&hv)[8], float al) {
    f32x2 ge[4], go[4], gs[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        ge[jj] = __builtin_elementwise_fma(f32x2{hv[2 * jj].z, hv[2 * jj].w}, f32x2{al, al},
                                           f32x2{hv[2 * jj].x, hv[2 * jj].y});
        go[jj] = __builtin_elementwise_fma(f32x2{hv[2 * jj + 1].z, hv[2 * jj + 1].w}, f32x2{al, al},
                                           f32x2{hv[2 * jj + 1].x, hv[2 * jj + 1].y});
        gs[jj] = ge[jj] + go[jj];
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int dk = 4 * I + jj - 16;                      // p - q
#pragma unroll
        for (int p = -1; p < 16; ++p) {
            const int q = p - dk;
            if (q >= 0 && q < 16) {
                if (p >= 0) {
                    fma2(fa[p], xr[2 * q], ge[jj]);
                    fma2(fp[p], xs[q], gs[jj]);
                }
                fma2(fb[p + 1], xr[2 * q + 1], go[jj]);
            }
        }
    }
}

template <bool HONLY = false>
__device__ __forceinline__ void ffa_row_step_x(f32x2 (&fa)[16], f32x2 (&fb)[17], f32x2 (&fp)[16], const float (&xr)[32],
                                                const float *__restrict__ hdrow, float al, unsigned live_mask) {
    float xs[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) xs[q] = xr[2 * q] + xr[2 * q + 1];
    f32x4 hv[8];
#define FFA_MASKED_OCTET(I)                           \
    if (live_mask & (1u << I)) {                      \
        hd_load_octet<HONLY>(hv, hdrow, I);           \
        ffa_octet_fma<I>(fa, fb, fp, xr, xs, hv, al); \
    }
    FFA_MASKED_OCTET(0) FFA_MASKED_OCTET(1) FFA_MASKED_OCTET(2) FFA_MASKED_OCTET(3)
    FFA_MASKED_OCTET(4) FFA_MASKED_OCTET(5) FFA_MASKED_OCTET(6) FFA_MASKED_OCTET(7)
#undef FFA_MASKED_OCTET
}

// The fast form for rows that hold NSUB = 2 subchunks (S = 16): inputs 0-15 of the row meet the taps formed with al[0],
// inputs 16-31 those formed with al[1]; x_e[q], x_o[q] and their sum all lie in group q / 8.
template <int I>
__device__ __forceinline__ void ffa2_octet_fma(f32x2 (&fa)[16], f32x2 (&fb)[17], f32x2 (&fp)[16], const float (&xr)[32],
                                                const float (&xs)[16], const f32x4 (&hv)[8], const float (&al)[2]) {
    f32x2 ge[2][4], go[2][4], gs[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            ge[u][jj] = __builtin_elementwise_fma(f32x2{hv[2 * jj].z, hv[2 * jj].w}, f32x2{al[u], al[u]},
                                                  f32x2{hv[2 * jj].x, hv[2 * jj].y});
            go[u][jj] = __builtin_elementwise_fma(f32x2{hv[2 * jj + 1].z, hv[2 * jj + 1].w}, f32x2{al[u], al[u]},
                                                  f32x2{hv[2 * jj + 1].x, hv[2 * jj + 1].y});
            gs[u][jj] = ge[u][jj] + go[u][jj];
        }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int dk = 4 * I + jj - 16;
#pragma unroll
        for (int p = -1; p < 16; ++p) {
            const int q = p - dk;
            if (q >= 0 && q < 16) {
                const int u = q >> 3;
                if (p >= 0) {
                    fma2(fa[p], xr[2 * q], ge[u][jj]);
                    fma2(fp[p], xs[q], gs[u][jj]);
                }
                fma2(fb[p + 1], xr[2 * q + 1], go[u][jj]);
            }
        }
    }
}

template <bool HONLY = false>
__device__ __forceinline__ void ffa2_row_step_x(f32x2 (&fa)[16], f32x2 (&fb)[17], f32x2 (&fp)[16], const float (&xr)[32],
                                                 const float *__restrict__ hdrow, const float (&al)[2], unsigned live_mask) {
    float xs[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) xs[q] = xr[2 * q] + xr[2 * q + 1];
    f32x4 hv[8];
#define FFA2_MASKED_OCTET(I)                           \
    if (live_mask & (1u << I)) {                       \
        hd_load_octet<HONLY>(hv, hdrow, I);            \
        ffa2_octet_fma<I>(fa, fb, fp, xr, xs, hv, al); \
    }
    FFA2_MASKED_OCTET(0) FFA2_MASKED_OCTET(1) FFA2_MASKED_OCTET(2) FFA2_MASKED_OCTET(3)
    FFA2_MASKED_OCTET(4) FFA2_MASKED_OCTET(5) FFA2_MASKED_OCTET(6) FFA2_MASKED_OCTET(7)
#undef FFA2_MASKED_OCTET
}
