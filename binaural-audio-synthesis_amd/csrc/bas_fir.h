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

