// Read plan of one (query, ear) of interpolate_2d (apply_hrtf.py:219-279) and its evaluator, shared by the plan /
// eval kernels (bas_interp.hip) and the fused FIR kernel's staging (bas_fused.hip).
//
// h[m] = sum_k w_k * sample_k(m): 16 table reads in 4 sets (reads 0-4, 5-8, 9-12, 13-15 = 5 samples of the bottom
// ring's T_q, 4 of its T_p, 4 of the top ring's T_q, 3 of its T_p).  Within a set the reads step through consecutive
// upsampled positions, i.e. through phase planes: "previous plane, or plane + U one sample earlier".  The plan kernel
// resolves that into one byte offset per read, so the evaluator has no per-read selection work left.
#pragma once
#include "bas_internal.h"

// Floats per phase plane of the packed table: [1 front guard = last sample][L samples][3 back guards =
// first three samples], so a lane can read up to 4 consecutive taps and "one sample earlier" unconditionally.
// BAS_PLANE_DOUBLE = 1 (round 4, measured, not shipped): [guard][L samples][the L samples AGAIN][3 guards] - a lane then reads
// at (tap + circular offset) without the wrap test, three vector instructions per read set less in the stagers (16 of a chunk
// IR's ~75).  Bit-identical; FIR kernel -0.2 .. -1.1 % in the A/B, 0 in the next collection, and 14 MB more table traffic per
// launch from the 3.1 MB table's L2 misses (profiles/r04_ab_double_planes.txt): not worth its bytes.
#ifndef BAS_PLANE_DOUBLE
#define BAS_PLANE_DOUBLE 0
#endif
#if BAS_PLANE_DOUBLE
#define BAS_PLANE(L) (2 * (L) + 4)
#else
#define BAS_PLANE(L) ((L) + 4)
#endif

// A set's five reads need j - ph0 <= U for every j <= 4: upsampling factors below 4 use bas_interp2d_f32's plain kernel.
#define BAS_PLAN_MIN_U 4

#define BAS_PLANS_WORDS 36
struct EarPlanS {
    unsigned off[16];  // byte offset into `packed` of read k's plane sample 0 (the -1 of a wrapped phase included)
    float w[16];       // folded blend weights, same order
    unsigned o4[4];    // 4 * o of the four read sets (bytes): lane offset = wrap(4 m + o4)
};

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Evaluator: taps m .. m+3 of one ear.  `pl` points at the plan as 9 x 16 bytes (off[16], w[16], o4[4]) in LDS, where
// every lane of a half-wave reads the same addresses (broadcast reads).  The work is cut in halves of 8 reads so
// that a caller can keep the loads of the next half in flight while it folds the current one (the compiler, left
// alone, serialises the loads once registers get tight).
struct FzHalf {
    u32x4 v[8];
};

__device__ __forceinline__ unsigned fz_wrap(unsigned m4, unsigned o4, unsigned L4) {
#if BAS_PLANE_DOUBLE
    return m4 + o4;                                            // (the plane holds its samples twice: m + o < 2 L needs no wrap)
#endif
    const unsigned idx = m4 + o4;                              // 4 (m + o): m < L, o < L
    const unsigned wr = idx - L4;
    return idx < wr ? idx : wr;                                // idx >= 4 L ? idx - 4 L : idx
}

template <int HSEL>
__device__ __forceinline__ void fz_issue(__amdgpu_buffer_rsrc_t tab, const f32x4 *__restrict__ pl, unsigned m4,
                                          unsigned L4, FzHalf &H) {
    const u32x4 o4 = __builtin_bit_cast(u32x4, pl[8]);
    const u32x4 pa = __builtin_bit_cast(u32x4, pl[2 * HSEL]), pb = __builtin_bit_cast(u32x4, pl[2 * HSEL + 1]);
    unsigned a[8];
    if (HSEL == 0) {                                           // reads 0-4: set 0, reads 5-7: set 1
        const unsigned s0 = fz_wrap(m4, o4.x, L4), s1 = fz_wrap(m4, o4.y, L4);
        a[0] = s0 + pa.x; a[1] = s0 + pa.y; a[2] = s0 + pa.z; a[3] = s0 + pa.w;
        a[4] = s0 + pb.x; a[5] = s1 + pb.y; a[6] = s1 + pb.z; a[7] = s1 + pb.w;
    } else {                                                   // read 8: set 1, reads 9-12: set 2, reads 13-15: set 3
        const unsigned s1 = fz_wrap(m4, o4.y, L4), s2 = fz_wrap(m4, o4.z, L4), s3 = fz_wrap(m4, o4.w, L4);
        a[0] = s1 + pa.x; a[1] = s2 + pa.y; a[2] = s2 + pa.z; a[3] = s2 + pa.w;
        a[4] = s2 + pb.x; a[5] = s3 + pb.y; a[6] = s3 + pb.z; a[7] = s3 + pb.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) H.v[j] = __builtin_amdgcn_raw_buffer_load_b128(tab, (int)a[j], 0, 0);
}

template <int HSEL>
__device__ __forceinline__ f32x4 fz_finish(const f32x4 *__restrict__ pl, const FzHalf &H, f32x4 acc) {
    const f32x4 wa = pl[4 + 2 * HSEL], wb = pl[5 + 2 * HSEL];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float w = j < 4 ? wa[j] : wb[j - 4];
        acc = __builtin_elementwise_fma(__builtin_bit_cast(f32x4, H.v[j]), f32x4{w, w, w, w}, acc);
    }
    return acc;
}
