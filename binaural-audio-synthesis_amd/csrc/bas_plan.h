// Read plan of one (query, ear) of interpolate_2d, shared by the plan/eval kernels (bas_interp.hip)
// and the fused FIR staging (bas_render.hip).
#pragma once
#include "bas_internal.h"

// Floats per phase plane of the packed table: [1 front guard = last sample][L samples][3 back guards =
// first three samples], so a lane can read up to 4 consecutive taps and "one sample earlier" unconditionally.
#define BAS_PLANE(L) ((L) + 4)

// A read set steps through up to five consecutive upsampled positions as "previous plane, or plane + U one
// sample earlier" (set_dot*): valid while 4 <= U.  Smaller factors use bas_interp2d_f32's plain kernel.
#define BAS_PLAN_MIN_U 4

struct SetPlan {
    int base;          // float index of plane ph0's sample 0 in `packed` (guard is at base - 1)
    int ph0;           // phase of read j = 0; reads j <= ph0 stay in plane ph0 - j at offset o,
    int o;             // reads j > ph0 continue in plane ph0 - j + U one sample earlier
    int dir;           // direction index of the table row the set reads
};

struct EarPlanW {
    SetPlan set[4];
    float w[16];       // same order as the sets: 5 + 4 + 4 + 3
};

// The same plan in the form the fused FIR kernel reads (bas_fused.hip): byte offsets instead of (plane, phase)
// descriptors, so the evaluator has no per-read selection work left.  A wave stages the plans of its chunk IRs
// in LDS and every lane picks up the values of ITS ear's plan with broadcast reads (16-byte LDS reads of one
// address per half-wave): off[] and o4[] feed address adds, w[] packed FMAs.
#define BAS_PLANS_WORDS 36
struct EarPlanS {
    unsigned off[16];  // byte offset into `packed` of read j's plane sample 0 (the -1 of a wrapped phase included)
    float w[16];       // folded blend weights, same order (read sets of 5 + 4 + 4 + 3)
    unsigned o4[4];    // 4 * o of the four read sets (bytes): lane offset = wrap(4 m + o4)
};

__device__ __forceinline__ void make_set(SetPlan &sp, int row, int c, int L, int U, int dir) {
    const int o = c / U, ph = c - o * U;           // c in [0, M)
    sp.base = row + ph * BAS_PLANE(L) + 1;
    sp.ph0 = ph;
    sp.o = o;
    sp.dir = dir;
}

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float rflf(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

// acc += sum_j w[j] * packed[plane(j)][wrap(m + o) (- 1 after the phase wrapped) + {0, 1}]
template <int N>
__device__ __forceinline__ f32x2 set_dot(const float *__restrict__ packed, int base, int ph0, int o,
                                          const float *w, int m, int L, int U, f32x2 acc) {
    const unsigned idx = (unsigned)(m + o);                  // m < L, o < L
    const unsigned wr = idx - (unsigned)L;
    const unsigned off = idx < wr ? idx : wr;                // idx >= L ? idx - L : idx
#pragma unroll
    for (int j = 0; j < N; ++j) {
        // wave-uniform plane base; j > ph0: plane + U, one sample earlier (the guards make -1 and +1 safe)
        const int pb = j <= ph0 ? base - j * BAS_PLANE(L) : base + (U - j) * BAS_PLANE(L) - 1;
        const f32x2 v = *reinterpret_cast<const f32x2_a4 *>(packed + pb + off);
        acc = __builtin_elementwise_fma(v, f32x2{w[j], w[j]}, acc);
    }
    return acc;
}


// Evaluates taps m, m+1 of one (query, ear) from its plan held in registers of lane k = word k
// (one 128-byte load per wave): the 12 descriptors and 16 weights are spread with v_readlane.
__device__ __forceinline__ f32x2 plan_eval_pair(const float *__restrict__ packed, int word, int m, int L, int U) {
    int base[4], ph0[4], o[4];
    float wt[16];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        base[t] = __builtin_amdgcn_readlane(word, 4 * t);
        ph0[t] = __builtin_amdgcn_readlane(word, 4 * t + 1);
        o[t] = __builtin_amdgcn_readlane(word, 4 * t + 2);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) wt[k] = __int_as_float(__builtin_amdgcn_readlane(word, 16 + k));
    f32x2 acc = f32x2{0.f, 0.f};
    acc = set_dot<5>(packed, base[0], ph0[0], o[0], wt, m, L, U, acc);
    acc = set_dot<4>(packed, base[1], ph0[1], o[1], wt + 5, m, L, U, acc);
    acc = set_dot<4>(packed, base[2], ph0[2], o[2], wt + 9, m, L, U, acc);
    acc = set_dot<3>(packed, base[3], ph0[3], o[3], wt + 13, m, L, U, acc);
    return acc;
}

// Split form for software pipelining: plan_eval_issue requests the 16 sample pairs of one row,
// plan_eval_finish (later) folds them with the 16 weights.
template <int N>
__device__ __forceinline__ void set_issue(const float *__restrict__ packed, int base, int ph0, int o, int m,
                                           int L, int U, f32x2 *v) {
    const unsigned idx = (unsigned)(m + o);
    const unsigned wr = idx - (unsigned)L;
    const unsigned off = idx < wr ? idx : wr;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int pb = j <= ph0 ? base - j * BAS_PLANE(L) : base + (U - j) * BAS_PLANE(L) - 1;
        v[j] = *reinterpret_cast<const f32x2_a4 *>(packed + pb + off);
    }
}

__device__ __forceinline__ void plan_eval_issue(const float *__restrict__ packed, int word, int m, int L, int U,
                                                 f32x2 (&v)[16]) {
    int base[4], ph0[4], o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        base[t] = __builtin_amdgcn_readlane(word, 4 * t);
        ph0[t] = __builtin_amdgcn_readlane(word, 4 * t + 1);
        o[t] = __builtin_amdgcn_readlane(word, 4 * t + 2);
    }
    set_issue<5>(packed, base[0], ph0[0], o[0], m, L, U, v);
    set_issue<4>(packed, base[1], ph0[1], o[1], m, L, U, v + 5);
    set_issue<4>(packed, base[2], ph0[2], o[2], m, L, U, v + 9);
    set_issue<3>(packed, base[3], ph0[3], o[3], m, L, U, v + 13);
}

__device__ __forceinline__ f32x2 plan_eval_finish(int word, const f32x2 (&v)[16]) {
    f32x2 acc = f32x2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float w = __int_as_float(__builtin_amdgcn_readlane(word, 16 + k));
        acc = __builtin_elementwise_fma(v[k], f32x2{w, w}, acc);
    }
    return acc;
}

// ---- four adjacent taps per lane, plan values per lane (two rows per wave) ----------------------------
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

template <int N>
__device__ __forceinline__ f32x4 set_dot4(const float *__restrict__ packed, int base, int ph0, int o,
                                           const float *w, int m, int L, int U, f32x4 acc) {
    const unsigned idx = (unsigned)(m + o);                  // m < L, o < L
    const unsigned wr = idx - (unsigned)L;
    const unsigned off = idx < wr ? idx : wr;                // idx >= L ? idx - L : idx
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int pb = j <= ph0 ? base - j * BAS_PLANE(L) : base + (U - j) * BAS_PLANE(L) - 1;
        const f32x4 v = *reinterpret_cast<const f32x4_a4 *>(packed + pb + off);
        acc = __builtin_elementwise_fma(v, f32x4{w[j], w[j], w[j], w[j]}, acc);
    }
    return acc;
}

// word = plan word (lane & 31) of THIS lane's row; both halves of the wave hold different rows, so
// every plan value is fetched from lane (half*32 + k) with a wave shuffle
__device__ __forceinline__ f32x4 plan_eval_quad(const float *__restrict__ packed, int word, int half, int m, int L,
                                                 int U) {
    int base[4], ph0[4], o[4];
    float wt[16];
    const int src0 = half << 5;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        base[t] = __shfl(word, src0 + 4 * t);
        ph0[t] = __shfl(word, src0 + 4 * t + 1);
        o[t] = __shfl(word, src0 + 4 * t + 2);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) wt[k] = __int_as_float(__shfl(word, src0 + 16 + k));
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    acc = set_dot4<5>(packed, base[0], ph0[0], o[0], wt, m, L, U, acc);
    acc = set_dot4<4>(packed, base[1], ph0[1], o[1], wt + 5, m, L, U, acc);
    acc = set_dot4<4>(packed, base[2], ph0[2], o[2], wt + 9, m, L, U, acc);
    acc = set_dot4<3>(packed, base[3], ph0[3], o[3], wt + 13, m, L, U, acc);
    return acc;
}
