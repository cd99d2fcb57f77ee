// Tail of a kernel whose workgroups together finish y: max|y| and the peak rule of make_signal_move_2d
// (apply_hrtf.py:462-464: m = max|y|; if m > 1: y /= m) WITHOUT a launch of their own.
//
// Before: memset of the peak + FIR / reduce kernel with an atomicMax per workgroup + bas_scale_kernel - the memset and the
// scale launch are ~4.4 us each with nothing to do in the usual case (m <= 1), 26 % of a single-source step and what the root
// of a multi-GPU group pays on top of its own render.  Now every workgroup leaves its own maximum in
// `wgpeak[block]` and is counted in (two levels, see bas_tail); the workgroups that complete a group - at most eight, all of
// them running when they wait for each other: everybody else has exited - fold the maxima and, only when the rule fires,
// rescale one share of y each.  The very last one publishes the peak; the last to finish resets the counters, so the
// control block is zero again when the kernel ends (hipGraph replays, the next call on the same workspace).
//
// Visibility across XCDs (MI355X_MICROARCH.md "inter-workgroup visibility"): y and wgpeak are written with sc1 stores, every
// storing wave waits vmcnt(0) before the workgroup's ticket (an agent-scope atomic add), the late workgroups read with sc1
// loads only after the counter says everyone has arrived.  No fence, no L2 write-back, nothing invalidated.
#pragma once
#include "bas_internal.h"

#define BAS_CTL_WORDS 512       // control block: 2048 bytes at the start of a workspace, zero between calls
#define BAS_CTL_GROUPS_DONE 0   //   groups of workgroups (blockIdx & 7) whose last member has delivered
#define BAS_CTL_HELPERS_DONE 1  //   group-last workgroups that have finished the tail
#define BAS_CTL_STATUS 4        //   4 words: device-side error record (bas_render_status), see BAS_STATUS_MAGIC*
#define BAS_CTL_PEAK 8          //   max|y| where the caller gave no peak pointer but asked for the rule
#define BAS_CTL_GROUP(g) (32 * ((g) + 1))       // arrivals of group g: a 128-byte line of its own (atomics on ONE address
#define BAS_CTL_GROUP_PEAK(g) (32 * ((g) + 1) + 1)  // serialise at ~12 ns each: 863 workgroups on one counter were 10 us)
#define BAS_STATUS_MAGIC0 0xBA5E7707u
#define BAS_STATUS_MAGIC1 0xDEADFA11u
#define BAS_TAIL_SPINS (1 << 20)

struct BasTail {
    unsigned *ctl;              // control block (device), zero on entry
    float *wgpeak;              // [n_wg] scratch
    float *y;                   // [n] the finished output (both ears)
    long n;
    float *peak;                // receives max|y| before the rule (may be null)
    unsigned n_wg;              // workgroups of this launch that call bas_tail
    int normalize;              // apply the rule
};

// (s_nop 1: a store of more than 8 bytes reads its data registers up to two wait states after it issues, gfx940+; for its
// own stores the compiler inserts that distance before any instruction that overwrites them - for an asm statement it
// cannot know to.  Without it `v_max_f32 v3, |v3|, |v3|` right behind the store put |y| into memory for a few lanes.)
__device__ __forceinline__ void bas_store4_sc1(float *p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void bas_store1_sc1(float *p, float v) {
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 bas_load4_sc1(const float *p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ float bas_load1_sc1(const float *p) {
    float v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// device-side error record: recognised by its signature (a workspace is never cleared on the hot path)
__device__ __forceinline__ void bas_report_status(unsigned *ctl, unsigned code, unsigned detail) {
    unsigned *s = ctl + BAS_CTL_STATUS;
    __hip_atomic_store(s + 2, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(s + 3, detail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(s + 1, BAS_STATUS_MAGIC1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(s + 0, BAS_STATUS_MAGIC0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#define BAS_STATUS_HANDOVER_TIMEOUT 1u   // a stager wave never saw its neighbour's boundary chunk IR
#define BAS_STATUS_TAIL_TIMEOUT 2u       // a late workgroup never saw the others arrive

// Every thread of every workgroup of the launch calls this once, after its last store to y (sc1 stores where the rule may be
// applied here) and with the maximum |value| it stored.  THREADS = workgroup size (a multiple of 64, at most 512).
//
// Arrivals are counted in two levels: workgroup b belongs to group b & 7 (its own 128-byte counter line: eight counters take
// the arrivals in parallel), the LAST member of a group folds the group's maxima and reports the group; the last group to
// report knows the whole launch has delivered.  With the rule, the (at most eight) group-last workgroups wait for each other -
// all of them are running by then - and share the rescale.
template <int THREADS>
__device__ __forceinline__ void bas_tail(const BasTail &T, float lmax) {
    constexpr int NWV = THREADS / 64;
    __shared__ float t_wmax[NWV];
    __shared__ unsigned t_word, t_ok;
    const int tid = threadIdx.x;
    const unsigned n_groups = T.n_wg < 8u ? T.n_wg : 8u;
    const unsigned g = blockIdx.x & 7u;                      // (n_wg < 8: every workgroup its own group)
    const unsigned n_g = (T.n_wg - g + 7u) >> 3;             // members of this group
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if ((tid & 63) == 0) t_wmax[tid >> 6] = lmax;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's y stores have left
    __syncthreads();
    if (tid == 0) {
        float m = t_wmax[0];
#pragma unroll
        for (int w = 1; w < NWV; ++w) m = fmaxf(m, t_wmax[w]);
        bas_store1_sc1(T.wgpeak + blockIdx.x, m);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t_word = __hip_atomic_fetch_add(T.ctl + BAS_CTL_GROUP(g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (t_word != n_g - 1u) return;                          // not the last of its group: done
    // ---- the group's last member: the group's maximum, then the group is reported
    float m = 0.f;
    for (unsigned i = g + 8u * (unsigned)tid; i < T.n_wg; i += 8u * THREADS)
        m = fmaxf(m, __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned *>(T.wgpeak) + i, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT)));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __syncthreads();
    if ((tid & 63) == 0) t_wmax[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        m = t_wmax[0];
#pragma unroll
        for (int w = 1; w < NWV; ++w) m = fmaxf(m, t_wmax[w]);
        __hip_atomic_store(T.ctl + BAS_CTL_GROUP(g), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (all members are in)
        __hip_atomic_store(T.ctl + BAS_CTL_GROUP_PEAK(g), __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(T.ctl + BAS_CTL_GROUPS_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned ok = 1u;
        if (T.normalize && t != n_groups - 1u) {             // the rule: the group-last workgroups share it - wait for the others
            int spins = 0;
            while (__hip_atomic_load(T.ctl + BAS_CTL_GROUPS_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_groups) {
                if (++spins >= BAS_TAIL_SPINS) {
                    ok = 0u;
                    bas_report_status(T.ctl, BAS_STATUS_TAIL_TIMEOUT, blockIdx.x);
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        t_word = t;
        t_ok = ok;
    }
    __syncthreads();
    const unsigned t = t_word;
    const bool last = t == n_groups - 1u;
    if (!T.normalize && !last) return;                       // no rule: only the last group's last member has anything left to do
    m = 0.f;
    if (tid < (int)n_groups)
        m = __uint_as_float(__hip_atomic_load(T.ctl + BAS_CTL_GROUP_PEAK(tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    for (int o = 4; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    m = __shfl(m, 0);
    __syncthreads();
    if (tid == 0) t_wmax[0] = m;
    __syncthreads();
    m = t_wmax[0];
    if (T.normalize && m > 1.0f && t_ok) {                   // :463-464 - this workgroup's share of y
        const long n4 = T.n >> 2;
        const long lo = n4 * t / n_groups, hi = n4 * (t + 1) / n_groups;
        for (long i = lo + tid; i < hi; i += THREADS) {
            f32x4 v = bas_load4_sc1(T.y + 4 * i);
            v = f32x4{v.x / m, v.y / m, v.z / m, v.w / m};
            bas_store4_sc1(T.y + 4 * i, v);
        }
        if (last)                                            // the n % 4 last values
            for (long i = 4 * n4 + tid; i < T.n; i += THREADS) bas_store1_sc1(T.y + i, bas_load1_sc1(T.y + i) / m);
    }
    if (T.peak && last && tid == 0) *T.peak = m;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {                                          // the last one to finish leaves the control block zero
        const unsigned helpers = T.normalize ? n_groups : 1u;
        const unsigned d = helpers == 1u ? 0u
                                         : __hip_atomic_fetch_add(T.ctl + BAS_CTL_HELPERS_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == helpers - 1u) {
            __hip_atomic_store(T.ctl + BAS_CTL_HELPERS_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(T.ctl + BAS_CTL_GROUPS_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

