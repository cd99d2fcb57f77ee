// Tail of a kernel whose workgroups together finish y: max|y| and the peak rule of make_signal_move_2d
// (apply_hrtf.py:462-464: m = max|y|; if m > 1: y /= m) WITHOUT a launch of their own.
//
// Before: memset of the peak + FIR / reduce kernel with an atomicMax per workgroup + bas_scale_kernel - the memset and the
// scale launch are ~4.4 us each with nothing to do in the usual case (m <= 1), 26 % of a single-source step and what the root
// of a multi-GPU group pays on top of its own render.  Now every workgroup leaves its own maximum in
// `wgpeak[block]` and takes a ticket; the workgroups that arrive LAST - at most `k_last` of them - wait for the rest (all of
// whom hold a later ticket and are running or about to be dispatched: everybody else has exited, so k_last <= the number of
// workgroups the chip holds at once is all the co-residency this needs), fold the maxima, and - only when the rule fires -
// rescale one share of y each.  The very last one publishes the peak; the last to finish resets the two counters, so the
// control block is zero again when the kernel ends (hipGraph replays, the next call on the same workspace).
//
// Visibility across XCDs (MI355X_MICROARCH.md "inter-workgroup visibility"): y and wgpeak are written with sc1 stores, every
// storing wave waits vmcnt(0) before the workgroup's ticket (an agent-scope atomic add), the late workgroups read with sc1
// loads only after the counter says everyone has arrived.  No fence, no L2 write-back, nothing invalidated.
#pragma once
#include "bas_internal.h"

#define BAS_CTL_WORDS 16        // control block: 64 bytes at the end of a workspace, zero between calls
#define BAS_CTL_TICKET 0        //   workgroups that have delivered their part
#define BAS_CTL_DONE 1          //   late workgroups that have finished the tail
#define BAS_CTL_STATUS 4        //   4 words: device-side error record (bas_render_status), see BAS_STATUS_MAGIC*
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
    unsigned k_last;            // late workgroups that share the rescale (1 when normalize == 0)
    int normalize;              // apply the rule
};

__device__ __forceinline__ void bas_store4_sc1(float *p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
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

// Every thread of every workgroup of the launch calls this once, after its last store to y (sc1 stores) and with the
// maximum |value| it stored.  THREADS = workgroup size (a multiple of 64, at most 512).
template <int THREADS>
__device__ __forceinline__ void bas_tail(const BasTail &T, float lmax) {
    constexpr int NWV = THREADS / 64;
    __shared__ float t_wmax[NWV];
    __shared__ unsigned t_ticket, t_ok;
    const int tid = threadIdx.x;
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
        t_ticket = __hip_atomic_fetch_add(T.ctl + BAS_CTL_TICKET, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned ticket = t_ticket;
    if (ticket + T.k_last < T.n_wg) return;                  // not one of the last k_last: done
    // ---- a late workgroup: everybody with a smaller ticket has delivered; wait for the (at most k_last - 1) others
    if (tid == 0) {
        int spins = 0;
        unsigned ok = 1u;
        while (ticket != T.n_wg - 1 &&                       // (the very last one knows)
               __hip_atomic_load(T.ctl + BAS_CTL_TICKET, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < T.n_wg) {
            if (++spins >= BAS_TAIL_SPINS) {
                ok = 0u;
                bas_report_status(T.ctl, BAS_STATUS_TAIL_TIMEOUT, blockIdx.x);
                break;
            }
            __builtin_amdgcn_s_sleep(16);                    // (~0.5 us: the pollers must not crowd out the arrivals' atomics)
        }
        t_ok = ok;
    }
    __syncthreads();
    float m = 0.f;
    for (unsigned i = tid; i < T.n_wg; i += THREADS) m = fmaxf(m, bas_load1_sc1(T.wgpeak + i));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __syncthreads();                                         // (t_wmax is read above by thread 0 only, before the ticket)
    if ((tid & 63) == 0) t_wmax[tid >> 6] = m;
    __syncthreads();
    m = t_wmax[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) m = fmaxf(m, t_wmax[w]);
    if (T.normalize && m > 1.0f && t_ok) {                   // :463-464 - this workgroup's share of y
        const unsigned j = ticket - (T.n_wg - T.k_last);
        const long n4 = T.n >> 2;
        const long lo = n4 * j / T.k_last, hi = n4 * (j + 1) / T.k_last;
        for (long i = lo + tid; i < hi; i += THREADS) {
            f32x4 v = bas_load4_sc1(T.y + 4 * i);
            v = f32x4{v.x / m, v.y / m, v.z / m, v.w / m};
            bas_store4_sc1(T.y + 4 * i, v);
        }
        if (j == T.k_last - 1)                               // the n % 4 last values
            for (long i = 4 * n4 + tid; i < T.n; i += THREADS) bas_store1_sc1(T.y + i, bas_load1_sc1(T.y + i) / m);
    }
    if (T.peak && ticket == T.n_wg - 1 && tid == 0) *T.peak = m;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {                                          // the last one to finish leaves the control block zero
        const unsigned d = T.k_last == 1 ? 0u
                                         : __hip_atomic_fetch_add(T.ctl + BAS_CTL_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == T.k_last - 1) {
            __hip_atomic_store(T.ctl + BAS_CTL_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(T.ctl + BAS_CTL_TICKET, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// host side: how many late workgroups share the rescale (bas_render.hip)
unsigned bas_tail_k_last(unsigned n_wg, int normalize);
