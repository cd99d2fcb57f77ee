// Internal helpers shared by the translation units of libbas_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/bas.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// thread-local last-error text (bas_abi.hip)
void bas_set_error(const char *fmt, ...);
int bas_fail(int code, const char *fmt, ...);
int bas_check_launch(const char *what);

static inline hipStream_t bas_stream(bas_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// host helpers shared by the FIR launchers (bas_render.hip)
int bas_grid_for(long items, int cap);
int bas_device_cus();                                  // CU count of the current device, cached
hipError_t bas_allow_full_lds(const void *fn);         // dynamic-LDS limit of a kernel raised once per device
struct BasTail;                                        // bas_tail.h
// Carried state of a stream block (bas_stream.hip; bas.h "streaming"): what bas_stream_epilogue_f32 moves after a window's
// render.  The slab reduce kernels take one and do that work behind their own (x == null: nothing) - a launch less per block.
struct BasCarry {
    float *x;                   // [n_src] rows [halo | block]: the last halo samples move to the front
    long x_stride;
    int n_src, halo;
    long B;
    double *elev, *azim;        // [n_src] rows of nh + nb angles: the halo's boundaries move to the front
    long ang_stride;
    int nh, nb;
    double *last;               // [2][n_src]: the angles at the block's end
    unsigned int *running_peak; // max|y| bits over the samples emitted so far (may be null)
};
int bas_launch_slab_reduce(const float *slab, int tile, int n_src, int units_per_wg, int parts_per_wg, int n_wg,
                           long T_out, float *y, int accumulate, unsigned int *peak_bits, const BasTail *tail,
                           int *tail_skipped, const BasCarry *carry, hipStream_t st, const char *what);
// Workspaces start with a head the library owns: the 2048-byte control block (zero between calls) and BAS_TAIL_MAX_WG
// per-workgroup maxima (bas_tail.h); the slabs follow.
#define BAS_TAIL_MAX_WG 4096
#define BAS_WS_CONTROL_BYTES 2048
#define BAS_WS_HEAD_BYTES (BAS_WS_CONTROL_BYTES + 4 * BAS_TAIL_MAX_WG)

#define BAS_REQUIRE(cond, code, ...) \
    do { if (!(cond)) return bas_fail((code), __VA_ARGS__); } while (0)

// non-negative remainder for any int c and M > 0
__device__ __forceinline__ int bas_pmod(long long c, int M) {
    long long r = c % (long long)M;
    return (int)(r < 0 ? r + M : r);
}

// s = b + f with b = floor(s) (saturated to a range whose sums cannot overflow), f in [0,1)
__device__ __forceinline__ void bas_split_shift(double s, long long &b, float &f) {
    if (!(s == s)) s = 0.0;                       // NaN shift: treat as 0 (reference would raise)
    if (s > 1e15) s = 1e15;
    if (s < -1e15) s = -1e15;
    double fl = floor(s);
    b = (long long)fl;
    f = (float)(s - fl);
}

// s = b + f with b = floor(s), f in [0,1); returns b reduced into [0, M) (all in binary64, exact for
// |s| < 2^52) so that every later position sum stays in cheap 32-bit arithmetic.
__device__ __forceinline__ int bas_split_shift_mod(double s, int M, float &f) {
    if (!(s == s)) s = 0.0;                       // NaN shift: treat as 0 (reference would raise)
    if (s > 1e15) s = 1e15;
    if (s < -1e15) s = -1e15;
    const double fl = floor(s);
    f = (float)(s - fl);
    // fl * (1 / M) instead of fl / M (a binary64 division is ~30 instructions): the product can miss the quotient's
    // integer part by one when fl / M is within an ulp of an integer; r is then off by exactly M and the two guards
    // below (needed anyway) put it right - the result is the same integer either way.
    const double r = fl - (double)M * floor(fl * (1.0 / (double)M));
    int b = (int)r;
    if (b >= M) b -= M;
    if (b < 0) b += M;
    return b;
}

// (a - b) mod M for a, b in [0, M)
__device__ __forceinline__ int bas_submod(int a, int b, int M) {
    const int c = a - b;
    return c < 0 ? c + M : c;
}

// max|.| of a wave into *peak_bits (float bits of a non-negative value: unsigned order = float order).
// Atomics on ONE address serialise at ~12 ns each whatever the CU they come from - 1 724 waves of the slab reduce
// spent 21 of its 25 us there - so a wave first reads the published peak (coherently: past the per-XCD L2) and only
// sends its atomic when it would raise it; the value only grows within a launch, a stale read costs an atomic, no more.
__device__ __forceinline__ void bas_wave_peak_max(float lmax, unsigned int *peak_bits) {
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if ((threadIdx.x & 63) == 0) {
        const unsigned int mine = __float_as_uint(lmax);
        if (mine > __hip_atomic_load(peak_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(peak_bits, mine);
    }
}

// The same for a whole workgroup of 256 threads (every thread must call it): the waves meet in LDS first, one
// thread reads the published peak and sends at most one atomic per workgroup.
__device__ __forceinline__ void bas_block_peak_max(float lmax, unsigned int *peak_bits) {
    __shared__ float wave_max[4];
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if ((threadIdx.x & 63) == 0) wave_max[(threadIdx.x >> 6) & 3] = lmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int mine = __float_as_uint(fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3])));
        if (mine > __hip_atomic_load(peak_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(peak_bits, mine);
    }
}

// The moves of a stream block's carried state (everything but the running peak), by `nthreads` threads of which this is `tid`.
__device__ __forceinline__ void bas_carry_moves(const BasCarry &C, long tid, long nthreads) {
    float *x = C.x;
    const int halo = C.halo, n_src = C.n_src, nh = C.nh, nb = C.nb;
    const long B = C.B;
    // ---- input halo: x[s][0 .. halo) = x[s][B .. B + halo)
    if (B >= halo) {                                         // source and destination ranges are disjoint
        for (long i = tid; i < (long)n_src * halo; i += nthreads) {
            const long s = i / halo, j = i - s * halo;
            x[s * C.x_stride + j] = x[s * C.x_stride + B + j];
        }
    } else {                                                 // block shorter than the halo (L - 1 > B): the ranges overlap,
        for (long s = tid; s < n_src; s += nthreads)         // one thread moves a row front to back (reads run ahead of writes)
            for (int j = 0; j < halo; ++j) x[s * C.x_stride + j] = x[s * C.x_stride + B + j];
    }
    // ---- angles: boundaries t0+B-halo .. t0+B-K move to the front; the boundary at t0+B is remembered for finish()
    for (long s = tid; s < n_src; s += nthreads) {
        double *e = C.elev + s * C.ang_stride, *a = C.azim + s * C.ang_stride;
        C.last[s] = e[nh + nb - 1];
        C.last[n_src + s] = a[nh + nb - 1];
        for (int j = 0; j < nh; ++j) {                       // ascending: source index nb - 1 + j > j
            e[j] = e[nb - 1 + j];
            a[j] = a[nb - 1 + j];
        }
    }
}
