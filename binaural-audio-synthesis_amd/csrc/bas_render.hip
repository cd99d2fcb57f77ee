// Time-varying FIR + overlap-add + multi-source mix (a7/a8), gfx950.
// Reference semantics: apply_hrtf.py:431-453 (per-subchunk crossfaded IR, direct
// convolution, overlap-add), :459-464 (float32 cast, peak rule).
//
//   y[e][n] = sum_s sum_k x_s[n-k] * G_{s,row(n-k)}[k][e],
//   G_{s,row}[k][e] = (1-al) H[s][c][e][k] + al H[s][c+1][e][k],  c = m/K, al = ((m%K)/S)*S/K
//
// The IR is constant over one subchunk of S input samples.  The fast kernel needs
// S % 32 == 0 and works on "rows" of 32 input samples (one crossfaded IR per row).
//
// Fast kernel, output-stationary Toeplitz blocks
// ----------------------------------------------
// A workgroup (4 waves) owns a tile of 2048 outputs and walks over (tile, source)
// work units; the mix over sources stays in registers, so HBM sees each input
// sample once and each output once.  Lane t of wave u owns the 8 consecutive
// outputs n0 + 32 t + 8 u + r (both ears: 16 accumulators as 8 float2).  In step q
// every lane multiplies an aligned block of 8 inputs (its own row, two
// ds_read_b128) with taps 8q-7 .. 8q+7 of THAT row's IR: 8x8x2 FMAs on 10 loaded
// 128-bit words.  Because all lanes of a wave sit at the same offset inside their
// (different) rows, the tap window slides uniformly: 8 new taps per step, and a
// reload of the 8 old ones only when the wave crosses into the previous row (every
// 4th step).  Row stride 8*128+16 B and a column-major x image make every
// ds_read_b128 conflict-free.  Taps are processed in segments of 128 so the LDS
// footprint (68 IR rows + x window = 79.4 KB) is independent of L and two
// workgroups share a CU.  No MFMA: this is a 1-D FIR (BASELINE.json north_star).
#include "bas_internal.h"
#include "bas_plan.h"
#include "bas_fir.h"
#include "bas_tail.h"
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>

#define RT_THREADS 256
#define RT_TILE 2048            // outputs per tile: 64 lanes x 32
#define RT_RS (2 * RT_SEG + 4)  // floats per IR row in LDS (+16 B: rows land on distinct banks)
#define RT_ROWS (RT_TILE / 32 + RT_SEG / 32)      // 68
#define RT_XR (RT_ROWS + 1)     // odd row count of the column-major x image (conflict-free stores)
#define RT_LDS_BYTES ((RT_ROWS * RT_RS + 8 * RT_XR * 4) * 4)

struct RenderArgs {
    const float *x;
    long x_stride;
    const float *H;
    int n_src;
    long T_in;
    int K, S, L, Lp;       // Lp = L rounded up to a multiple of 8
    int n_chunks;
    long units_total;      // n_tiles * n_src
    int units_per_wg;
    int parts_per_wg;
    float *slab;           // [n_wg][parts_per_wg][2][RT_TILE]
#ifdef BAS_DIAG
    int dbg;               // ablation flags (BAS_DEBUG_FLAGS; diagnostic build only, make diag)
#endif
};

// Ablation switches exist only in the diagnostic build (make diag / make stamps -> libbas_hip_diag.so,
// libbas_hip_stamps.so); the shipped library has no environment hooks and no way to skip work.
#ifdef BAS_DIAG
#define BAS_DBG(A, bit) (((A).dbg & (bit)) != 0)
#else
#define BAS_DBG(A, bit) false
#endif

#ifdef BAS_STAMPS
// Diagnostic build only (make stamps): per-wave cycle totals of the pass loop phases.
__device__ unsigned long long bas_dbg_stamps[1024 * 4 * 8];
#define STAMP(var) unsigned long long var = __builtin_readcyclecounter()
#define STAMP_ADD(slot, t0, t1) st_acc[slot] += (t1) - (t0)
#else
#define STAMP(var)
#define STAMP_ADD(slot, t0, t1)
#endif

struct Win {
    f32x2 t[8];            // 8 taps x (left,right)
};

__device__ __forceinline__ void win_load(Win &w, const float *__restrict__ p) {
    const f32x4 *p4 = reinterpret_cast<const f32x4 *>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4 v = p4[i];
        w.t[2 * i] = f32x2{v.x, v.y};
        w.t[2 * i + 1] = f32x2{v.z, v.w};
    }
}

// taps 0..7 of the segment (window lower half absent): k = r - a >= 0
__device__ __forceinline__ void fir_first(f32x2 (&acc)[8], const float (&x)[8], const Win &nw) {
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = a; r < 8; ++r) fma2(acc[r], x[a], nw.t[r - a]);
}

// full 8x8 block: window = old taps (rel 0..7) then new taps (rel 8..15), rel = r - a + 8
__device__ __forceinline__ void fir_mid(f32x2 (&acc)[8], const float (&x)[8], const Win &old,
                                         const Win &nw) {
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int rel = r - a + 8;
            fma2(acc[r], x[a], rel < 8 ? old.t[rel] : nw.t[rel - 8]);
        }
}

// last 8 taps of the segment (window upper half absent): r < a
__device__ __forceinline__ void fir_last(f32x2 (&acc)[8], const float (&x)[8], const Win &old) {
#pragma unroll
    for (int a = 1; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < a; ++r) fma2(acc[r], x[a], old.t[r - a + 8]);
}

__device__ __forceinline__ void load_x8(float (&x)[8], const f32x4 *__restrict__ xs4, int cp, int r) {
    f32x4 lo = xs4[(2 * cp) * RT_XR + r];
    f32x4 hi = xs4[(2 * cp + 1) * RT_XR + r];
    x[0] = lo.x; x[1] = lo.y; x[2] = lo.z; x[3] = lo.w;
    x[4] = hi.x; x[5] = hi.y; x[6] = hi.z; x[7] = hi.w;
}

#define RT_NH 8                 // chunk IRs a thread keeps in registers per pass
#define RT_NX 3                 // float4 of x per thread per pass: ceil(68*8 / 256)

// One LDS pass = one (tile, source) work unit x one 128-tap segment.
struct Pass {
    long tile;
    int s, seg0, Lseg;
    long top;          // n0 - seg0: output-aligned origin of the input window (multiple of 32)
    long xbase;        // first input sample held in LDS (multiple of 32, may be negative)
    int nrows;         // rows of 32 inputs staged
    int c0;            // floor(xbase / K): chunk of the first row (negative before the signal)
    int mo0;           // offset of the first row inside its chunk (multiple of 32)
};

__device__ __forceinline__ Pass make_pass(const RenderArgs &A, long unit0, int nseg, long pid) {
    Pass P;
    const long unit = unit0 + pid / nseg;
    const int sg = (int)(pid % nseg);
    P.tile = unit / A.n_src;
    P.s = (int)(unit - P.tile * A.n_src);
    P.seg0 = sg * RT_SEG;
    P.Lseg = A.Lp - P.seg0 < RT_SEG ? A.Lp - P.seg0 : RT_SEG;              // multiple of 8
    P.top = P.tile * RT_TILE - P.seg0;
    const long rho0 = (P.top - P.Lseg) >> 5;                               // floor
    P.xbase = rho0 << 5;
    P.nrows = (int)((P.top >> 5) + RT_TILE / 32 - rho0);                   // <= RT_ROWS
    // floor division that stays consistent across 0 (rows before the signal only meet x = 0,
    // they just need finite IR values)
    long cf = P.xbase / A.K;
    if (cf * A.K > P.xbase) --cf;
    P.c0 = (int)cf;
    P.mo0 = (int)(P.xbase - cf * A.K);
    return P;
}

// Register image of one pass's global loads: every thread holds its share of the x window and,
// as lane l of its wave, the chunk IRs H[c0 .. c0+RT_NH-1][both ears][taps 2l, 2l+1] (L is even on
// this path).  Every load is unconditional with a clamped address (no branches); values that must
// read as zero are masked when they are consumed.  The loads are issued at the top of a loop
// iteration and consumed at its end, never carried across the back edge: loop-carried load results
// made hipcc copy them behind an early s_waitcnt vmcnt(0), serialising load latency and FIR.
struct Prefetch {
    f32x4 xv[RT_NX];
    f32x2 hl[RT_NH], hr[RT_NH];    // left / right ear, taps (2l, 2l+1)
};

__device__ __forceinline__ void prefetch_h(Prefetch &F, const RenderArgs &A, const Pass &P, int c_first,
                                            int lane) {
    int k = P.seg0 + 2 * lane;
    if (k > A.L - 2) k = A.L - 2;                            // L even: stays 8-byte aligned
    const float *Hk = A.H + ((long)P.s * (A.n_chunks + 1)) * 2 * A.L + k;
#pragma unroll
    for (int j = 0; j < RT_NH; ++j) {
        const float *p = Hk + (long)clampi(c_first + j, 0, A.n_chunks) * 2 * A.L;
        F.hl[j] = *reinterpret_cast<const f32x2 *>(p);
        F.hr[j] = *reinterpret_cast<const f32x2 *>(p + A.L);
    }
}

__device__ __forceinline__ void prefetch_x(Prefetch &F, const RenderArgs &A, const Pass &P, int tid) {
    const float *xsrc = A.x + (long)P.s * A.x_stride;
#pragma unroll
    for (int j = 0; j < RT_NX; ++j) {
        long m = P.xbase + 4L * (tid + j * RT_THREADS);
        m = m < 0 ? 0 : (m > A.T_in - 4 ? A.T_in - 4 : m);
        F.xv[j] = *reinterpret_cast<const f32x4 *>(xsrc + m);
    }
}

__global__ __launch_bounds__(RT_THREADS, 2) void bas_render_rows32_kernel(RenderArgs A) {
    extern __shared__ f32x4 lds4[];
    float *gs = reinterpret_cast<float *>(lds4);             // [RT_ROWS][RT_RS]
    f32x4 *xs4 = lds4 + (RT_ROWS * RT_RS) / 4;               // [8][RT_XR] float4

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int u = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index = 8-block inside a row (scalar)

    const long unit0 = (long)blockIdx.x * A.units_per_wg;
    long unit1 = unit0 + A.units_per_wg;
    if (unit1 > A.units_total) unit1 = A.units_total;
    const int nseg = (A.Lp + RT_SEG - 1) / RT_SEG;
    const long n_pass = (unit1 - unit0) * nseg;

    f32x2 acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = f32x2{0.f, 0.f};

    const long first_tile = unit0 / A.n_src;
    long cur_tile = first_tile;
    float *slab_wg = A.slab + (long)blockIdx.x * A.parts_per_wg * 2 * RT_TILE;

    auto flush = [&](long tile) {
        float *dst = slab_wg + (tile - first_tile) * 2 * RT_TILE + 32 * lane + 8 * u;
        f32x4 *l4 = reinterpret_cast<f32x4 *>(dst);
        f32x4 *r4 = reinterpret_cast<f32x4 *>(dst + RT_TILE);
        l4[0] = f32x4{acc[0].x, acc[1].x, acc[2].x, acc[3].x};
        l4[1] = f32x4{acc[4].x, acc[5].x, acc[6].x, acc[7].x};
        r4[0] = f32x4{acc[0].y, acc[1].y, acc[2].y, acc[3].y};
        r4[1] = f32x4{acc[4].y, acc[5].y, acc[6].y, acc[7].y};
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = f32x2{0.f, 0.f};
    };

    if (n_pass <= 0) return;
    const float invK = 1.0f / (float)A.K;
    const bool s_pow2 = (A.S & (A.S - 1)) == 0;

    auto load_pass = [&](Prefetch &F, const Pass &P) {
        prefetch_x(F, A, P, tid);
        prefetch_h(F, A, P, P.c0, lane);
    };

    // ---- registers of pass P -> LDS: the x window (all threads) and the crossfaded IR of every row.
    // Wave w writes rows w, w+4, ..; lane l of it the taps 2l, 2l+1 of both ears (one ds_write_b128):
    //   G_row = H_c + al (H_{c+1} - H_c),  al = subchunk offset / K            (apply_hrtf.py:442-443)
    auto stage_pass = [&](Prefetch &F, const Pass &P) {
        const int nrows = P.nrows;
        // column-major image, xs4[c][r] = x[xbase + 32 r + 4 c .. +3]
#pragma unroll
        for (int j = 0; j < RT_NX; ++j) {
            const int i4 = tid + j * RT_THREADS;
            const long m = P.xbase + 4L * i4;
            const bool inside = m >= 0 && m < A.T_in;
            f32x4 v = F.xv[j];
            v = inside ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            if (i4 < nrows * 8) xs4[(i4 & 7) * RT_XR + (i4 >> 3)] = v;
        }
        if (2 * lane < P.Lseg && !BAS_DBG(A, 1)) {
            const int k0 = P.seg0 + 2 * lane;
            const f32x2 live = f32x2{k0 < A.L ? 1.0f : 0.0f, k0 + 1 < A.L ? 1.0f : 0.0f};   // taps >= L are zero
            f32x4 *gdst = reinterpret_cast<f32x4 *>(gs) + lane;
            int c = P.c0, mo = P.mo0;
            int r = 0;                                       // first row of the current chunk slot
            while (r < nrows) {
#pragma unroll
                for (int j = 0; j < RT_NH - 1; ++j) {
                    int r_end = r + ((A.K - mo) >> 5);       // rows left in this chunk
                    if (r_end > nrows) r_end = nrows;
                    const f32x2 h0l = F.hl[j] * live, h0r = F.hr[j] * live;
                    const f32x2 dl = (F.hl[j + 1] - F.hl[j]) * live, dr = (F.hr[j + 1] - F.hr[j]) * live;
                    // this wave's rows inside [r, r_end): r + ((u - r) mod 4), step 4
                    for (int rr = r + ((u - r) & 3); rr < r_end; rr += 4) {
                        const int m_in = mo + ((rr - r) << 5);                   // offset inside the chunk
                        const float al = (float)(m_in - (s_pow2 ? (m_in & (A.S - 1)) : m_in % A.S)) * invK;
                        const f32x2 gl = __builtin_elementwise_fma(dl, f32x2{al, al}, h0l);
                        const f32x2 gr = __builtin_elementwise_fma(dr, f32x2{al, al}, h0r);
                        gdst[rr * (RT_RS / 4)] = f32x4{gl.x, gr.x, gl.y, gr.y};  // [tap][ear]
                    }
                    mo += (r_end - r) << 5;
                    r = r_end;
                    if (mo >= A.K) {
                        mo = 0;
                        ++c;
                    }
                }
                if (r < nrows) prefetch_h(F, A, P, c, lane);  // small K: next batch of chunk IRs on demand
            }
        }
    };

    // ---- FIR of the pass staged in LDS: blocks q = 0..Q; block q = inputs [n_t - seg0 - 8q, +8) x taps
    // seg0 + 8q-7 .. 8q+7.  Block q of wave u sits at 8-block cp = (u - q) mod 4 of its row, so a wave
    // enters a new (previous) row at q = u+1, u+5, ...  Blocks 0..u (top row) and the tail are handled
    // one by one; in between, groups of four blocks [enter row, slide, slide, slide] run with fixed
    // register roles and immediate LDS offsets (no copies, no address arithmetic).
    auto fir_pass = [&](const Pass &P) {
        const int Q = P.Lseg >> 3;                           // 8-tap groups in this segment
        Win wx, wy;
        float x8[8];
        const int rsh = (int)((P.top - P.xbase) >> 5);       // rows above the tile's first input row
        const int off_top = 32 * rsh + 8 * u;
        auto generic_mid = [&](int q) {                      // any middle block, both halves loaded
            const int off = off_top - 8 * q;
            const int r = lane + (off >> 5), cp = (off & 31) >> 3;
            const float *grow = gs + r * RT_RS + 16 * q;
            load_x8(x8, xs4, cp, r);
            win_load(wx, grow - 16);
            win_load(wy, grow);
            fir_mid(acc, x8, wx, wy);
        };
        {                                                    // q = 0: taps 0..7 only
            const int r = lane + rsh;
            load_x8(x8, xs4, u, r);
            win_load(wy, gs + r * RT_RS);
            fir_first(acc, x8, wy);
        }
        int q = 1;
        for (; q <= u && q < Q; ++q) generic_mid(q);         // rest of the top row
        if (q == u + 1) {
            const float *gp = gs + (lane + rsh - 1) * RT_RS + 16 * q;
            const f32x4 *xp = xs4 + (lane + rsh - 1);
            for (; q + 3 < Q; q += 4) {
                auto xld = [&](int cp) {
                    f32x4 lo = xp[(2 * cp) * RT_XR], hi = xp[(2 * cp + 1) * RT_XR];
                    x8[0] = lo.x; x8[1] = lo.y; x8[2] = lo.z; x8[3] = lo.w;
                    x8[4] = hi.x; x8[5] = hi.y; x8[6] = hi.z; x8[7] = hi.w;
                };
                xld(3);                                      // enter the row: both halves
                win_load(wx, gp - 16);
                win_load(wy, gp);
                fir_mid(acc, x8, wx, wy);
                xld(2);
                win_load(wx, gp + 16);
                fir_mid(acc, x8, wy, wx);
                xld(1);
                win_load(wy, gp + 32);
                fir_mid(acc, x8, wx, wy);
                xld(0);
                win_load(wx, gp + 48);
                fir_mid(acc, x8, wy, wx);
                gp += 64 - RT_RS;
                xp -= 1;
            }
        }
        for (; q < Q; ++q) generic_mid(q);                   // tail middles
        {                                                    // q = Q: the last 8 taps only
            const int off = off_top - 8 * Q;
            const int r = lane + (off >> 5), cp = (off & 31) >> 3;
            load_x8(x8, xs4, cp, r);
            win_load(wx, gs + r * RT_RS + 16 * Q - 16);
            fir_last(acc, x8, wx);
        }
    };

#ifdef BAS_STAMPS
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long st_begin = __builtin_readcyclecounter();
#endif
    Pass P = make_pass(A, unit0, nseg, 0);
    {
        Prefetch F0;
        load_pass(F0, P);
        stage_pass(F0, P);
    }
    __syncthreads();
    for (long pid = 0; pid < n_pass; ++pid) {
        // loads of the NEXT pass are issued here and consumed after this pass's FIR, inside the same
        // iteration (the last iteration re-loads its own pass: harmless, keeps the loads unconditional)
        const bool has_next = pid + 1 < n_pass;
        const Pass N = make_pass(A, unit0, nseg, has_next ? pid + 1 : pid);
        Prefetch F;
        STAMP(t0);
        load_pass(F, N);
        if (P.tile != cur_tile) {
            flush(cur_tile);
            cur_tile = P.tile;
        }
        STAMP(t1);
        if (!BAS_DBG(A, 2)) fir_pass(P);
        STAMP(t2);
        __syncthreads();                                     // every wave has finished reading LDS
        STAMP(t3);
#ifdef BAS_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(t3b);
        STAMP_ADD(5, t3, t3b);
#endif
        if (has_next) stage_pass(F, N);
        STAMP(t4);
        __syncthreads();
        STAMP(t5);
        STAMP_ADD(0, t0, t1); STAMP_ADD(1, t1, t2); STAMP_ADD(2, t2, t3); STAMP_ADD(3, t3, t4); STAMP_ADD(4, t4, t5);
        P = N;
    }
#ifdef BAS_STAMPS
    if (lane == 0 && blockIdx.x < 1024) {
        for (int i = 0; i < 6; ++i) bas_dbg_stamps[(blockIdx.x * 4 + u) * 8 + (i < 5 ? i : 6)] = st_acc[i];
        bas_dbg_stamps[(blockIdx.x * 4 + u) * 8 + 5] = __builtin_readcyclecounter() - st_begin;
    }
#endif
    flush(cur_tile);
}

// ===========================================================================
// Fast kernel 2 ("hd"): lane = one row of 32 outputs, crossfaded IR formed on the fly
// ===========================================================================
// Each lane owns 32 consecutive, row-aligned outputs (both ears: 64 accumulators).  Its 128 taps
// reach back over 5 input rows; for input row rho' (32 inputs, ONE crossfaded IR) the lane needs the
// taps 32 rho' - 31 .. 32 rho' + 31 of that row's IR: a 32 x 32 Toeplitz block = 1024 FMAs per ear on
// 8 + 63 LDS words of 128 bit.  LDS holds only the x window and, per chunk slot, the pair
// (H_c, H_{c+1} - H_c) interleaved per tap as (h0_L, h0_R, d_L, d_R); the lane forms
// g = h0 + al d (al = its row's crossfade weight) with one packed FMA per tap, 6 % on top of the FIR.
// Nothing row-specific is staged, so a pass costs two barriers and a few dozen LDS stores; with ~71 KB
// of LDS two 4-wave workgroups share a CU.  Variants (template parameters of the kernel below):
//   NSUB   S = 16 / 8 / 4 with K a multiple of 32: a row holds 2 / 4 / 8 subchunks, each with its own formed taps
//   HONLY  more than HD_MAXSLOTS chunk slots under a tile (K < 448): the slots hold (h_L, h_R) only
//   DUAL   any other subchunk / chunk size: the row step runs once per part of a row (multi-part rows)
// (The fused form - chunk IRs evaluated from the table while staging - is bas_render_fz_kernel, bas_fused.hip.)
#define HD_NW 4                               // waves per workgroup: one per SIMD, so the CU stays balanced
#define HD_THREADS (64 * HD_NW)
#define HD_TILE (2048 * HD_NW)                // outputs per tile
#define HD_ROWS (HD_TILE / 32 + HD_HALO)      // rows in the x window (260)
#define HD_XR (HD_ROWS + 1)                   // odd: conflict-free column-major image
#define HD_NX ((HD_ROWS * 8 + HD_THREADS - 1) / HD_THREADS)     // float4 of x per thread (9)
#define HD_MAXSLOTS 20                     // chunk slots for two workgroups per CU
#define HD_HALFSLOTS (HD_MAXSLOTS / 2)     // chunk slots staged by one half of the threads
#define HD_X_FLOATS (8 * HD_XR * 4)

__device__ __forceinline__ void hd_load_xrow(float (&xr)[32], const f32x4 *__restrict__ xrow) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4 v = xrow[c * HD_XR];
        xr[4 * c] = v.x; xr[4 * c + 1] = v.y; xr[4 * c + 2] = v.z; xr[4 * c + 3] = v.w;
    }
}

// Row step with a run-time set of live octets (segments shorter than 128 taps, last row).
template <int NSUB, bool HONLY = false>
__device__ __forceinline__ void hd_row_step_masked(f32x2 (&acc)[32], const f32x4 *__restrict__ xrow,
                                                    const float *__restrict__ hdrow, const float (&al)[NSUB],
                                                    unsigned live_mask) {
    float xr[32];
    hd_load_xrow(xr, xrow);
    hd_row_step_x<NSUB, HONLY>(acc, xr, hdrow, al, live_mask);
}

// DUAL: any chunk size >= 32 and any subchunk size.  A row of 32 inputs then meets subchunk (or chunk) boundaries
// at inputs that differ from lane to lane: the row step runs once per part of the row, on that part's inputs
// (the others zeroed) with its (slot, alpha) - two parts for subchunks >= 32 (twice the FMAs), up to
// ceil(32 / S) + 1 below that; still far ahead of the generic kernel.
template <int NSUB, bool HONLY = false, bool DUAL = false>
__global__ __launch_bounds__(HD_THREADS, 2) void bas_render_hd_kernel(RenderArgs A, int nslots) {
    static_assert(!DUAL || NSUB == 1, "multi-part rows: one alpha per part");
    extern __shared__ f32x4 lds4[];
    f32x4 *xs4 = lds4;                                       // [8][HD_XR] float4
    float *hd = reinterpret_cast<float *>(lds4) + HD_X_FLOATS;   // [nslots][HD_SLOT]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    const long unit0 = (long)blockIdx.x * A.units_per_wg;
    long unit1 = unit0 + A.units_per_wg;
    if (unit1 > A.units_total) unit1 = A.units_total;
    const int nseg = (A.Lp + RT_SEG - 1) / RT_SEG;
    const long n_pass = (unit1 - unit0) * nseg;
    if (n_pass <= 0) return;
#ifdef BAS_STAMPS
    const unsigned long long st_begin = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_fir = 0;
#endif

    // one alpha per whole row, or two with (h0, d) slots (NSUB = 2: S = 16; with h-only slots 52 spilled registers make it
    // slower), no multi-part rows (200 spilled registers): the row step runs as a 2-parallel fast FIR on half-rate partial
    // sums that are combined into the 32 outputs when the tile is flushed (bas_fir.h); otherwise the direct form on acc
    constexpr bool FFA = (NSUB == 1 || (NSUB == 2 && !HONLY)) && !DUAL;
    f32x2 acc[32];
    f32x2 fa[16], fb[17], fp[16];
#pragma unroll
    for (int o = 0; o < 32; ++o) acc[o] = f32x2{0.f, 0.f};
    ffa_zero(fa, fb, fp);

    const long first_tile = unit0 / A.n_src;
    long cur_tile = first_tile;
    float *slab_wg = A.slab + (long)blockIdx.x * A.parts_per_wg * 2 * HD_TILE;
    const float invK = 1.0f / (float)A.K;
    const unsigned prio_flip = blockIdx.x >= (gridDim.x >> 1) ? 1u : 0u;
#ifdef BAS_DIAG
    const int prio_slice = (A.dbg >> 8) & 31 ? (A.dbg >> 8) & 31 : ((A.dbg & 16) ? 0 : 12);
#else
    constexpr int prio_slice = 12;                           // priority slices of 2^12 ticks = 41 us
#endif

    auto flush = [&](long tile) {
        float *dst = slab_wg + (tile - first_tile) * 2 * HD_TILE + 2048 * wv + 32 * lane;
        f32x4 *l4 = reinterpret_cast<f32x4 *>(dst);
        f32x4 *r4 = reinterpret_cast<f32x4 *>(dst + HD_TILE);
        if constexpr (FFA) {
            ffa_combine(acc, fa, fb, fp);
            ffa_zero(fa, fb, fp);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            l4[i] = f32x4{acc[4 * i].x, acc[4 * i + 1].x, acc[4 * i + 2].x, acc[4 * i + 3].x};
            r4[i] = f32x4{acc[4 * i].y, acc[4 * i + 1].y, acc[4 * i + 2].y, acc[4 * i + 3].y};
        }
#pragma unroll
        for (int o = 0; o < 32; ++o) acc[o] = f32x2{0.f, 0.f};
    };

    for (long pid = 0; pid < n_pass; ++pid) {
        const long unit = unit0 + pid / nseg;
        const int seg0 = (int)(pid % nseg) * RT_SEG;
        const long tile = unit / A.n_src;
        const int s = (int)(unit - tile * A.n_src);
        const int Lseg = A.Lp - seg0 < RT_SEG ? A.Lp - seg0 : RT_SEG;
        const int halo = (Lseg + 31) >> 5;                   // input rows above the tile that matter
        const long xbase = tile * HD_TILE - seg0 - 32L * halo;   // first input sample in LDS (mult. of 32)
        const int nrows = HD_TILE / 32 + halo;
        long cf = xbase / A.K;                               // floor division, consistent across 0
        if (cf * A.K > xbase) --cf;
        const int c0 = (int)cf;
        const int mo0 = (int)(xbase - cf * A.K);
        if (tile != cur_tile) {
            flush(cur_tile);
            cur_tile = tile;
        }

        // ---- global -> registers: x window (all threads), chunk IRs of tap `tid` (threads < Lseg)
        // window offsets i = 4 (tid + 256 j) are per-thread constants; the signal's ends become two scalar
        // bounds on them (x rows are 16-byte aligned and T_in is a multiple of 4, so a float4 is all in or out)
        const float *xwin = A.x + (long)s * A.x_stride + xbase;
        const long lo_l = -xbase, hi_l = A.T_in - xbase;     // offsets of the signal's first sample / one past its last
        const int x_lo = lo_l < -(1 << 30) ? -(1 << 30) : (lo_l > (1 << 30) ? (1 << 30) : (int)lo_l);
        const int x_hi = hi_l < -(1 << 30) ? -(1 << 30) : (hi_l > (1 << 30) ? (1 << 30) : (int)hi_l);
        const bool x_inside = x_lo <= 0 && x_hi >= 4 * HD_NX * HD_THREADS;   // whole window inside the signal
        // T_in % 4 != 0 (dual rows, odd chunk sizes): the float4 that straddles the signal's end is read whole -
        // the row stride is a multiple of 4, so the floats exist - and masked element by element below
        const int x_hi4 = (x_hi + 3) & ~3;
        f32x4 xv[HD_NX];
#pragma unroll
        for (int j = 0; j < HD_NX; ++j) {
            int i = 4 * (tid + j * HD_THREADS);
            i = i < x_lo ? x_lo : i;
            i = i > x_hi4 - 4 ? x_hi4 - 4 : i;               // clamped into the row (x_hi4 - 4 >= x_lo as T_in >= 1)
            xv[j] = *reinterpret_cast<const f32x4 *>(xwin + i);
        }
        // chunk IRs: thread = (tap, half), each half of the threads stages half of the chunk slots from H
        const int tap = tid & (RT_SEG - 1);
        // (h0, d) image: a half stages slots [slot_a, slot_b) and needs H of slot_b too; h-only image: a half
        // stages the H values [slot_a, slot_b] of the nslots + 1 chunk boundaries (slot_b inclusive = last - 1 + 1)
        const int per_half = HONLY ? ((nslots + 2) >> 1) : ((nslots + 1) >> 1);
        const int slot_a = __builtin_amdgcn_readfirstlane(tid >> 7) * per_half;   // wave-uniform: scalar slot math
        int slot_b = slot_a + per_half;
        if (HONLY) slot_b -= 1;                               // inclusive end of the values this half loads
        if (slot_b > nslots) slot_b = nslots;
        f32x2 hlr[HD_HALFSLOTS + 1];                         // (left, right) tap of the slots this thread stages
        int k = seg0 + tap;
        if (k > A.L - 1) k = A.L - 1;
        const float *Hk = A.H + ((long)s * (A.n_chunks + 1)) * 2 * A.L + k;
#pragma unroll
        for (int j = 0; j <= HD_HALFSLOTS; ++j) {
            if (slot_a + j <= slot_b) {                  // uniform per wave
                const float *p = Hk + (long)clampi(c0 + slot_a + j, 0, A.n_chunks) * 2 * A.L;
                hlr[j].x = p[0];
                hlr[j].y = p[A.L];
            }
        }

        __syncthreads();                                     // previous pass has finished reading LDS
#pragma unroll
        for (int j = 0; j < HD_NX; ++j) {
            const int i4 = tid + j * HD_THREADS;
            f32x4 v = xv[j];
            if (!x_inside)                                   // uniform: only windows that overlap an end of the signal
            {
                const int e = 4 * i4;                        // x_lo is a multiple of 4: one lower test for all four
                const bool lo_ok = e >= x_lo;
                v.x = (lo_ok && e < x_hi) ? v.x : 0.f;
                v.y = (lo_ok && e + 1 < x_hi) ? v.y : 0.f;
                v.z = (lo_ok && e + 2 < x_hi) ? v.z : 0.f;
                v.w = (lo_ok && e + 3 < x_hi) ? v.w : 0.f;
            }
            if (i4 < nrows * 8) xs4[(i4 & 7) * HD_XR + (i4 >> 3)] = v;
        }
        const bool beyond = seg0 + tap >= A.L;           // taps >= L read as zero (only when L % 8 != 0)
        f32x2 *dst = reinterpret_cast<f32x2 *>(hd) + (HONLY ? slot_a * (HO_SLOT / 2) + tap
                                                             : 2 * (slot_a * (HD_SLOT / 4) + tap));
        if (HONLY) {
            // h-only image: plain copies of the H values slot_a .. slot_b, rounds of HD_HALFSLOTS + 1
            if (tap < Lseg) {
#pragma unroll
                for (int j = 0; j <= HD_HALFSLOTS; ++j)
                    if (slot_a + j <= slot_b) dst[j * (HO_SLOT / 2)] = beyond ? f32x2{0.f, 0.f} : hlr[j];
            }
            constexpr int XH = 6;
            for (int base = HD_HALFSLOTS + 1; slot_a + base <= slot_b; base += XH) {      // uniform per wave
                int k = seg0 + tap;
                if (k > A.L - 1) k = A.L - 1;
                const float *Hk = A.H + ((long)s * (A.n_chunks + 1)) * 2 * A.L + k;
                f32x2 more[XH];
#pragma unroll
                for (int j = 0; j < XH; ++j) {
                    if (slot_a + base + j <= slot_b) {
                        const float *p = Hk + (long)clampi(c0 + slot_a + base + j, 0, A.n_chunks) * 2 * A.L;
                        more[j].x = beyond ? 0.f : p[0];
                        more[j].y = beyond ? 0.f : p[A.L];
                    }
                }
                if (tap < Lseg) {
#pragma unroll
                    for (int j = 0; j < XH; ++j)
                        if (slot_a + base + j <= slot_b) dst[(base + j) * (HO_SLOT / 2)] = more[j];
                }
            }
        } else if (tap < Lseg) {
            if (beyond) {
#pragma unroll
                for (int j = 0; j <= HD_HALFSLOTS; ++j) hlr[j] = f32x2{0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < HD_HALFSLOTS; ++j) {
                if (slot_a + j < slot_b) {               // (h0_L, h0_R | d_L, d_R): two 8-byte halves, no repacking
                    dst[j * (HD_SLOT / 2)] = hlr[j];
                    dst[j * (HD_SLOT / 2) + 1] = hlr[j + 1] - hlr[j];
                }
            }
        }
        __syncthreads();

        // ---- FIR: input rows rho' = 0..halo above/at the lane's output row
        const int row_out = 64 * wv + lane + halo;           // window row holding the lane's outputs
        int pos = mo0 + 32 * row_out;
        int sl = pos / A.K;                                  // chunk slot of that row
        int m_in = pos - sl * A.K;                           // offset of the row inside its chunk
        const f32x4 *xrow = xs4 + row_out;
        // octet i of row step rp holds taps 32 rp - 32 + 8 i .. +7 (relative to seg0); live iff in [0, Lseg)
        auto mask_of = [&](int rp) {
            unsigned mk = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int t0 = 32 * rp - 32 + 8 * i;
                if (t0 >= 0 && t0 < Lseg) mk |= 1u << i;
            }
            return mk;
        };
        const bool s_is_pow2 = (A.S & (A.S - 1)) == 0;       // uniform
        const float invS1 = 1.0f / (float)A.S;
        auto step_setup = [&](int rp, float (&al)[NSUB], const float *&hdrow) {
            if (NSUB == 1) {
                if (s_is_pow2) {
                    al[0] = (float)(m_in - (m_in & (A.S - 1))) * invK;
                } else {                                     // any multiple of 32: m_in / S by float estimate, corrected
                    int q = (int)((float)m_in * invS1);
                    int r = m_in - q * A.S;
                    if (r < 0) q -= 1;
                    if (r >= A.S) q += 1;
                    al[0] = (float)(q * A.S) * invK;
                }
            } else {
#pragma unroll
                for (int u = 0; u < NSUB; ++u) al[u] = (float)(m_in + u * (32 / NSUB)) * invK;   // S = 32 / NSUB
            }
            hdrow = HONLY ? hd + sl * HO_SLOT + (32 * rp - 32) * 2 : hd + sl * HD_SLOT + (32 * rp - 32) * 4;
        };
        auto step_done = [&]() {
            xrow -= 1;
            m_in -= 32;
            if (m_in < 0) {
                m_in += A.K;
                sl -= 1;
            }
        };
#ifdef BAS_STAMPS
        const unsigned long long tf0 = __builtin_amdgcn_s_memrealtime();
#endif
        // Fair time slicing between the two workgroups of a CU.  The SIMD arbiter serves the older
        // wave first, so without this the first-dispatched workgroup runs at nearly full speed,
        // finishes ~35 % early and leaves the CU with one wave per SIMD.  Priority alternates every
        // 2^slice_shift ticks of the 100 MHz clock, in opposite phase for the two halves of the grid
        // (blocks b and b + grid/2 share a CU under the observed dispatch order; speed only).
        if (prio_slice) {
            const unsigned t = (unsigned)(__builtin_amdgcn_s_memrealtime() >> prio_slice);
            if ((t & 1u) ^ prio_flip) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
        }
        if (DUAL) {
            const float invS = 1.0f / (float)A.S;
            for (int rp = 0; rp <= halo; ++rp) {
                // subchunk of the row's first input: q = m_in / S (float estimate, corrected), r = m_in % S
                int q = (int)((float)m_in * invS);
                int r = m_in - q * A.S;
                if (r < 0) { q -= 1; r += A.S; }
                if (r >= A.S) { q += 1; r -= A.S; }
                const int tap_off = (32 * rp - 32) * (HONLY ? 2 : 4);
                const int slot_f = HONLY ? HO_SLOT : HD_SLOT;
                // the row's inputs [a0, a1) lie in subchunk q of chunk slot `slot`; walk the parts until every
                // lane of the wave is through its row (S >= 32: at most two parts; S = 10: up to five)
                int a0 = 0, slot = sl;
                int a1 = A.S - r;                            // inputs left in the first subchunk
                while (__any(a0 < 32)) {
                    float xa[32];
                    hd_load_xrow(xa, xrow);                  // read again per part rather than hold 32 more registers
#pragma unroll
                    for (int a = 0; a < 32; ++a) xa[a] = (a >= a0 && a < a1) ? xa[a] : 0.f;
                    float al[1] = {(float)(q * A.S) * invK};
                    hd_row_step_x<1, HONLY>(acc, xa, hd + slot * slot_f + tap_off, al, mask_of(rp));
                    if (a1 < 32) {                           // next subchunk; past the chunk's last one: next chunk, alpha 0
                        q += 1;
                        if (q * A.S >= A.K) {
                            q = 0;
                            slot += 1;
                        }
                    }
                    a0 = a1;                                 // (a finished lane keeps a0 >= 32: its parts are empty)
                    a1 = a1 + A.S;
                }
                step_done();
            }
        } else {
            for (int rp = 0; rp <= halo; ++rp) {
                float al[NSUB];
                const float *hdrow;
                step_setup(rp, al, hdrow);
                if constexpr (FFA && NSUB == 1) {
                    float xr[32];
                    hd_load_xrow(xr, xrow);
                    ffa_row_step_x<HONLY>(fa, fb, fp, xr, hdrow, al[0], mask_of(rp));
                } else if constexpr (FFA && NSUB == 2) {
                    float xr[32];
                    hd_load_xrow(xr, xrow);
                    ffa2_row_step_x<HONLY>(fa, fb, fp, xr, hdrow, al, mask_of(rp));
                } else {
                    hd_row_step_masked<NSUB, HONLY>(acc, xrow, hdrow, al, mask_of(rp));
                }
                step_done();
            }
        }
#ifdef BAS_STAMPS
        st_fir += __builtin_amdgcn_s_memrealtime() - tf0;
#endif
    }
    flush(cur_tile);
#ifdef BAS_STAMPS
    if (lane == 0 && blockIdx.x < 1024) {
        unsigned long long *d = bas_dbg_stamps + (blockIdx.x * 4 + wv) * 8;
        d[0] = st_begin;
        d[1] = __builtin_amdgcn_s_memrealtime();
        d[2] = st_fir;
        d[3] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
        d[4] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    }
#endif
}

// ---------------------------------------------------------------------------
// Generic fallback: any K, S | K, L (e.g. S not a multiple of 32, unaligned x).
// One thread per output sample, plain loops; correctness path, not tuned.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bas_render_generic_kernel(const float *__restrict__ x, long x_stride,
                                                                   const float *__restrict__ H, int n_src,
                                                                   long T_in, int K, int S, int L,
                                                                   int n_chunks, long T_out,
                                                                   float *__restrict__ y, int accumulate,
                                                                   unsigned int *peak_bits) {
    float lmax = 0.f;
    for (long n = blockIdx.x * 256L + threadIdx.x; n < T_out; n += (long)gridDim.x * 256L) {
        float al_l = 0.f, al_r = 0.f;
        const float invK = 1.0f / (float)K;
        for (int s = 0; s < n_src; ++s) {
            const float *xs = x + (long)s * x_stride;
            const float *Hs = H + (long)s * (n_chunks + 1) * 2 * L;
            for (int k = 0; k < L; ++k) {
                const long m = n - k;
                if (m < 0) break;
                if (m >= T_in) continue;
                const long c = m / K;
                const int mo = (int)(m - c * K);
                const float al = (float)((mo / S) * S) * invK;
                const float *h0 = Hs + c * 2 * L + k, *h1 = h0 + 2 * L;
                const float xv = xs[m];
                al_l += xv * ((1.0f - al) * h0[0] + al * h1[0]);
                al_r += xv * ((1.0f - al) * h0[L] + al * h1[L]);
            }
        }
        if (accumulate) {
            al_l += y[n];
            al_r += y[T_out + n];
        }
        y[n] = al_l;
        y[T_out + n] = al_r;
        lmax = fmaxf(lmax, fmaxf(fabsf(al_l), fabsf(al_r)));
    }
    if (peak_bits) {
        bas_block_peak_max(lmax, peak_bits);
    }
}

// ---------------------------------------------------------------------------
// Slab reduction (fixed order => deterministic) + optional fused max|y|
// ---------------------------------------------------------------------------
// T.ctl != null: y is stored with sc1 stores and the kernel ends in bas_tail (bas_tail.h: max|y| without atomics or a cleared
// peak word, and the peak rule apply_hrtf.py:462-464 without a launch of its own); else the round-3 form (atomicMax into
// a peak word that the FIR kernel in front has cleared).
__global__ __launch_bounds__(256) void bas_slab_reduce_kernel(const float *__restrict__ slab, int tile_len,
                                                                int n_src, int units_per_wg,
                                                                int parts_per_wg, int n_wg, long T_out,
                                                                float *__restrict__ y, int accumulate,
                                                                unsigned int *peak_bits, BasTail T, BasCarry C) {
    float lmax = 0.f;
    const long n4 = (T_out + 3) / 4;
    // C.x != null (a stream block, never together with a tail): the maximum is taken over the samples the block emits and
    // goes into the stream's running peak; the carried state is moved behind the sums (bas_carry_moves)
    const long plo = C.x ? C.halo : 0, phi = C.x ? C.halo + C.B : T_out;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
        const long n = i * 4;
        // (tiles * n_src = units_total < 2^31 is checked by the launcher: 32-bit divisions suffice)
        const unsigned tile = (unsigned)(n / tile_len);
        const int off = (int)(n - (long)tile * tile_len);
        const unsigned ulo = tile * (unsigned)n_src, uhi = ulo + n_src - 1;
        const int wlo = (int)(ulo / (unsigned)units_per_wg);
        int whi = (int)(uhi / (unsigned)units_per_wg);
        if (whi > n_wg - 1) whi = n_wg - 1;
        f32x4 sl = f32x4{0.f, 0.f, 0.f, 0.f}, sr = sl;
        for (int w = wlo; w <= whi; ++w) {
            const unsigned first_tile = ((unsigned)w * (unsigned)units_per_wg) / (unsigned)n_src;
            const float *p = slab + (((long)w * parts_per_wg + (tile - first_tile)) * 2) * tile_len + off;
            sl += *reinterpret_cast<const f32x4 *>(p);
            sr += *reinterpret_cast<const f32x4 *>(p + tile_len);
        }
        if (T.ctl && n + 3 < T_out) {                        // whole quads: one 16-byte sc1 store per ear
            if (accumulate) {
                sl += *reinterpret_cast<const f32x4_a4 *>(y + n);
                sr += *reinterpret_cast<const f32x4_a4 *>(y + T_out + n);
            }
            bas_store4_sc1(y + n, sl);
            bas_store4_sc1(y + T_out + n, sr);
            lmax = fmaxf(lmax, fmaxf(fmaxf(fmaxf(fabsf(sl.x), fabsf(sl.y)), fmaxf(fabsf(sl.z), fabsf(sl.w))),
                                     fmaxf(fmaxf(fabsf(sr.x), fabsf(sr.y)), fmaxf(fabsf(sr.z), fabsf(sr.w)))));
            continue;
        }
        float vl[4] = {sl.x, sl.y, sl.z, sl.w}, vr[4] = {sr.x, sr.y, sr.z, sr.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (n + j < T_out) {
                float a = vl[j], b = vr[j];
                if (accumulate) {
                    a += y[n + j];
                    b += y[T_out + n + j];
                }
                if (T.ctl) {
                    bas_store1_sc1(y + n + j, a);
                    bas_store1_sc1(y + T_out + n + j, b);
                } else {
                    y[n + j] = a;
                    y[T_out + n + j] = b;
                }
                if (n + j >= plo && n + j < phi) lmax = fmaxf(lmax, fmaxf(fabsf(a), fabsf(b)));
            }
        }
    }
    if (T.ctl) {
        bas_tail<256>(T, lmax);
    } else if (C.x) {
        if (C.running_peak) bas_block_peak_max(lmax, C.running_peak);
        bas_carry_moves(C, blockIdx.x * 256L + threadIdx.x, (long)gridDim.x * 256L);
    } else if (peak_bits) {
        bas_block_peak_max(lmax, peak_bits);
    }
}

// Many parts per tile (a short signal with many sources: every source's workgroup holds its own part - the
// streaming renderer's small blocks): the loop over parts of the kernel above becomes the critical path (84 us for
// 256 parts).  Here a block sums 4 float4 columns of one ear: thread = (column, part lane); part lane p adds the
// parts p, p + 64, .. in that order, then the 64 lanes are folded in a fixed tree through LDS: deterministic too.
__global__ __launch_bounds__(256) void bas_slab_reduce_wide_kernel(const float *__restrict__ slab, int tile_len,
                                                                     int n_src, int units_per_wg,
                                                                     int parts_per_wg, int n_wg, long T_out,
                                                                     float *__restrict__ y, int accumulate,
                                                                     unsigned int *peak_bits, BasTail T, BasCarry C) {
    __shared__ f32x4 part_sum[4][64];
    const long plo = C.x ? C.halo : 0, phi = C.x ? C.halo + C.B : T_out;     // (a stream block: see bas_slab_reduce_kernel)
    const int col = threadIdx.x & 3, pl = threadIdx.x >> 2;
    const long cols_per_ear = (T_out + 3) / 4;
    const long blocks_per_ear = (cols_per_ear + 3) / 4;
    const int ear = blockIdx.x >= blocks_per_ear ? 1 : 0;
    const long c4 = (blockIdx.x - ear * blocks_per_ear) * 4 + col;        // float4 column of this thread
    const long n = 4 * c4;
    f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
    if (n < T_out) {
        const unsigned tile = (unsigned)(n / tile_len);
        const int off = (int)(n - (long)tile * tile_len);
        const unsigned ulo = tile * (unsigned)n_src, uhi = ulo + n_src - 1;
        const int wlo = (int)(ulo / (unsigned)units_per_wg);
        int whi = (int)(uhi / (unsigned)units_per_wg);
        if (whi > n_wg - 1) whi = n_wg - 1;
        for (int w = wlo + pl; w <= whi; w += 64) {
            const unsigned first_tile = ((unsigned)w * (unsigned)units_per_wg) / (unsigned)n_src;
            const float *p = slab + (((long)w * parts_per_wg + (tile - first_tile)) * 2 + ear) * tile_len + off;
            sum += *reinterpret_cast<const f32x4 *>(p);
        }
    }
    part_sum[col][pl] = sum;
    __syncthreads();
    for (int step = 32; step > 0; step >>= 1) {                           // fixed tree over the 64 part lanes
        if (pl < step) part_sum[col][pl] += part_sum[col][pl + step];
        __syncthreads();
    }
    float lmax = 0.f;
    if (pl == 0 && n < T_out) {
        const f32x4 v = part_sum[col][0];
        const float vv[4] = {v.x, v.y, v.z, v.w};
        float *ye = y + (long)ear * T_out;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (n + j < T_out) {
                const float a = accumulate ? vv[j] + ye[n + j] : vv[j];
                if (T.ctl)
                    bas_store1_sc1(ye + n + j, a);
                else
                    ye[n + j] = a;
                if (n + j >= plo && n + j < phi) lmax = fmaxf(lmax, fabsf(a));
            }
        }
    }
    if (T.ctl) {
        bas_tail<256>(T, lmax);
    } else if (C.x) {
        if (C.running_peak) bas_block_peak_max(lmax, C.running_peak);
        bas_carry_moves(C, blockIdx.x * 256L + threadIdx.x, (long)gridDim.x * 256L);
    } else if (peak_bits) {
        bas_block_peak_max(lmax, peak_bits);
    }
}

// ---------------------------------------------------------------------------
// peak rule (apply_hrtf.py:462-464)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bas_absmax_kernel(const float *__restrict__ y, long n,
                                                           unsigned int *peak_bits) {
    float lmax = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L)
        lmax = fmaxf(lmax, fabsf(y[i]));
    bas_block_peak_max(lmax, peak_bits);
}

__global__ __launch_bounds__(256) void bas_scale_kernel(float *__restrict__ y, long n,
                                                          const float *__restrict__ peak) {
    const float m = *peak;
    if (!(m > 1.0f)) return;                                 // :463
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) y[i] = y[i] / m;
}

// fixed-order sum of partial mixes (one per GPU) + fused max|y|
__global__ __launch_bounds__(256) void bas_mix_partials_kernel(const float *__restrict__ parts, int n_parts,
                                                                 long part_stride, long n,
                                                                 float *__restrict__ y,
                                                                 unsigned int *peak_bits) {
    float lmax = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        float v = 0.f;
        for (int p = 0; p < n_parts; ++p) v += parts[p * part_stride + i];
        y[i] = v;
        lmax = fmaxf(lmax, fabsf(v));
    }
    if (peak_bits) {
        bas_block_peak_max(lmax, peak_bits);
    }
}

// the same sum, max|y| and the peak rule in ONE launch (bas_tail.h): what the root rank of a multi-GPU group runs on the
// gathered partial mixes (round 3: memset + sum + scale = three launches, ~40 us beside the root's own render)
__global__ __launch_bounds__(256) void bas_mix_finish_kernel(const float *__restrict__ parts, int n_parts,
                                                               long part_stride, long n, float *__restrict__ y, BasTail T) {
    float lmax = 0.f;
    const long n4 = n >> 2;
    // 16 bytes per lane and part whatever the parts' alignment: a gathered mix of T_out = 441 471 samples puts every second
    // part 8 bytes off a 16-byte boundary, which global loads of four dwords do not mind (4-byte alignment is all they ask);
    // round 4's first form fell back to one float per lane there: 14.6 us for eight parts instead of 9.
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < n_parts; ++p) v += *reinterpret_cast<const f32x4_a4 *>(parts + p * part_stride + 4 * i);
        bas_store4_sc1(y + 4 * i, v);
        lmax = fmaxf(lmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (long i = 4 * n4 + blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        float v = 0.f;
        for (int p = 0; p < n_parts; ++p) v += parts[p * part_stride + i];
        bas_store1_sc1(y + i, v);
        lmax = fmaxf(lmax, fabsf(v));
    }
    bas_tail<256>(T, lmax);
}

int bas_grid_for(long items, int cap) {
    long g = (items + 255) / 256;
    if (g < 1) g = 1;
    return (int)(g < cap ? g : cap);
}

// Per-device facts and one-time kernel attributes, cached: bas_render_mix_f32 is called once per block by the
// streaming renderer, so nothing on its path may query the runtime or the environment per call.
#define BAS_MAX_DEVICES 64
static std::atomic<int> g_cus[BAS_MAX_DEVICES];

static int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        dev = 0;
    }
    return dev;
}

int bas_device_cus() {
    const int dev = current_device();
    const bool cached = dev >= 0 && dev < BAS_MAX_DEVICES;
    int cus = cached ? g_cus[dev].load(std::memory_order_relaxed) : 0;
    if (cus > 0) return cus;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        (void)hipGetLastError();
        cus = 256;                                           // MI355X
    }
    if (cached) g_cus[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

// hipFuncAttributeMaxDynamicSharedMemorySize, raised once per (kernel, device) to the whole LDS
hipError_t bas_allow_full_lds(const void *fn) {
    static std::mutex mu;
    static struct { const void *fn; unsigned long long devs; } seen[64];
    static int n_seen = 0;
    const int dev = current_device();
    const unsigned long long bit = 1ull << (dev & 63);
    std::lock_guard<std::mutex> lock(mu);
    int slot = -1;
    for (int i = 0; i < n_seen; ++i)
        if (seen[i].fn == fn) slot = i;
    if (slot >= 0 && (seen[slot].devs & bit)) return hipSuccess;
    // (dynamic + static LDS must fit the CU's 160 KB: the kernel tails of bas_tail.h keep a few words of static LDS)
    hipFuncAttributes fa;
    hipError_t e = hipFuncGetAttributes(&fa, fn);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)fa.sharedSizeBytes);
    if (e != hipSuccess) return e;
    if (slot < 0 && n_seen < 64) {
        slot = n_seen++;
        seen[slot].fn = fn;
        seen[slot].devs = 0;
    }
    if (slot >= 0) seen[slot].devs |= bit;
    return hipSuccess;
}

// slabs of partial tiles -> y (fixed order), fused max|y|: shared by the FIR launchers (bas_fused.hip too).
// tail != null (fused entry points): tail->ctl / wgpeak / y / n / peak / normalize are filled in, the launch geometry here;
// the kernel then ends in bas_tail.  Launches with more workgroups than the control area has maxima slots keep the round-3
// form (returns 1 in *tail_skipped: the caller launches the scale kernel itself).
int bas_launch_slab_reduce(const float *slab, int tile, int n_src, int units_per_wg, int parts_per_wg, int n_wg,
                           long T_out, float *y, int accumulate, unsigned int *peak_bits, const BasTail *tail,
                           int *tail_skipped, const BasCarry *carry, hipStream_t st, const char *what) {
    BasCarry C = {};
    if (carry) {                                             // a stream block: its own maximum, no tail (see the kernels)
        C = *carry;
        tail = nullptr;
        peak_bits = nullptr;
    }
    const int parts = (n_src + units_per_wg - 1) / units_per_wg + 1;       // workgroups that can share one tile
    BasTail T = {};
    if (tail_skipped) *tail_skipped = 1;
    if (parts > 24) {                                        // many sources on a short signal: sum the parts in parallel
        const long blocks_per_ear = (((T_out + 3) / 4) + 3) / 4;
        if (tail && 2 * blocks_per_ear <= BAS_TAIL_MAX_WG) {
            T = *tail;
            T.n_wg = (unsigned)(2 * blocks_per_ear);
            if (tail_skipped) *tail_skipped = 0;
        }
        hipLaunchKernelGGL(bas_slab_reduce_wide_kernel, dim3((unsigned)(2 * blocks_per_ear)), dim3(256), 0, st, slab, tile,
                           n_src, units_per_wg, parts_per_wg, n_wg, T_out, y, accumulate, peak_bits, T, C);
        return bas_check_launch(what);
    }
    const int grid = bas_grid_for((T_out + 3) / 4, 2048);
    if (tail && grid <= BAS_TAIL_MAX_WG) {
        T = *tail;
        T.n_wg = (unsigned)grid;
        if (tail_skipped) *tail_skipped = 0;
    }
    hipLaunchKernelGGL(bas_slab_reduce_kernel, dim3(grid), dim3(256), 0, st, slab, tile,
                       n_src, units_per_wg, parts_per_wg, n_wg, T_out, y, accumulate, peak_bits, T, C);
    return bas_check_launch(what);
}

enum { KIND_GENERIC = 0, KIND_ROWS32 = 1, KIND_HD = 2 };

struct RenderPlan {
    int kind;
    int tile;                  // outputs per tile
    long n_tiles, units_total;
    int n_wg, units_per_wg, parts_per_wg;
    int hd_slots;              // chunk slots of the hd kernel
    int honly;                 // 1: h-only LDS image (small chunks), see hd_load_octet
    int dual;                  // 1: dual row step (subchunk size not a power of two / multiple of 32)
    size_t lds_bytes, slab_bytes;
};

// Which kernel renders (n_src, T_in, K, S, L), and how the (tile, source) work units are dealt to
// workgroups: equal contiguous shares, so every workgroup finishes at the same time.
static RenderPlan plan_render(int n_src, long T_in, int K, int S, int L, bool aligned) {
    RenderPlan p = {};
    p.kind = KIND_GENERIC;
    const bool s_pow2 = (S & (S - 1)) == 0;
    const bool hd_small_s = s_pow2 && S >= 4 && S < 32 && K % 32 == 0;        // rows of 32 hold 2 / 4 / 8 subchunks
                                                                              // (16 / 32 per row build for minutes: not offered)
#ifdef BAS_DIAG
    const char *force = getenv("BAS_FORCE_KERNEL");          // diagnostic build only: tests force the fallback kernels
#else
    const char *force = nullptr;
#endif
    if (!(aligned && n_src > 0 && T_in > 0)) return p;
    if (!(S % 32 == 0 || hd_small_s)) {
        // Any other subchunk size >= 2 (any chunk size >= 32; the caller keeps rows 16-byte aligned through x_stride):
        // the hd kernel's multi-part row step.  One slot more than the rows reach: the part of a row behind a chunk
        // boundary reads the next chunk's slot.
        const int dual_slots = (K - 1 + 32 * (HD_ROWS - 1) + 31) / K + 2;
        const bool full = dual_slots <= HD_MAXSLOTS;
        const size_t lds = full ? (size_t)(HD_X_FLOATS + (dual_slots + 1) * HD_SLOT) * sizeof(float)
                                : (size_t)(HD_X_FLOATS + (dual_slots + 1) * HO_SLOT) * sizeof(float);
        if (S < 2 || K < 32 || lds > 160 * 1024 || (force && strcmp(force, "hd"))) return p;
        p.kind = KIND_HD;
        p.dual = 1;
        p.honly = !full;
        p.hd_slots = dual_slots;
        p.tile = HD_TILE;
        p.lds_bytes = lds;
        long wg_per_cu = (long)(160 * 1024 / lds);
        wg_per_cu = wg_per_cu > 2 ? 2 : (wg_per_cu < 1 ? 1 : wg_per_cu);
        const long T_out = T_in + L - 1;
        p.n_tiles = (T_out + p.tile - 1) / p.tile;
        p.units_total = p.n_tiles * n_src;
        long slots = wg_per_cu * bas_device_cus();
        long wg = p.units_total < slots ? p.units_total : slots;
        p.units_per_wg = (int)((p.units_total + wg - 1) / wg);
        p.n_wg = (int)((p.units_total + p.units_per_wg - 1) / p.units_per_wg);
        p.parts_per_wg = (p.units_per_wg + n_src - 2) / n_src + 1;
        p.slab_bytes = (size_t)p.n_wg * p.parts_per_wg * 2 * p.tile * sizeof(float);
        return p;
    }
    const int hd_slots = (K - 32 + 32 * (HD_ROWS - 1)) / K + 1;
    // hd kernel: up to HD_MAXSLOTS chunk slots per tile (K >= 448) the LDS image holds (h0, d) per tap; smaller
    // chunks put more slots under a tile and use the h-only image (half the bytes per slot, d taken in the row
    // step: two workgroups per CU down to K ~ 192, one below).  On the 256-source scene: 0.89 ms at K = 256,
    // S = 32 (rows32: 1.24 ms), 0.93 ms at K = 256, S = 16 and 1.2 ms at K = 128, S = 16 (generic: 32 ms).
    const bool hd_fits = (s_pow2 || S % 32 == 0) && (hd_slots <= HD_MAXSLOTS ||
                                    (size_t)(HD_X_FLOATS + (hd_slots + 1) * HO_SLOT) * sizeof(float) <= 160 * 1024);
    int kind = hd_fits ? KIND_HD : KIND_ROWS32;
    if (force && !strcmp(force, "rows32")) kind = KIND_ROWS32;
    if (kind == KIND_ROWS32 && S % 32 != 0) return p;        // small subchunks: hd kernel or nothing
    if (force && !strcmp(force, "generic")) return p;
    if (kind == KIND_ROWS32 && L % 2 != 0) return p;         // rows32 loads tap pairs
    long wg_per_cu;
    if (kind == KIND_HD) {
        p.tile = HD_TILE;
        p.hd_slots = hd_slots;
        p.honly = hd_slots > HD_MAXSLOTS;                   // measured faster than the full image at one workgroup per CU
        p.lds_bytes = p.honly ? (size_t)(HD_X_FLOATS + (hd_slots + 1) * HO_SLOT) * sizeof(float)
                              : (size_t)(HD_X_FLOATS + (hd_slots + 1) * HD_SLOT) * sizeof(float);
        wg_per_cu = (long)(160 * 1024 / p.lds_bytes);
        if (wg_per_cu > 2) wg_per_cu = 2;
        if (wg_per_cu < 1) wg_per_cu = 1;
    } else {
        p.tile = RT_TILE;
        p.lds_bytes = RT_LDS_BYTES;
        wg_per_cu = 2;
    }
    p.kind = kind;
    const long T_out = T_in + L - 1;
    p.n_tiles = (T_out + p.tile - 1) / p.tile;
    p.units_total = p.n_tiles * n_src;
    long slots = wg_per_cu * bas_device_cus();
    long wg = p.units_total < slots ? p.units_total : slots;
    p.units_per_wg = (int)((p.units_total + wg - 1) / wg);
    p.n_wg = (int)((p.units_total + p.units_per_wg - 1) / p.units_per_wg);
    p.parts_per_wg = (p.units_per_wg + n_src - 2) / n_src + 1;
    p.slab_bytes = (size_t)p.n_wg * p.parts_per_wg * 2 * p.tile * sizeof(float);
    return p;
}

extern "C" size_t bas_render_workspace_bytes(int n_src, long T_in, int K, int S, int L) {
    if (n_src <= 0 || T_in <= 0 || K <= 0 || S <= 0 || L <= 0) return BAS_WS_HEAD_BYTES + 16;
    // the kernel is chosen at launch time (pointer alignment, diagnostics override): both fast kernels
    // use the same slab formula, so size for the larger tile count of the two
    RenderPlan p = plan_render(n_src, T_in, K, S, L, true);
    size_t need = p.slab_bytes;
    if (p.kind == KIND_HD && L % 2 == 0) {
        RenderPlan q = p;
        const long T_out = T_in + L - 1;
        q.tile = RT_TILE;
        q.n_tiles = (T_out + RT_TILE - 1) / RT_TILE;
        q.units_total = q.n_tiles * n_src;
        long slots = 2L * bas_device_cus();
        long wg = q.units_total < slots ? q.units_total : slots;
        q.units_per_wg = (int)((q.units_total + wg - 1) / wg);
        q.n_wg = (int)((q.units_total + q.units_per_wg - 1) / q.units_per_wg);
        q.parts_per_wg = (q.units_per_wg + n_src - 2) / n_src + 1;
        size_t alt = (size_t)q.n_wg * q.parts_per_wg * 2 * RT_TILE * sizeof(float);
        if (alt > need) need = alt;
    }
    return BAS_WS_HEAD_BYTES + need + 16;                    // (the head is the library's control area: never slab space)
}

extern "C" const char *bas_render_kernel_name(int n_src, long T_in, int K, int S, int L) {
    if (n_src <= 0 || T_in <= 0 || K <= 0 || S <= 0 || L <= 0) return "bas_render_generic_kernel";
    RenderPlan p = plan_render(n_src, T_in, K, S, L, true);
    return p.kind == KIND_HD ? "bas_render_hd_kernel"
                             : (p.kind == KIND_ROWS32 ? "bas_render_rows32_kernel" : "bas_render_generic_kernel");
}

static int render_mix_impl(const float *x, long x_stride, const float *H, int n_src, long T_in, int K, int S,
                           int L, float *y, int accumulate, float *peak, void *ws, size_t ws_bytes,
                           bas_stream_t stream, hipEvent_t ev_begin, hipEvent_t ev_end) {
    BAS_REQUIRE(y, BAS_E_NULL, "bas_render_mix_f32: y is null");
    BAS_REQUIRE(n_src >= 0 && T_in >= 0 && K > 0 && S > 0 && L > 0, BAS_E_SHAPE,
                "bas_render_mix_f32: need n_src>=0, T_in>=0, K,S,L>0 (n_src=%d T_in=%ld K=%d S=%d L=%d)",
                n_src, T_in, K, S, L);
    BAS_REQUIRE(K % S == 0, BAS_E_SHAPE,
                "bas_render_mix_f32: subchunksize does not divide chunksize evenly (K=%d S=%d)", K, S);
    BAS_REQUIRE(T_in % K == 0, BAS_E_SHAPE, "bas_render_mix_f32: T_in (%ld) must be a multiple of K (%d)", T_in,
                K);
    BAS_REQUIRE(n_src == 0 || T_in == 0 || (x && H), BAS_E_NULL, "bas_render_mix_f32: x or H is null");
    BAS_REQUIRE(n_src == 0 || x_stride >= T_in, BAS_E_SHAPE, "bas_render_mix_f32: x_stride < T_in");
    BAS_REQUIRE(T_in / K < (1L << 30), BAS_E_SHAPE, "bas_render_mix_f32: too many chunks");
    hipStream_t st = bas_stream(stream);
    const long T_out = T_in + L - 1;
    unsigned int *peak_bits = reinterpret_cast<unsigned int *>(peak);
    if (peak) {
        hipError_t e = hipMemsetAsync(peak, 0, sizeof(float), st);
        if (e != hipSuccess) return bas_fail((int)e, "bas_render_mix_f32: hipMemsetAsync: %s", hipGetErrorString(e));
    }
    const int n_chunks = (int)(T_in / K);
    const bool aligned = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (x_stride % 4 == 0) &&
                         (reinterpret_cast<uintptr_t>(ws) % 16 == 0) && (reinterpret_cast<uintptr_t>(H) % 8 == 0);
    const int live_src = T_in == 0 ? 0 : n_src;
    RenderPlan p = plan_render(live_src, T_in, K, S, L, aligned);
    if (p.kind == KIND_GENERIC) {
        if (ev_begin) (void)hipEventRecord(ev_begin, st);
        hipLaunchKernelGGL(bas_render_generic_kernel, dim3(bas_grid_for(T_out, 8192)), dim3(256), 0, st, x, x_stride,
                           H, live_src, T_in, K, S, L, n_chunks, T_out, y, accumulate, peak_bits);
        if (ev_end) (void)hipEventRecord(ev_end, st);
        return bas_check_launch("bas_render_mix_f32(generic)");
    }
    BAS_REQUIRE(p.units_total < (1L << 31) - 65536, BAS_E_SHAPE,
                "bas_render_mix_f32: %ld (tile, source) work units exceed 2^31: render in blocks", p.units_total);
    BAS_REQUIRE(ws && ws_bytes >= BAS_WS_HEAD_BYTES + p.slab_bytes, BAS_E_WORKSPACE,
                "bas_render_mix_f32: workspace of %zu bytes needed, %zu given", (size_t)BAS_WS_HEAD_BYTES + p.slab_bytes, ws_bytes);
    RenderArgs A;
    A.x = x; A.x_stride = x_stride; A.H = H; A.n_src = live_src; A.T_in = T_in;
    A.K = K; A.S = S; A.L = L; A.Lp = (L + 7) & ~7; A.n_chunks = n_chunks;
    A.units_total = p.units_total; A.units_per_wg = p.units_per_wg; A.parts_per_wg = p.parts_per_wg;
    A.slab = reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + BAS_WS_HEAD_BYTES);   // (behind the control area)
#ifdef BAS_DIAG
    { const char *d = getenv("BAS_DEBUG_FLAGS"); A.dbg = d ? atoi(d) : 0; }
#endif
    const int nsub = S >= 32 ? 1 : 32 / S;
    typedef void (*hd_fn)(RenderArgs, int);
    auto pick = [&](bool honly) -> hd_fn {
        switch (nsub) {
        case 1: return honly ? bas_render_hd_kernel<1, true> : bas_render_hd_kernel<1>;
        case 2: return honly ? bas_render_hd_kernel<2, true> : bas_render_hd_kernel<2>;
        case 4: return honly ? bas_render_hd_kernel<4, true> : bas_render_hd_kernel<4>;
        default: return honly ? bas_render_hd_kernel<8, true> : bas_render_hd_kernel<8>;
        }
    };
    hd_fn hdk = pick(false);
    if (p.kind == KIND_HD && p.dual)
        hdk = p.honly ? bas_render_hd_kernel<1, true, true> : bas_render_hd_kernel<1, false, true>;
    else if (p.kind == KIND_HD && p.honly)
        hdk = pick(true);
    const void *fn = p.kind == KIND_HD ? reinterpret_cast<const void *>(hdk)
                                        : reinterpret_cast<const void *>(bas_render_rows32_kernel);
    hipError_t e = bas_allow_full_lds(fn);
    if (e != hipSuccess) return bas_fail((int)e, "bas_render_mix_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    if (ev_begin) (void)hipEventRecord(ev_begin, st);
    if (p.kind == KIND_HD)
        hipLaunchKernelGGL(hdk, dim3(p.n_wg), dim3(HD_THREADS), p.lds_bytes, st, A, p.hd_slots);
    else
        hipLaunchKernelGGL(bas_render_rows32_kernel, dim3(p.n_wg), dim3(RT_THREADS), p.lds_bytes, st, A);
    if (ev_end) (void)hipEventRecord(ev_end, st);
    int rc = bas_check_launch(p.kind == KIND_HD ? "bas_render_mix_f32(hd)" : "bas_render_mix_f32(rows32)");
    if (rc) return rc;
    return bas_launch_slab_reduce(A.slab, p.tile, live_src, p.units_per_wg, p.parts_per_wg, p.n_wg, T_out, y, accumulate,
                                  peak_bits, nullptr, nullptr, nullptr, st, "bas_render_mix_f32(reduce)");
}

extern "C" int bas_render_mix_f32(const float *x, long x_stride, const float *H, int n_src, long T_in,
                                  int K, int S, int L, float *y, int accumulate, float *peak, void *ws,
                                  size_t ws_bytes, bas_stream_t stream) {
    return render_mix_impl(x, x_stride, H, n_src, T_in, K, S, L, y, accumulate, peak, ws, ws_bytes, stream,
                           nullptr, nullptr);
}

extern "C" int bas_render_mix_profiled_f32(const float *x, long x_stride, const float *H, int n_src, long T_in,
                                           int K, int S, int L, float *y, int accumulate, float *peak,
                                           void *ws, size_t ws_bytes, bas_stream_t stream, void *ev_begin,
                                           void *ev_end) {
    return render_mix_impl(x, x_stride, H, n_src, T_in, K, S, L, y, accumulate, peak, ws, ws_bytes, stream,
                           reinterpret_cast<hipEvent_t>(ev_begin), reinterpret_cast<hipEvent_t>(ev_end));
}

extern "C" int bas_peak_normalize_f32(float *y, long n, float *peak, int apply, bas_stream_t stream) {
    BAS_REQUIRE(y || n == 0, BAS_E_NULL, "bas_peak_normalize_f32: y is null");
    BAS_REQUIRE(peak, BAS_E_NULL, "bas_peak_normalize_f32: peak (device float) is null");
    BAS_REQUIRE(n >= 0, BAS_E_SHAPE, "bas_peak_normalize_f32: n < 0");
    hipStream_t st = bas_stream(stream);
    hipError_t e = hipMemsetAsync(peak, 0, sizeof(float), st);
    if (e != hipSuccess) return bas_fail((int)e, "bas_peak_normalize_f32: hipMemsetAsync: %s", hipGetErrorString(e));
    if (n == 0) return 0;
    hipLaunchKernelGGL(bas_absmax_kernel, dim3(bas_grid_for(n, 2048)), dim3(256), 0, st, y, n,
                       reinterpret_cast<unsigned int *>(peak));
    int rc = bas_check_launch("bas_peak_normalize_f32(absmax)");
    if (rc || !apply) return rc;
    return bas_scale_by_peak_f32(y, n, peak, stream);
}

extern "C" int bas_scale_by_peak_f32(float *y, long n, const float *peak, bas_stream_t stream) {
    BAS_REQUIRE(peak && (y || n == 0), BAS_E_NULL, "bas_scale_by_peak_f32: null pointer");
    BAS_REQUIRE(n >= 0, BAS_E_SHAPE, "bas_scale_by_peak_f32: n < 0");
    if (n == 0) return 0;
    hipLaunchKernelGGL(bas_scale_kernel, dim3(bas_grid_for(n, 512)), dim3(256), 0, bas_stream(stream), y, n, peak);
    return bas_check_launch("bas_scale_by_peak_f32");
}

extern "C" int bas_mix_partials_f32(const float *parts, int n_parts, long part_stride, long n, float *y,
                                    float *peak, bas_stream_t stream) {
    BAS_REQUIRE(y || n == 0, BAS_E_NULL, "bas_mix_partials_f32: y is null");
    BAS_REQUIRE(parts || n_parts == 0 || n == 0, BAS_E_NULL, "bas_mix_partials_f32: parts is null");
    BAS_REQUIRE(n >= 0 && n_parts >= 0 && part_stride >= n, BAS_E_SHAPE,
                "bas_mix_partials_f32: need n>=0, n_parts>=0, part_stride>=n");
    hipStream_t st = bas_stream(stream);
    if (peak) {
        hipError_t e = hipMemsetAsync(peak, 0, sizeof(float), st);
        if (e != hipSuccess) return bas_fail((int)e, "bas_mix_partials_f32: hipMemsetAsync: %s", hipGetErrorString(e));
    }
    if (n == 0) return 0;
    hipLaunchKernelGGL(bas_mix_partials_kernel, dim3(bas_grid_for(n, 2048)), dim3(256), 0, st, parts, n_parts,
                       part_stride, n, y, reinterpret_cast<unsigned int *>(peak));
    return bas_check_launch("bas_mix_partials_f32");
}

extern "C" size_t bas_mix_workspace_bytes(void) { return BAS_WS_HEAD_BYTES; }

extern "C" int bas_mix_finish_f32(const float *parts, int n_parts, long part_stride, long n, float *y, float *peak,
                                  int normalize, void *ws, size_t ws_bytes, bas_stream_t stream) {
    BAS_REQUIRE(y || n == 0, BAS_E_NULL, "bas_mix_finish_f32: y is null");
    BAS_REQUIRE(parts || n_parts == 0 || n == 0, BAS_E_NULL, "bas_mix_finish_f32: parts is null");
    BAS_REQUIRE(n >= 0 && n_parts >= 0 && part_stride >= n, BAS_E_SHAPE,
                "bas_mix_finish_f32: need n>=0, n_parts>=0, part_stride>=n");
    BAS_REQUIRE(ws && ws_bytes >= BAS_WS_HEAD_BYTES, BAS_E_WORKSPACE,
                "bas_mix_finish_f32: workspace of %zu bytes needed (bas_mix_workspace_bytes), %zu given",
                (size_t)BAS_WS_HEAD_BYTES, ws_bytes);
    BAS_REQUIRE(reinterpret_cast<uintptr_t>(ws) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0, BAS_E_ALIGN,
                "bas_mix_finish_f32: y and ws must be 16-byte aligned");
    hipStream_t st = bas_stream(stream);
    if (n == 0) {
        if (peak) {
            hipError_t e = hipMemsetAsync(peak, 0, sizeof(float), st);
            if (e != hipSuccess) return bas_fail((int)e, "bas_mix_finish_f32: hipMemsetAsync: %s", hipGetErrorString(e));
        }
        return 0;
    }
    BasTail T = {};
    T.ctl = reinterpret_cast<unsigned *>(ws);
    T.wgpeak = reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + BAS_CTL_WORDS * 4);
    T.y = y; T.n = n; T.peak = peak; T.normalize = normalize ? 1 : 0;
    const int grid = bas_grid_for((n + 3) / 4, 2048);
    T.n_wg = (unsigned)grid;
    hipLaunchKernelGGL(bas_mix_finish_kernel, dim3(grid), dim3(256), 0, st, parts, n_parts, part_stride, n, y, T);
    return bas_check_launch("bas_mix_finish_f32");
}

// Device-side error record of the workspace's control block (bas_tail.h): synchronises the stream, reads four words.
extern "C" int bas_render_status(void *ws, size_t ws_bytes, bas_stream_t stream) {
    BAS_REQUIRE(ws && ws_bytes >= BAS_WS_HEAD_BYTES, BAS_E_WORKSPACE, "bas_render_status: not a workspace of this library");
    hipStream_t st = bas_stream(stream);
    unsigned rec[4] = {0, 0, 0, 0};
    unsigned *dev = reinterpret_cast<unsigned *>(ws) + BAS_CTL_STATUS;
    hipError_t e = hipMemcpyAsync(rec, dev, sizeof(rec), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return bas_fail((int)e, "bas_render_status: %s", hipGetErrorString(e));
    if (rec[0] != BAS_STATUS_MAGIC0 || rec[1] != BAS_STATUS_MAGIC1) return 0;
    (void)hipMemsetAsync(reinterpret_cast<unsigned *>(ws), 0, BAS_CTL_WORDS * 4, st);   // whole control block: usable again
    (void)hipStreamSynchronize(st);
    return bas_fail((int)hipErrorLaunchTimeOut,
                    rec[2] == BAS_STATUS_HANDOVER_TIMEOUT
                        ? "device-side error: a stager wave of workgroup %u never received its neighbour's boundary chunk IR "
                          "(hand-over timeout); the audio of that launch holds NaN where it happened"
                        : "device-side error: a late workgroup (%u) of a kernel tail never saw the others arrive; peak rule not applied",
                    rec[3]);
}

#ifdef BAS_STAMPS
extern "C" int bas_debug_read_stamps(unsigned long long *host, size_t count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bas_dbg_stamps), count * sizeof(unsigned long long));
}
#endif
