// a6 + a7 + a8 in one kernel (gfx950): interpolate_2d of every chunk IR a tile needs (apply_hrtf.py:219-279),
// per-subchunk crossfade (:442-443), direct FIR (:445-446), overlap-add and the mix over sources (:450-453).
// The [n_src][n_chunks+1][2][L] chunk-IR array of the two-kernel path never exists in HBM.
//
// Work split and FIR: as bas_render_hd_kernel (bas_render.hip) - a workgroup of NW waves owns a tile of
// 2048 NW outputs and walks over (tile, source) units with the mix in registers; lane = one row of 32 outputs;
// row step = 32 x 32 Toeplitz block of packed FMAs on an x row and 64 (h0, d) taps from LDS, evaluated as a
// 2-parallel fast FIR (three half-rate products instead of four: bas_fir.h).
//
// What is different here
//   * chunk IRs come from read plans with precomputed byte offsets (EarPlanS, bas_plan.h).  Wave w evaluates
//     a contiguous range of chunk IRs: it copies their plans (144 bytes per ear) into its own LDS region, then
//     per chunk IR lanes 0-31 evaluate four adjacent taps of the left ear, lanes 32-63 of the right ear: every
//     plan value reaches the lanes of "its" half-wave as a broadcast LDS read, so one IR costs 12 offset ops +
//     16 address adds + 32 packed FMAs and 16 sixteen-byte table reads per lane (the plan-in-lanes form of
//     round 1 needed ~195 vector instructions per IR for shuffles and per-read plane selection).
//   * the wave keeps its chunk IRs in registers, forms d = H_{c+1} - H_c itself, regroups (left, right) with
//     v_permlane32_swap and stores (h0_L, h0_R, d_L, d_R) with 16-byte LDS writes: no second pass over the
//     LDS image, one barrier less per pass.
//   * (tile, source, tap segment) advance by scalar counters and the window's chunk arithmetic is redone only
//     when the tile changes: no 64-bit divisions per pass; the per-lane chunk slot comes from a float estimate.
//   * NW = 4 or 1 waves per workgroup (tile 8192 / 2048): scenes with few sources (BASELINE configs 2 and 3:
//     ONE source) get four times the workgroups out of the same signal.
//   * chunk sizes 256 .. 447 (template parameter HONLY): LDS rows of (h_L, h_R) only, d formed in the row step,
//     the waves evaluate disjoint sets of chunk IRs.
// No MFMA: this is a 1-D FIR (BASELINE.json north_star).
#include "bas_internal.h"
#include "bas_plan.h"
#include "bas_fir.h"
#include "bas_fused.h"
#include <stdlib.h>

#ifdef BAS_STAMPS
// Diagnostic build only (make stamps): per-wave totals of the pass phases in 10 ns ticks (s_memrealtime).
__device__ unsigned long long bas_fz_stamps[2048 * 4 * 8];
#ifdef BAS_LIFETIME_ONLY      // only the wave's first and last instruction are stamped: no waits added inside a pass
#define FZ_STAMP(var)
#define FZ_STAMP_NW(var)
#define FZ_ADD(slot, a, b)
#else
#define FZ_STAMP(var) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#define FZ_STAMP_NW(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#define FZ_ADD(slot, a, b) st_acc[slot] += (b) - (a)
#endif
#else
#define FZ_STAMP(var)
#define FZ_STAMP_NW(var)
#define FZ_ADD(slot, a, b)
#endif

#ifndef FZ_STAGE_PRIO
#define FZ_STAGE_PRIO 3         // wave priority while staging (0: leave it alone)
#endif
#ifndef FZ_SLICE_SHIFT
#define FZ_SLICE_SHIFT 14       // FIR priority alternates between the two workgroups of a CU every 2^shift x 10 ns (0: off)
#endif
#ifndef FZ_START_DELAY
#define FZ_START_DELAY 0        // x 10 ns: head start of the first workgroup of every CU (0: none)
#endif
#ifndef FZ_NT_LOADS
#define FZ_NT_LOADS 1           // x window and plans are read once: non-temporal, so that L2 keeps the table (same speed,
#endif                          // 13 % fewer bytes fetched past L2)
#ifndef FZ_NT_STORES
#define FZ_NT_STORES 0          // non-temporal slab stores: WRITE_SIZE 40 -> 135 MB per launch - off
#endif
#ifndef FZ_FFA
#define FZ_FFA 1                // row step as a 2-parallel fast FIR (bas_fir.h: 3/4 of the FMAs); 0: direct form
#endif
#ifndef FZ_SPLIT
#define FZ_SPLIT FZ_ASM            // scenes with more than one (tile of 8192, source) unit per CU: the split-role kernel (bas_fused_split.hip)
#endif
#ifndef FZ_QUAD
#define FZ_QUAD FZ_ASM             // small scenes: four waves per tile of 2048 (bas_fused_quad.hip) ...
#endif
#ifndef FZ_QUAD_ROUNDS
#define FZ_QUAD_ROUNDS 3           // ... whose tiles fit in this many rounds of 2 workgroups per CU
#endif
#ifndef FZ_SPLIT_MIN_UNITS
#define FZ_SPLIT_MIN_UNITS 1    // ... from MORE than this many units per CU on: some workgroup then has two units and the second one's staging
#endif                          // runs under the first one's FIR (profiles/r03b_ab_split_threshold.txt: 5-9 sources -3 .. -5 %, 10-14 -22 %)
#ifndef FZ_PREFETCH
#define FZ_PREFETCH 1           // the x window and the read plans of pass i + 1 are requested in front of the FIR phase of pass i and
#endif                          // travel under it (44 registers per lane held across the row steps); 0: requested at the pass's top
#define FZ_HO_MAXEV 9           // h-only rows: chunk IRs per wave (K = 256: 35 rows under a tile of 8192)
#define FZ_MAXSLOTS 20          // chunk slots under one tile (LDS: two 4-wave workgroups per CU at K >= 448)

template <int XR>
__device__ __forceinline__ void fz_load_xrow(float (&xr)[32], const f32x4 *__restrict__ xrow) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4 v = xrow[c * XR];
        xr[4 * c] = v.x; xr[4 * c + 1] = v.y; xr[4 * c + 2] = v.z; xr[4 * c + 3] = v.w;
    }
}

// HONLY (chunk sizes below ~448: more than FZ_MAXSLOTS chunk slots under a tile): the LDS rows hold (h_L, h_R) of the
// nslots + 1 chunk boundaries only and the row step takes d = H_{c+1} - H_c from two rows (bas_fir.h): half the LDS per
// chunk, one more packed op per tap, and the waves evaluate disjoint sets of chunk IRs (no boundary IR twice).
template <int NW, bool HONLY>
__global__ __launch_bounds__(64 * NW, 2) void bas_render_fz_kernel(
    FzArgs A, const float *__restrict__ x,                   // [n_src] rows of T_in floats, stride A.x_stride
    float *__restrict__ slab,                                // [n_wg][parts_per_wg][2][tile]
    const float *__restrict__ packed,                        // table in phase-plane layout
    const unsigned *__restrict__ plans,                      // [n_src][n_chunks+1][2 ears][BAS_PLANS_WORDS]
    float *__restrict__ y, unsigned int *__restrict__ peak_bits) {   // direct output only: [2][T_out], max|y| bits
    constexpr int THREADS = 64 * NW;
    constexpr int TILE = 2048 * NW;
    constexpr int ROWS = TILE / 32 + HD_HALO;                // rows of 32 inputs in the x window
    constexpr int XR = ROWS + 1;                             // odd: conflict-free column-major image
    constexpr int NX = (ROWS * 8 + THREADS - 1) / THREADS;   // float4 of x per thread (9)
    constexpr int XFLOATS = 8 * XR * 4;
    constexpr int MAXEV = HONLY ? FZ_HO_MAXEV : (NW == 4 ? 6 : 7);   // chunk IRs one wave evaluates (its slots + 1; HONLY: its share)
    constexpr int SLOTF = HONLY ? HO_SLOT : HD_SLOT;         // floats per LDS row of taps
    constexpr int PL4 = 2 * BAS_PLANS_WORDS / 4;             // float4 per chunk IR's pair of plans (18)
    constexpr int NPV = (MAXEV * PL4 + 63) / 64;             // 16-byte plan pieces per lane
    static_assert(XR % 2 == 1, "x image rows must be odd");
    static_assert(NW != 1 || MAXEV * 2 * BAS_PLANS_WORDS <= HD_SLOT, "NW = 1: the plans must fit the last chunk slot");
    static_assert(!HONLY || NW == 4, "h-only rows: four-wave workgroups only");
    extern __shared__ f32x4 lds4[];
    f32x4 *xs4 = lds4;                                       // [8][XR] float4
    float *hd = reinterpret_cast<float *>(lds4) + XFLOATS;   // [nslots][HD_SLOT]: (h0_L, h0_R, d_L, d_R) per tap
                                                             // (HONLY: [nslots + 1][HO_SLOT]: (h_L, h_R) per tap)

    const int tid0 = threadIdx.x;
    const int lane0 = tid0 & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);

    const long unit0 = (long)blockIdx.x * A.units_per_wg;
    long unit1 = unit0 + A.units_per_wg;
    if (unit1 > A.units_total) unit1 = A.units_total;
    const int nseg = (A.Lp + RT_SEG - 1) / RT_SEG;
    const long n_pass = (unit1 - unit0) * nseg;
    if (!A.direct && peak_bits && blockIdx.x == 0 && tid0 == 0) *peak_bits = 0u;   // the reduce kernel maxes into it later
    if (n_pass <= 0) {
        if (A.direct && A.tail_mode) bas_tail<THREADS>(fz_tail(A, y, peak_bits), 0.f);
        return;
    }

    constexpr bool USE_ASM = FZ_ASM && FZ_FFA && !HONLY;
    static_assert(!USE_ASM || XR == 261 || XR == 69, "bas_fir_asm.inc holds the row step for these x-image strides");
#if FZ_FFA
    // half-rate partial sums (bas_fir.h), combined in flush().  Assembly row step: the same 49 pairs as three 32-float
    // vectors + one pair, pinned to v[0:97] by the asm statement's register constraints (A = fa, B = fb[0..15], P = fp)
    f32x2 fa[16], fb[17], fp[16];
    f32x32 accA, accB, accP;
    f32x2 accB16 = f32x2{0.f, 0.f};
    if constexpr (USE_ASM) {
#pragma unroll
        for (int i = 0; i < 32; ++i) accA[i] = accB[i] = accP[i] = 0.f;
    } else {
        ffa_zero(fa, fb, fp);
    }
#define FZ_ACC_CLEAR()                                                               \
    do {                                                                             \
        if constexpr (USE_ASM) {                                                     \
            _Pragma("unroll") for (int i = 0; i < 32; ++i) accA[i] = accB[i] = accP[i] = 0.f; \
            accB16 = f32x2{0.f, 0.f};                                                \
        } else {                                                                     \
            ffa_zero(fa, fb, fp);                                                    \
        }                                                                            \
    } while (0)
#else
    f32x2 acc[32];
#define FZ_ACC_CLEAR()                                        \
    _Pragma("unroll") for (int o = 0; o < 32; ++o) acc[o] = f32x2{0.f, 0.f}
    FZ_ACC_CLEAR();
#endif

    const long first_tile = unit0 / A.n_src;
    float *slab_wg = slab + (long)blockIdx.x * A.parts_per_wg * 2 * TILE;
    unsigned wmax_bits = 0u;                                 // direct output with a tail: max|y| this wave has stored (uniform)
    const unsigned prio_flip = blockIdx.x >= (gridDim.x >> 1) ? 1u : 0u;
    const __amdgpu_buffer_rsrc_t tab =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(packed), 0, (int)A.packed_bytes, 0x00020000);
    const unsigned L4 = 4u * (unsigned)A.L;

    auto flush = [&](long tile) {
#if FZ_FFA
        f32x2 acc[32];
        if constexpr (USE_ASM) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {                   // y[2p] = A[p] + B[p-1];  y[2p+1] = P[p] - A[p] - B[p]
                const f32x2 a = f32x2{accA[2 * p], accA[2 * p + 1]}, b0 = f32x2{accB[2 * p], accB[2 * p + 1]};
                const f32x2 b1 = p < 15 ? f32x2{accB[2 * p + 2], accB[2 * p + 3]} : accB16;
                const f32x2 pp = f32x2{accP[2 * p], accP[2 * p + 1]};
                acc[2 * p] = a + b0;
                acc[2 * p + 1] = (pp - a) - b1;
            }
        } else {
            ffa_combine(acc, fa, fb, fp);
        }
#endif
        if (A.direct) {                                      // uniform
            const long n0 = tile * TILE + 2048 * wv + 32 * lane0;       // this lane's first output
            const float lmax = fz_store_row_direct(acc, y, A.T_out, n0, A.accumulate);
            if (A.tail_mode) {
                const unsigned b = fz_wave_max_bits(lmax);
                wmax_bits = b > wmax_bits ? b : wmax_bits;
            } else if (peak_bits) {
                bas_wave_peak_max(lmax, peak_bits);
            }
            FZ_ACC_CLEAR();
            return;
        }
        float *dst = slab_wg + (tile - first_tile) * 2 * TILE + 2048 * wv + 32 * lane0;
        f32x4 *l4 = reinterpret_cast<f32x4 *>(dst);
        f32x4 *r4 = reinterpret_cast<f32x4 *>(dst + TILE);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#if FZ_NT_STORES
            __builtin_nontemporal_store(f32x4{acc[4 * i].x, acc[4 * i + 1].x, acc[4 * i + 2].x, acc[4 * i + 3].x}, l4 + i);
            __builtin_nontemporal_store(f32x4{acc[4 * i].y, acc[4 * i + 1].y, acc[4 * i + 2].y, acc[4 * i + 3].y}, r4 + i);
#else
            l4[i] = f32x4{acc[4 * i].x, acc[4 * i + 1].x, acc[4 * i + 2].x, acc[4 * i + 3].x};
            r4[i] = f32x4{acc[4 * i].y, acc[4 * i + 1].y, acc[4 * i + 2].y, acc[4 * i + 3].y};
#endif
        }
        FZ_ACC_CLEAR();
    };

    // scalar state of the walk over (tile, source, tap segment): advanced by counters, never re-divided
    long tile = first_tile;
    int s = (int)(unit0 - first_tile * A.n_src);
    int sg = 0;
    // window geometry of a (tile, segment); recomputed only when either changes
    struct Geo {
        int seg0, Lseg, halo, nrows, c0, mo0;
        long xbase;
    };
    auto make_geo = [&](long t, int g) {
        Geo G;
        G.seg0 = g * RT_SEG;
        G.Lseg = A.Lp - G.seg0 < RT_SEG ? A.Lp - G.seg0 : RT_SEG;
        G.halo = (G.Lseg + 31) >> 5;                         // input rows above the tile that matter
        G.xbase = t * TILE - G.seg0 - 32L * G.halo;          // first input sample in LDS (multiple of 32)
        G.nrows = TILE / 32 + G.halo;
        long cf = G.xbase / A.K;                             // floor division, consistent across 0
        if (cf * A.K > G.xbase) --cf;
        G.c0 = (int)cf;
        G.mo0 = (int)(G.xbase - cf * A.K);
        return G;
    };
    Geo G = make_geo(tile, sg);
    long open_tile = -1;                                     // tile whose partial sums the accumulators hold
    // this wave's share of the chunk IRs.  HONLY: rows [slot_a, slot_b) of the nslots + 1 (h_L, h_R) rows, one chunk IR
    // each.  (h0, d) slots: the nslots + 1 chunk IRs are dealt in balanced contiguous ranges [slot_a, slot_b) - 5, 5, 4, 4
    // for the 18 IRs under a tile of 8192 at K = 512 - and a wave writes the slots [slot_a, slot_b): slot i holds
    // (IR i, IR i+1 - IR i), so its LAST slot needs the FIRST IR of the next wave, which that wave leaves in LDS as soon
    // as it has it (bnd / flags below; no IR is evaluated twice: 18 per pass where round 2 evaluated 21, and the
    // longest chain a wave evaluates is 5 IRs instead of 6).
    const int nrows_h = A.nslots + (HONLY ? 1 : 0);
    int slot_a, slot_b;
    if (HONLY) {
        slot_a = wv * A.spw;
        slot_b = slot_a + A.spw < nrows_h ? slot_a + A.spw : nrows_h;
    } else {
        const int n_ir = A.nslots + 1, base = n_ir / NW, rem = n_ir - base * NW;
        slot_a = wv * base + (wv < rem ? wv : rem);
        slot_b = slot_a + base + (wv < rem ? 1 : 0);
    }
#ifdef FZ_NO_EVAL          // ablation (never shipped): no chunk-IR evaluation at all - the FIR runs on whatever the LDS holds
    const int n_ev = 0;
#else
    const int n_ev = slot_a < slot_b ? slot_b - slot_a : 0;      // chunk IRs this wave evaluates per pass
#endif
    const bool need_next = !HONLY && NW > 1 && n_ev > 0 && slot_b <= A.nslots;   // slot slot_b - 1 exists: it needs IR slot_b

#if FZ_START_DELAY
    // The two workgroups of a CU alternate between a latency-bound staging phase and a VALU-bound FIR phase.
    // Started together they stage together (VALU idle) and then share the FIR phase; started half a pass apart,
    // one stages while the other has the SIMDs to itself.  The second half of the grid (the blocks that join the
    // first half's CUs under the observed dispatch order; speed only) waits FZ_START_DELAY x 10 ns before its
    // first pass.
    if (prio_flip) {
        const unsigned long long t_go = __builtin_amdgcn_s_memrealtime() + FZ_START_DELAY;
        while (__builtin_amdgcn_s_memrealtime() < t_go) __builtin_amdgcn_s_sleep(32);
    }
#endif
#ifdef BAS_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long st_begin = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (!HONLY && NW > 1) {                        // hand-over flags of the boundary IRs: no pass has id 0
        f32x4 *pl_base0 = reinterpret_cast<f32x4 *>(hd + ((nrows_h * SLOTF + 3) & ~3));
        unsigned *flags0 = reinterpret_cast<unsigned *>(pl_base0 + NW * (MAXEV * PL4) + NW * 64);
        if (tid0 < NW) flags0[tid0] = 0u;                      // (ordered before their first use by the first pass's barrier)
    }
    // ---- global -> registers: this wave's read plans (chunk IRs slot_a .. slot_b, both ears) and the x window of a pass
    auto plan_src = [&](int lane, int r, const Geo &Gq, int src) {
        const f32x4 *pl_src = reinterpret_cast<const f32x4 *>(plans) + (long)src * (A.n_chunks + 1) * PL4;
        int p = lane + 64 * r;                               // 16-byte piece of the wave's n_ev * 18
        p = p < n_ev * PL4 ? p : 0;
        const int i = (p * 3641) >> 16;                      // p / 18 for p < 1000
        const int c = clampi(Gq.c0 + slot_a + i, 0, A.n_chunks);
        return pl_src + (long)c * PL4 + (p - i * PL4);
    };
    auto issue_plans = [&](int lane, const Geo &Gq, int src, f32x4 (&pv)[NPV]) {
#pragma unroll
        for (int r = 0; r < NPV; ++r) {
#if FZ_NT_LOADS
            pv[r] = __builtin_nontemporal_load(plan_src(lane, r, Gq, src));
#else
            pv[r] = *plan_src(lane, r, Gq, src);
#endif
        }
    };
    auto issue_x = [&](int tid, const Geo &Gq, int src, f32x4 (&xv)[NX]) {
        const float *xwin = x + (long)src * A.x_stride + Gq.xbase;
        const long lo_l = -Gq.xbase, hi_l = A.T_in - Gq.xbase;
        const int x_lo = lo_l < -(1 << 30) ? -(1 << 30) : (lo_l > (1 << 30) ? (1 << 30) : (int)lo_l);
        const int x_hi = hi_l < -(1 << 30) ? -(1 << 30) : (hi_l > (1 << 30) ? (1 << 30) : (int)hi_l);
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            int i = 4 * (tid + j * THREADS);
            i = i < x_lo ? x_lo : i;
            i = i > x_hi - 4 ? x_hi - 4 : i;                 // clamped into the row (T_in is a multiple of K >= 32)
#if FZ_NT_LOADS
            xv[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(xwin + i));   // streamed once: keep L2 for the table
#else
            xv[j] = *reinterpret_cast<const f32x4 *>(xwin + i);
#endif
        }
    };
    // prefetch: the NEXT pass's x window is requested in front of the FIR phase and held in 36 registers per lane across the
    // row steps - only the assembly row step (78 operand registers against ~104 of hipcc's schedule) leaves them: with
    // hipcc's row step the same source spills 166 registers; its read plans go by LDS-DMA straight into the wave's plan
    // region, which is dead during the FIR phase (no registers at all)
    constexpr bool PREFETCH = FZ_PREFETCH && USE_ASM && NW == 4;
    f32x4 pv[NPV], xv[NX];
    for (long pid = 0; pid < n_pass; ++pid) {
        // the thread index is made opaque once per pass: everything derived from it (load / LDS addresses, tap indices, masks) is
        // then recomputed per pass - a few dozen integer ops - instead of being hoisted out of the loop and held in (or spilled
        // from) registers across the FIR phase, whose row step needs them all
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        FZ_STAMP_NW(t0);
        if (tile != open_tile) {                             // the accumulators belong to another tile: hand them over
            if (open_tile >= 0) flush(open_tile);
            open_tile = tile;
        }
        const int seg0 = G.seg0, Lseg = G.Lseg, halo = G.halo, nrows = G.nrows, mo0 = G.mo0;
        const long xbase = G.xbase;

#if FZ_STAGE_PRIO
        __builtin_amdgcn_s_setprio(FZ_STAGE_PRIO);           // staging is latency bound: its few instructions go first
#endif
        const bool fetched = PREFETCH && pid > 0;           // (requested in front of the previous FIR phase)
        if (!fetched) {
            issue_plans(lane, G, s, pv);
            issue_x(tid, G, s, xv);
        }
        const long lo_l = -xbase, hi_l = A.T_in - xbase;     // offsets of the signal's first sample / one past its last
        const int x_lo = lo_l < -(1 << 30) ? -(1 << 30) : (lo_l > (1 << 30) ? (1 << 30) : (int)lo_l);
        const int x_hi = hi_l < -(1 << 30) ? -(1 << 30) : (hi_l > (1 << 30) ? (1 << 30) : (int)hi_l);
        const bool x_inside = x_lo <= 0 && x_hi >= 4 * NX * THREADS;   // whole window inside the signal
        FZ_STAMP_NW(t1);
        __syncthreads();                                     // previous pass has finished reading LDS
        FZ_STAMP_NW(t2);

        // ---- registers -> LDS: plans into this wave's own region, the x window as a column-major image
        // (NW = 1: the plans overlay the last chunk slot, which this wave writes only after its last plan read)
        f32x4 *pl_base = reinterpret_cast<f32x4 *>(hd + ((nrows_h * SLOTF + 3) & ~3));
        f32x4 *plw = NW == 1 ? reinterpret_cast<f32x4 *>(hd + (A.nslots - 1) * HD_SLOT) : pl_base + wv * (MAXEV * PL4);
        f32x4 *bnd = pl_base + NW * (MAXEV * PL4);                       // [NW][64]: a wave's first chunk IR, lane-linear
        volatile unsigned *flags = reinterpret_cast<volatile unsigned *>(bnd + NW * 64);   // [NW]: pass id of bnd[w]
        const unsigned pass_id = (unsigned)pid + 1u;
        if (!fetched) {
#pragma unroll
            for (int r = 0; r < NPV; ++r)
                if (lane + 64 * r < n_ev * PL4) plw[lane + 64 * r] = pv[r];
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the plans' LDS-DMA has landed (only this wave reads them)
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i4 = tid + j * THREADS;
            f32x4 v = xv[j];
            if (!x_inside) {                                 // uniform: only windows that overlap an end of the signal
                const int e = 4 * i4;                        // x_lo, x_hi are multiples of 4: all four in or out
                if (!(e >= x_lo && e < x_hi)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (i4 < nrows * 8) xs4[(i4 & 7) * XR + (i4 >> 3)] = v;
        }
        __builtin_amdgcn_wave_barrier();                     // other lanes of this wave read the plan words below
        FZ_STAMP(t3);

        // ---- chunk IRs from the table: lanes 0-31 four adjacent taps of the left ear, lanes 32-63 of the right.
        // Two halves of 8 reads are in flight at any time; slot i - 1 = (IR i-1, IR i - IR i-1) is stored as soon
        // as IR i is known: both ears of two taps per 16-byte LDS write.
        {
            const int half = lane >> 5;
            const int m = seg0 + 4 * (lane & 31);
            const int m_c = m < A.L ? m : A.L - 1;           // idle lanes evaluate a valid tap and drop it
            const unsigned m4 = 4u * (unsigned)m_c;
            const f32x4 live = f32x4{m < A.L ? 1.f : 0.f, m + 1 < A.L ? 1.f : 0.f, m + 2 < A.L ? 1.f : 0.f,
                                     m + 3 < A.L ? 1.f : 0.f};                 // taps >= L read as zero
            const f32x4 *pl = plw + half * (BAS_PLANS_WORDS / 4);
            const int tq = 4 * (lane & 31) + (lane < 32 ? 2 : 0);          // first of the two taps this lane stores
            f32x4 *dst = reinterpret_cast<f32x4 *>(hd) + slot_a * (HD_SLOT / 4) + tq;
            f32x2 *dst_h = reinterpret_cast<f32x2 *>(hd) + slot_a * (HO_SLOT / 2) + tq;    // HONLY rows: 8 bytes per tap
            FzHalf ha, hb;
            f32x4 prev = f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 *pl_last = pl + (n_ev > 0 ? n_ev - 1 : 0) * PL4;
            if (n_ev > 0) fz_issue<0>(tab, pl, m4, L4, ha);  // (a wave without slots - few, long chunks - skips it all)
            // a real loop without branches in its body: the load counters then carry across iterations and each
            // half is folded while the next one is in flight (the last iteration re-requests its own first half)
            for (int i = 0; i < n_ev; ++i) {
                const f32x4 *pl_next = pl + PL4 < pl_last ? pl + PL4 : pl_last;
                fz_issue<1>(tab, pl, m4, L4, hb);
                f32x4 h = fz_finish<0>(pl, ha, f32x4{0.f, 0.f, 0.f, 0.f});
                fz_issue<0>(tab, pl_next, m4, L4, ha);
                h = fz_finish<1>(pl, hb, h) * live;
                if constexpr (HONLY) {                       // row i of this wave = (left, right) of its IR i
                    f32x2 ta, tb;
                    fz_pair_ears(h, ta, tb);
                    if (tq < Lseg) {
                        dst_h[0] = ta;
                        dst_h[1] = tb;
                    }
                    dst_h += HO_SLOT / 2;
                } else {
                    if (NW > 1 && i == 0 && wv > 0) {        // (uniform) the previous wave's last slot needs this IR
                        bnd[wv * 64 + lane] = h;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0 && !fz_inject(A)) flags[wv] = pass_id;
                    }
                    // slot slot_a + i - 1 = (IR i-1, IR i - IR i-1), written once IR i is known
                    if (i > 0) {
                        f32x2 h0a, h0b, da, db;
                        fz_pair_ears(prev, h0a, h0b);
                        fz_pair_ears(h - prev, da, db);
                        if (tq < Lseg) {
                            dst[0] = f32x4{h0a.x, h0a.y, da.x, da.y};
                            dst[1] = f32x4{h0b.x, h0b.y, db.x, db.y};
                        }
                        dst += HD_SLOT / 4;
                    }
                    prev = h;
                }
                pl = pl_next;
            }
            if constexpr (!HONLY && NW > 1) {
                if (need_next) {                             // (uniform) IR slot_b comes from the next wave's LDS copy
                    const bool got = fz_wait_handover(flags + wv + 1, pass_id, A);   // it stored that IR first thing
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    f32x4 nx = bnd[(wv + 1) * 64 + lane];
                    if (!got) nx = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
                    f32x2 h0a, h0b, da, db;
                    fz_pair_ears(prev, h0a, h0b);
                    fz_pair_ears(nx - prev, da, db);
                    if (tq < Lseg) {
                        dst[0] = f32x4{h0a.x, h0a.y, da.x, da.y};
                        dst[1] = f32x4{h0b.x, h0b.y, db.x, db.y};
                    }
                }
            }
        }
        FZ_STAMP(t4);
        __syncthreads();
        FZ_STAMP_NW(t5);

        // ---- next (tile, source, segment); its x window and read plans are requested now and arrive under the FIR phase
        long n_tile = tile;
        int n_s = s, n_sg = sg + 1;
        if (n_sg == nseg) {
            n_sg = 0;
            if (++n_s == A.n_src) {
                n_s = 0;
                ++n_tile;
            }
        }
        Geo GN = G;
        if (n_tile != tile || n_sg != sg) GN = make_geo(n_tile, n_sg);
        if constexpr (PREFETCH) {
            if (pid + 1 < n_pass) {
#if defined(__HIP_DEVICE_COMPILE__)                          // (the builtin exists in the device pass only)
                typedef __attribute__((address_space(3))) void *lds_ptr_t;
#pragma unroll
                for (int r = 0; r < NPV; ++r)
                    if (lane + 64 * r < n_ev * PL4)          // (inactive lanes write nothing: the region holds n_ev * 18 pieces)
                        __builtin_amdgcn_global_load_lds(plan_src(lane, r, GN, n_s),
                                                         (lds_ptr_t)(uintptr_t)(unsigned)reinterpret_cast<uintptr_t>(plw + 64 * r), 16, 0, 2);
#endif
                issue_x(tid, GN, n_s, xv);
            }
        }

        // ---- FIR: input rows rho' = 0..halo above/at the lane's output row
        const int row_out = 64 * wv + lane + halo;           // window row holding the lane's outputs
        const int pos = mo0 + 32 * row_out;
        int sl = (int)((float)pos * A.invK);                 // chunk slot of that row (float estimate, corrected)
        int m_in = pos - sl * A.K;                           // offset of the row inside its chunk
        if (m_in < 0) { m_in += A.K; sl -= 1; }
        if (m_in >= A.K) { m_in -= A.K; sl += 1; }
        const f32x4 *xrow = xs4 + row_out;
#if FZ_SLICE_SHIFT
        {   // fair time slicing between the two workgroups of a CU (see bas_render_hd_kernel)
            const unsigned t = (unsigned)(__builtin_amdgcn_s_memrealtime() >> FZ_SLICE_SHIFT);
            if ((t & 1u) ^ prio_flip) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
        }
#elif defined(FZ_ASYM_PRIO)
        if (prio_flip) __builtin_amdgcn_s_setprio(0);        // experiment: the first half of the grid always goes first
        else __builtin_amdgcn_s_setprio(FZ_ASYM_PRIO);
#elif FZ_STAGE_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll 1
        for (int rp = 0; rp <= halo; ++rp) {
            float al[1];
            if (A.s_pow2) {
                al[0] = (float)(m_in & ~(A.S - 1)) * A.invK;
            } else {                                         // any multiple of 32: m_in / S by float estimate, corrected
                int q = (int)((float)m_in * A.invS);
                const int r = m_in - q * A.S;
                if (r < 0) q -= 1;
                if (r >= A.S) q += 1;
                al[0] = (float)(q * A.S) * A.invK;
            }
            // octet i holds taps t0 = 32 rp - 32 + 8 i .. + 7 of the segment and is live for 0 <= t0 < Lseg (a multiple of 8):
            // i in [lo, hi).  (Closed form: eight compare-and-or chains per row step were 60 scalar instructions, which a wave
            // that has its SIMD to itself pays ~6 clocks each for.)
            const int oct_lo = 4 - 4 * rp > 0 ? 4 - 4 * rp : 0;
            int oct_hi = (Lseg + 32 - 32 * rp) >> 3;         // >= 1 for rp <= halo
            oct_hi = oct_hi > 8 ? 8 : oct_hi;
            const unsigned mk = ((1u << oct_hi) - 1u) & ~((1u << oct_lo) - 1u);
#if !FZ_FFA
            float xr[32];
            fz_load_xrow<XR>(xr, xrow);
#endif
#if FZ_FFA
            if constexpr (USE_ASM) {
                ffa_row_step_asm<XR>(accA, accB, accB16, accP, (unsigned)reinterpret_cast<uintptr_t>(xrow),
                                     (unsigned)reinterpret_cast<uintptr_t>(hd + sl * SLOTF + (32 * rp - 32) * 4), al[0], mk);
            } else {
                float xr[32];
                fz_load_xrow<XR>(xr, xrow);
                ffa_row_step_x<HONLY>(fa, fb, fp, xr, hd + sl * SLOTF + (32 * rp - 32) * (HONLY ? 2 : 4), al[0], mk);
            }
#else
            hd_row_step_x<1, HONLY>(acc, xr, hd + sl * SLOTF + (32 * rp - 32) * (HONLY ? 2 : 4), al, mk);
#endif
            xrow -= 1;
            m_in -= 32;
            if (m_in < 0) {
                m_in += A.K;
                sl -= 1;
            }
        }

        FZ_STAMP_NW(t6);
        FZ_ADD(0, t0, t1); FZ_ADD(1, t1, t2); FZ_ADD(2, t2, t3); FZ_ADD(3, t3, t4); FZ_ADD(4, t4, t5); FZ_ADD(5, t5, t6);
        tile = n_tile;
        s = n_s;
        sg = n_sg;
        G = GN;
    }
    flush(open_tile);
    if (A.direct && A.tail_mode) bas_tail<THREADS>(fz_tail(A, y, peak_bits), __uint_as_float(wmax_bits));
#ifdef BAS_STAMPS
    if (lane0 == 0 && blockIdx.x < 2048) {
        unsigned long long *d = bas_fz_stamps + (blockIdx.x * 4 + wv) * 8;
        for (int i = 0; i < 6; ++i) d[i] = st_acc[i];
#ifdef BAS_LIFETIME_ONLY
        d[5] = st_begin;                                     // (absolute: who started when; no phase is stamped in this build)
#endif
        d[6] = __builtin_amdgcn_s_memrealtime() - st_begin;
        d[7] = (unsigned long long)n_pass;
    }
#endif
}

#ifdef BAS_STAMPS
extern "C" int bas_debug_read_fz_stamps(unsigned long long *host, size_t count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bas_fz_stamps), count * sizeof(unsigned long long));
}
#endif

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct FzPlan {
    int nw;                    // waves per workgroup: 4 or 1 (0: shape not served)
    int split;                 // one workgroup of 4 filter + 4 stager waves per CU (bas_fused_split.hip)
    int quad;                  // four waves per tile of 2048, the row steps dealt over them (bas_fused_quad.hip)
    int honly;                 // LDS rows hold (h_L, h_R) only (chunk sizes below ~448)
    int tile, nslots, spw;
    long n_tiles, units_total;
    int n_wg, units_per_wg, parts_per_wg;
    size_t lds_bytes, slab_bytes;
};

static int fz_slots(int nw, int K) {                         // chunk slots a window of this tile can touch (any alignment)
    const int rows = 2048 * nw / 32 + HD_HALO;
    return (K - 32 + 32 * (rows - 1)) / K + 1;
}

// The same for the windows that really occur: tile t, tap segment sg start at t * tile - 128 sg - 32 halo, whose offset
// inside its chunk repeats with period K / gcd(tile, K) - chunk sizes that divide the tile (512, 1024, ..) always start
// 128 samples before a chunk boundary and need one slot less than the worst case.
static int fz_slots_exact(int nw, int K, int Lp, long n_tiles) {
    const long tile = 2048L * nw;
    const int nseg = (Lp + RT_SEG - 1) / RT_SEG;
    int worst = 1;
    const long t_max = n_tiles < K ? n_tiles : K;
    for (long t = 0; t < t_max; ++t) {
        for (int sg = 0; sg < nseg; ++sg) {
            const int Lseg = Lp - sg * RT_SEG < RT_SEG ? Lp - sg * RT_SEG : RT_SEG;
            const int halo = (Lseg + 31) >> 5;
            const long xbase = t * tile - (long)sg * RT_SEG - 32L * halo;
            long mo = xbase % K;
            if (mo < 0) mo += K;
            const int slots = (int)((mo + 32L * (tile / 32 + halo - 1)) / K) + 1;
            if (slots > worst) worst = slots;
        }
    }
    return worst;
}

static FzPlan fz_plan_uncached(int n_src, long T_in, int K, int S, int L);

// The plan of a shape is asked for three times per render (supported?, workspace size, launch) and the streaming
// renderer renders the same shape block after block: remember the last one per thread.
static FzPlan fz_plan(int n_src, long T_in, int K, int S, int L) {
    struct Key {
        int n_src, K, S, L, cus;
        long T_in;
    };
    static thread_local Key last_key = {-1, 0, 0, 0, 0, 0};
    static thread_local FzPlan last_plan = {};
    const int cus = bas_device_cus();
#ifndef BAS_DIAG                                             // (the diagnostic build's BAS_FZ_NW may change between calls)
    if (last_key.n_src == n_src && last_key.T_in == T_in && last_key.K == K && last_key.S == S && last_key.L == L &&
        last_key.cus == cus)
        return last_plan;
#endif
    last_plan = fz_plan_uncached(n_src, T_in, K, S, L);
    last_key = Key{n_src, K, S, L, cus, T_in};
    return last_plan;
}

static FzPlan fz_plan_uncached(int n_src, long T_in, int K, int S, int L) {
    FzPlan p = {};
    // subchunks: multiples of 32 (a row of 32 inputs meets one crossfaded tap set), or 16 / 8 - two / four sets per row, which only the
    // unit blocks of the split-role kernel hold (scenes with more than one (tile of 8192, source) unit per CU; L = 97 .. 104, 121 .. 128,
    // or several whole 128-tap segments: bas_fs_unit_len)
    const bool s16 = (S == 16 || S == 8) && FZ_SPLIT && bas_fs_unit_len((L + 7) & ~7) != 0;    // (8: four sets per row)
    if (n_src <= 0 || T_in <= 0 || K < 32 || K % 32 != 0 || (S % 32 != 0 && !s16) || K % S != 0 || L <= 0) return p;
    const long T_out = T_in + L - 1;
    const int cus = bas_device_cus();
    if (fz_slots(4, K) > FZ_MAXSLOTS) {                      // K < 448 or so: h-only rows, four-wave workgroups, two per CU
        if (s16) return p;
        const long n_tiles = (T_out + 8191) / 8192;
        const int rows_h = fz_slots_exact(4, K, (L + 7) & ~7, n_tiles) + 1;
        const int spw = (rows_h + 3) / 4;
        const int xrows = 8192 / 32 + HD_HALO;
        const size_t lds = (size_t)(8 * (xrows + 1) * 4 + ((rows_h * HO_SLOT + 3) & ~3) + 4 * FZ_HO_MAXEV * 2 * BAS_PLANS_WORDS) * sizeof(float);
        const long units = n_tiles * n_src;
        if (spw > FZ_HO_MAXEV || 2 * lds > 160 * 1024 || units < 2L * cus) return p;   // (few sources: the stored-IR path)
#ifdef BAS_DIAG
        if (getenv("BAS_FZ_NW") && atoi(getenv("BAS_FZ_NW")) != 4) return p;
#endif
        p.nw = 4;
        p.honly = 1;
        p.tile = 8192;
        p.nslots = rows_h - 1;
        p.spw = spw;
        p.n_tiles = n_tiles;
        p.units_total = units;
        const long wg = 2L * cus;
        p.units_per_wg = (int)((units + wg - 1) / wg);
        p.n_wg = (int)((units + p.units_per_wg - 1) / p.units_per_wg);
        p.parts_per_wg = (p.units_per_wg + n_src - 2) / n_src + 1;
        p.lds_bytes = lds;
        p.slab_bytes = (size_t)p.n_wg * p.parts_per_wg * 2 * p.tile * sizeof(float);
        return p;
    }
    // largest tile that still gives every workgroup slot of the chip a unit; scenes with few sources (one source
    // x 10 s is 54 tiles of 8192) take smaller tiles and with them more, narrower workgroups
    const int cand[2] = {4, 1};                              // (a tile of 4096 never wins: same number of busy waves as 2048)
#ifdef BAS_DIAG
    const char *force_nw = getenv("BAS_FZ_NW");              // diagnostic build only: force the tile size
#endif
    for (int ci = 0; ci < 2; ++ci) {
        const int nw = cand[ci];
#ifdef BAS_DIAG
        if (force_nw && atoi(force_nw) != nw) continue;
#endif
        const long n_tiles_nw = (T_out + 2048L * nw - 1) / (2048L * nw);
#ifdef BAS_DIAG
        if (nw == 4 && !force_nw) {
#else
        if (nw == 4) {                                       // short signals (real-time blocks): a tile of 8192 that is mostly past the
#endif
            const long n1 = (T_out + 2047) / 2048;           // end of the output costs as much as a full one - narrow tiles then
            if (4 * T_out * n1 * 2048 < 3 * T_out * n_tiles_nw * 8192) continue;   // useful fraction below 3/4 of the narrow tiles'
            // small scenes: the tiles of 2048 fit in FZ_QUAD_ROUNDS rounds of four-wave workgroups (below): narrow tiles
            if (FZ_QUAD && n1 * n_src <= (long)FZ_QUAD_ROUNDS * 2 * cus) continue;
        }
        const int nslots = fz_slots_exact(nw, K, (L + 7) & ~7, n_tiles_nw);
        const int maxev = nw == 4 ? 6 : 7;
        const int spw = (nslots + nw - 1) / nw;                 // (h-only rows only; the (h0, d) path deals nslots + 1 IRs)
        if ((nslots + 1 + nw - 1) / nw > maxev) continue;
        const int rows = 2048 * nw / 32 + HD_HALO;
        // (one-wave workgroups keep their plans in the LAST slot's space: that slot is written after the last plan read)
        // (+ for nw > 1: one 1 KB boundary chunk IR per wave and the hand-over flags)
        const size_t lds = (size_t)(8 * (rows + 1) * 4 + ((nslots * HD_SLOT + 3) & ~3) +
                                    (nw == 1 ? 0 : nw * maxev * 2 * BAS_PLANS_WORDS + nw * 64 * 4 + 4)) * sizeof(float);
        long wg_per_cu = (long)(160 * 1024 / lds);
        const long by_waves = 8 / nw;                        // two waves per SIMD (register budget of the row step)
        if (wg_per_cu > by_waves) wg_per_cu = by_waves;
#ifdef BAS_DIAG
        if (getenv("BAS_FZ_WG_PER_CU")) wg_per_cu = atoi(getenv("BAS_FZ_WG_PER_CU"));   // diagnostic: a workgroup alone on its CU
#endif
        if (wg_per_cu < 1) continue;
        long slots = wg_per_cu * cus;
        const long n_tiles = (T_out + 2048L * nw - 1) / (2048L * nw);
        const long units = n_tiles * n_src;
        // split roles (one workgroup per CU, the staging of unit u + 1 under the FIR of unit u): worth it as soon as some
        // workgroup has two units (the first unit's staging is exposed); needs two (x image, taps) buffers in LDS
        const bool split_fits = FZ_SPLIT && nw == 4 && wg_per_cu == 2 && bas_fs_lds_bytes(nslots) <= 160 * 1024;
        bool split = split_fits && units > (long)FZ_SPLIT_MIN_UNITS * cus;
#ifdef BAS_DIAG
        if (getenv("BAS_FZ_SPLIT")) split = split_fits && atoi(getenv("BAS_FZ_SPLIT")) != 0;   // (1: also for small scenes)
#endif
        if (split) slots = cus;
        // four waves per tile of 2048 (staging and row steps dealt over them: a unit takes a third of the one-wave kernel's
        // time, a CU holds two such workgroups instead of eight one-wave ones): for scenes whose units fit in a few rounds
        bool quad = false;
        size_t quad_lds = 0;
        if (FZ_QUAD && nw == 1 && (nslots + 1 + 3) / 4 <= BAS_FQ_MAXEV) {
            quad_lds = bas_fq_lds_bytes(nslots);
            const long per_cu = quad_lds * 2 <= 160 * 1024 ? 2 : quad_lds <= 160 * 1024 ? 1 : 0;
            quad = per_cu > 0 && units <= (long)FZ_QUAD_ROUNDS * per_cu * cus;
#ifdef BAS_DIAG
            if (getenv("BAS_FZ_QUAD")) quad = per_cu > 0 && atoi(getenv("BAS_FZ_QUAD")) != 0;
#endif
            if (quad) slots = per_cu * cus;
        }
#ifdef BAS_DIAG
        if (units < slots && nw > 1 && !force_nw) continue;
#else
        if (units < slots && nw > 1) continue;               // not enough work for this tile: try a smaller one
#endif
        if (s16 && !split) return FzPlan{};                  // (subchunks of 16: the split-role kernel or the stored-IR path)
        p.nw = nw;
        p.split = split ? 1 : 0;
        p.quad = quad ? 1 : 0;
        p.tile = 2048 * nw;
        p.nslots = nslots;
        p.spw = spw;
        p.n_tiles = n_tiles;
        p.units_total = units;
        const long wg = units < slots ? units : slots;
        p.units_per_wg = (int)((units + wg - 1) / wg);
        p.n_wg = (int)((units + p.units_per_wg - 1) / p.units_per_wg);
        p.parts_per_wg = (p.units_per_wg + n_src - 2) / n_src + 1;
        p.lds_bytes = split ? bas_fs_lds_bytes(nslots) : quad ? quad_lds : lds;
        p.slab_bytes = (size_t)p.n_wg * p.parts_per_wg * 2 * p.tile * sizeof(float);
        return p;
    }
    return p;
}

extern "C" int bas_render_fused_supported(int n_src, long T_in, int K, int S, int L) {
    return fz_plan(n_src, T_in, K, S, L).nw ? 1 : 0;
}

extern "C" const char *bas_render_fused_kernel_name(int n_src, long T_in, int K, int S, int L) {
    const FzPlan p = fz_plan(n_src, T_in, K, S, L);
    if (!p.nw) return "";
    if (p.split) {
        const int u = bas_fs_unit_len((L + 7) & ~7);
        if (S == 16) return u == 128 ? "bas_render_fs_kernel<128,2>" : "bas_render_fs_kernel<104,2>";
        if (S == 8) return u == 128 ? "bas_render_fs_kernel<128,4>" : "bas_render_fs_kernel<104,4>";
        return u == 128 ? "bas_render_fs_kernel<128>" : u == 104 ? "bas_render_fs_kernel<104>" : "bas_render_fs_kernel<0>";
    }
    if (p.quad) return "bas_render_fq_kernel";
    return p.honly ? "bas_render_fz_kernel<4,1>" : p.nw == 4 ? "bas_render_fz_kernel<4,0>" : "bas_render_fz_kernel<1,0>";
}

#ifdef BAS_DIAG
// diagnostic build only (tests, tools/stress_fused.py): which kernel a shape gets - waves per workgroup | h-only rows << 4 |
// split roles << 5 | four waves per tile of 2048 << 6 (0: not served)
extern "C" int bas_debug_fused_plan(int n_src, long T_in, int K, int S, int L) {
    const FzPlan p = fz_plan(n_src, T_in, K, S, L);
    return p.nw | (p.honly << 4) | (p.split << 5) | (p.quad << 6);
}
#endif

extern "C" size_t bas_render_fused_workspace_bytes(int n_src, long T_in, int K, int S, int L) {
    return BAS_WS_HEAD_BYTES + fz_plan(n_src, T_in, K, S, L).slab_bytes + 16;
}

// phases: 1 = the FIR kernel (slab parts, or y itself for scenes whose tiles are each finished by one workgroup),
// 2 = the slab reduce (+ max|y| + peak rule), 3 = both.  One implementation behind bas_render_mix_fused_f32,
// bas_render_mix_fused_profiled_f32, bas_render_fused_fir_f32, bas_render_fused_reduce_f32 and bas_render_stream_block_f32.
// carry != null (a stream block: no peak, no rule): the carried state is moved - and the running peak taken over the emitted
// samples - by the reduce kernel behind its sums where there is one, by the epilogue kernel where the FIR kernel writes y itself.
static int fused_impl(const char *who, int phases, const float *x, long x_stride, const float *packed, const void *plans,
                      int n_src, long T_in, int K, int S, int L, int U, int ndir, float *y, int accumulate, float *peak,
                      int normalize, void *ws, size_t ws_bytes, bas_stream_t stream, void *ev_begin, void *ev_end,
                      const BasCarry *carry = nullptr) {
    BAS_REQUIRE(y, BAS_E_NULL, "%s: y is null", who);
    BAS_REQUIRE(n_src >= 0 && T_in >= 0 && K > 0 && S > 0 && L > 0 && ndir > 0, BAS_E_SHAPE,
                "%s: need n_src>=0, T_in>=0, K,S,L,ndir>0 (n_src=%d T_in=%ld K=%d S=%d L=%d)", who, n_src, T_in, K, S, L);
    BAS_REQUIRE(U >= BAS_PLAN_MIN_U, BAS_E_SHAPE, "%s: needs an upsampling factor >= %d (U=%d)", who, BAS_PLAN_MIN_U, U);
    BAS_REQUIRE(K % S == 0, BAS_E_SHAPE, "%s: subchunksize does not divide chunksize evenly (K=%d S=%d)", who, K, S);
    BAS_REQUIRE(T_in % K == 0, BAS_E_SHAPE, "%s: T_in (%ld) must be a multiple of K (%d)", who, T_in, K);
    BAS_REQUIRE(T_in / K < (1L << 30), BAS_E_SHAPE, "%s: too many chunks", who);
    hipStream_t st = bas_stream(stream);
    const long T_out = T_in + L - 1;
    unsigned int *peak_bits = reinterpret_cast<unsigned int *>(peak);
    const bool live = n_src > 0 && T_in > 0;
    if (!live) {                                             // nothing to render: y = 0 (or untouched), peak = max|y|
        if (!(phases & 2)) return 0;
        if (peak) {
            hipError_t e = hipMemsetAsync(peak, 0, sizeof(float), st);
            if (e != hipSuccess) return bas_fail((int)e, "%s: hipMemsetAsync: %s", who, hipGetErrorString(e));
        }
        if (!accumulate) {
            hipError_t e = hipMemsetAsync(y, 0, (size_t)2 * T_out * sizeof(float), st);
            if (e != hipSuccess) return bas_fail((int)e, "%s: hipMemsetAsync: %s", who, hipGetErrorString(e));
            return 0;
        }
        return (peak || normalize) ? bas_peak_normalize_f32(y, 2 * T_out, peak, normalize, stream) : 0;
    }
    BAS_REQUIRE(x && packed && plans, BAS_E_NULL, "%s: x, packed or plans is null", who);
    BAS_REQUIRE(x_stride >= T_in, BAS_E_SHAPE, "%s: x_stride < T_in", who);
    BAS_REQUIRE(reinterpret_cast<uintptr_t>(plans) % 16 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                    x_stride % 4 == 0 && reinterpret_cast<uintptr_t>(ws) % 16 == 0,
                BAS_E_ALIGN, "%s: x, plans and ws must be 16-byte aligned, x_stride a multiple of 4", who);
    const FzPlan p = fz_plan(n_src, T_in, K, S, L);
    BAS_REQUIRE(p.nw != 0, BAS_E_SHAPE,
                "%s: sizes not served by the fused kernel (bas_render_fused_supported); use bas_interp2d_f32 + "
                "bas_render_mix_f32", who);
    BAS_REQUIRE(p.units_total < (1L << 31) - 65536, BAS_E_SHAPE,
                "%s: %ld (tile, source) work units exceed 2^31: render in blocks", who, p.units_total);
    BAS_REQUIRE(ws && ws_bytes >= BAS_WS_HEAD_BYTES + p.slab_bytes, BAS_E_WORKSPACE,
                "%s: workspace of %zu bytes needed, %zu given", who, (size_t)BAS_WS_HEAD_BYTES + p.slab_bytes, ws_bytes);
    const size_t table_bytes = (size_t)2 * ndir * U * BAS_PLANE(L) * sizeof(float);
    BAS_REQUIRE(table_bytes < (1ul << 31), BAS_E_SHAPE, "%s: table too large", who);
    // workspace: [control block 64 B | BAS_TAIL_MAX_WG maxima | slabs]
    unsigned *ctl = reinterpret_cast<unsigned *>(ws);
    float *wgpeak = reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + BAS_CTL_WORDS * 4);
    float *slab = reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + BAS_WS_HEAD_BYTES);
    float *peak_dev = peak ? peak : reinterpret_cast<float *>(ctl + 8);       // (the rule needs the peak somewhere)
    const bool want_peak = peak != nullptr || normalize != 0;
    const bool y_quads = reinterpret_cast<uintptr_t>(y) % 16 == 0;            // the tail rescales 16 bytes at a time
    BasTail T = {};
    T.ctl = ctl; T.wgpeak = wgpeak; T.y = y; T.n = 2 * T_out; T.peak = peak_dev;
    T.normalize = normalize && y_quads ? 1 : 0;
    FzArgs A;
    A.x_stride = x_stride; A.n_src = n_src; A.T_in = T_in;
    A.K = K; A.S = S; A.L = L; A.Lp = (L + 7) & ~7; A.n_chunks = (int)(T_in / K);
    A.s_pow2 = (S & (S - 1)) == 0;
    A.invK = 1.0f / (float)K; A.invS = 1.0f / (float)S;
    A.units_total = p.units_total; A.units_per_wg = p.units_per_wg; A.parts_per_wg = p.parts_per_wg;
    A.nslots = p.nslots; A.spw = p.spw;
    A.packed_bytes = (unsigned)table_bytes;
    A.direct = p.units_per_wg % n_src == 0;
    A.accumulate = accumulate;
    A.T_out = T_out;
    A.ctl = ctl;
    A.tail_mode = 0;
#ifdef BAS_DIAG
    { const char *d = getenv("BAS_DEBUG_FLAGS"); A.inject = d ? (atoi(d) & 256) : 0; }
#endif
    bool rule_done = false;                                  // the peak rule has been applied inside a kernel tail
    if (A.direct && want_peak) {
        // Direct output: the FIR kernel ends in bas_tail for max|y| (no cleared peak word: one launch less), but NOT for the
        // rule: a lane owns a row here and stores 16 bytes per 128-byte line and instruction, and as sc1 stores - what a
        // rescale by another workgroup would need - those leave the L2 one by one: the four-wave kernel took 29 us instead
        // of 14 for one source x 10 s (profiles/r04_ab_kernel_tails.txt).  The rule stays a launch of its own here.
        if (p.n_wg <= BAS_TAIL_MAX_WG) {
            A.tail_mode = 1;                                 // max|y| only
        } else if (phases & 1) {                             // more workgroups than maxima slots: they max into a cleared word
            hipError_t e = hipMemsetAsync(peak_dev, 0, sizeof(float), st);
            if (e != hipSuccess) return bas_fail((int)e, "%s: hipMemsetAsync: %s", who, hipGetErrorString(e));
        }
    }
    if (!want_peak) peak_bits = nullptr; else peak_bits = reinterpret_cast<unsigned int *>(peak_dev);
    hipEvent_t eb = reinterpret_cast<hipEvent_t>(ev_begin), ee = reinterpret_cast<hipEvent_t>(ev_end);
    if (phases & 1) {
        if (p.split) {
            hipError_t e = bas_fs_launch(A, x, slab, packed, reinterpret_cast<const unsigned *>(plans), y, peak_bits, p.n_wg,
                                         p.lds_bytes, st, eb, ee);
            if (e != hipSuccess) return bas_fail((int)e, "%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
        } else if (p.quad) {
            hipError_t e = bas_fq_launch(A, x, slab, packed, reinterpret_cast<const unsigned *>(plans), y, peak_bits, p.n_wg,
                                         p.lds_bytes, st, eb, ee);
            if (e != hipSuccess) return bas_fail((int)e, "%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
        } else {
            typedef void (*fz_fn)(FzArgs, const float *, float *, const float *, const unsigned *, float *, unsigned int *);
            const fz_fn fn = p.honly ? bas_render_fz_kernel<4, true> : p.nw == 4 ? bas_render_fz_kernel<4, false> : bas_render_fz_kernel<1, false>;
            hipError_t e = bas_allow_full_lds(reinterpret_cast<const void *>(fn));
            if (e != hipSuccess) return bas_fail((int)e, "%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
            if (eb) (void)hipEventRecord(eb, st);
            hipLaunchKernelGGL(fn, dim3(p.n_wg), dim3(64 * p.nw), p.lds_bytes, st, A, x, slab, packed,
                               reinterpret_cast<const unsigned *>(plans), y, peak_bits);
            if (ee) (void)hipEventRecord(ee, st);
        }
        int rc = bas_check_launch(who);
        if (rc) return rc;
    }
    if (!(phases & 2)) return 0;
    if (!A.direct) {                                         // direct output: y and the peak are complete
        int skipped = 1;
        int rc = bas_launch_slab_reduce(slab, p.tile, n_src, p.units_per_wg, p.parts_per_wg, p.n_wg, T_out, y, accumulate,
                                        peak_bits, want_peak ? &T : nullptr, &skipped, carry, st, who);
        if (rc) return rc;
        rule_done = !skipped && T.normalize != 0;
    } else if (carry) {
        return bas_stream_epilogue_f32(carry->x, carry->x_stride, carry->n_src, carry->halo, carry->B, carry->elev, carry->azim,
                                       carry->ang_stride, carry->nh, carry->nb, carry->last, y, T_out,
                                       reinterpret_cast<float *>(carry->running_peak), stream);
    }
    if (normalize && !rule_done) return bas_scale_by_peak_f32(y, 2 * T_out, peak_dev, stream);
    return 0;
}

extern "C" int bas_render_mix_fused_f32(const float *x, long x_stride, const float *packed, const void *plans,
                                        int n_src, long T_in, int K, int S, int L, int U, int ndir, float *y,
                                        int accumulate, float *peak, int normalize, void *ws, size_t ws_bytes,
                                        bas_stream_t stream) {
    return fused_impl("bas_render_mix_fused_f32", 3, x, x_stride, packed, plans, n_src, T_in, K, S, L, U, ndir, y, accumulate,
                      peak, normalize, ws, ws_bytes, stream, nullptr, nullptr);
}

extern "C" int bas_render_mix_fused_profiled_f32(const float *x, long x_stride, const float *packed, const void *plans,
                                                 int n_src, long T_in, int K, int S, int L, int U, int ndir, float *y,
                                                 int accumulate, float *peak, int normalize, void *ws, size_t ws_bytes,
                                                 bas_stream_t stream, void *ev_begin, void *ev_end) {
    return fused_impl("bas_render_mix_fused_profiled_f32", 3, x, x_stride, packed, plans, n_src, T_in, K, S, L, U, ndir, y,
                      accumulate, peak, normalize, ws, ws_bytes, stream, ev_begin, ev_end);
}

extern "C" int bas_render_fused_fir_f32(const float *x, long x_stride, const float *packed, const void *plans, int n_src,
                                        long T_in, int K, int S, int L, int U, int ndir, float *y, int accumulate,
                                        float *peak, int normalize, void *ws, size_t ws_bytes, bas_stream_t stream) {
    return fused_impl("bas_render_fused_fir_f32", 1, x, x_stride, packed, plans, n_src, T_in, K, S, L, U, ndir, y, accumulate,
                      peak, normalize, ws, ws_bytes, stream, nullptr, nullptr);
}

extern "C" int bas_render_fused_reduce_f32(const float *x, long x_stride, const float *packed, const void *plans, int n_src,
                                           long T_in, int K, int S, int L, int U, int ndir, float *y, int accumulate,
                                           float *peak, int normalize, void *ws, size_t ws_bytes, bas_stream_t stream) {
    return fused_impl("bas_render_fused_reduce_f32", 2, x, x_stride, packed, plans, n_src, T_in, K, S, L, U, ndir, y,
                      accumulate, peak, normalize, ws, ws_bytes, stream, nullptr, nullptr);
}

// One block of a stream (stream.py; bas.h "streaming"): the fused render of the window [halo | block] into y - no peak of
// the window, no rule: a stream's samples are handed out before its peak is known - and the carried state of
// bas_stream_epilogue_f32, which rides in the reduce kernel where the scene has one (a launch less per block: 4.4 of the
// 30 us of a 256 x 512 real-time block).  x is written (its halo), T_in = halo + B.
static int stream_block_impl(const char *who, float *x, long x_stride, const float *packed, const void *plans, int n_src,
                             long T_in, int K, int S, int L, int U, int ndir, float *y, void *ws, size_t ws_bytes, int halo,
                             double *elev, double *azim, long ang_stride, int nh, int nb, double *last, float *running_peak,
                             bas_stream_t stream, void *ev_begin, void *ev_end) {
    const long B = T_in - halo;
    BAS_REQUIRE(n_src > 0 && halo >= 0 && B > 0 && nh >= 0 && nb >= 2, BAS_E_SHAPE,
                "%s: need n_src>0, halo>=0, T_in>halo, nh>=0, nb>=2 (n_src=%d halo=%d T_in=%ld nh=%d nb=%d)", who, n_src, halo,
                T_in, nh, nb);
    BAS_REQUIRE(ang_stride >= nh + nb, BAS_E_SHAPE, "%s: ang_stride shorter than nh + nb", who);
    BAS_REQUIRE(x && elev && azim && last, BAS_E_NULL, "%s: null pointer", who);
    BasCarry C;
    C.x = x; C.x_stride = x_stride; C.n_src = n_src; C.halo = halo; C.B = B;
    C.elev = elev; C.azim = azim; C.ang_stride = ang_stride; C.nh = nh; C.nb = nb; C.last = last;
    C.running_peak = reinterpret_cast<unsigned int *>(running_peak);
    return fused_impl(who, 3, x, x_stride, packed, plans, n_src, T_in, K, S, L, U, ndir, y, 0, nullptr, 0, ws, ws_bytes, stream,
                      ev_begin, ev_end, &C);
}

extern "C" int bas_render_stream_block_f32(float *x, long x_stride, const float *packed, const void *plans, int n_src,
                                           long T_in, int K, int S, int L, int U, int ndir, float *y, void *ws,
                                           size_t ws_bytes, int halo, double *elev, double *azim, long ang_stride, int nh,
                                           int nb, double *last, float *running_peak, bas_stream_t stream) {
    return stream_block_impl("bas_render_stream_block_f32", x, x_stride, packed, plans, n_src, T_in, K, S, L, U, ndir, y, ws,
                             ws_bytes, halo, elev, azim, ang_stride, nh, nb, last, running_peak, stream, nullptr, nullptr);
}

extern "C" int bas_render_stream_block_profiled_f32(float *x, long x_stride, const float *packed, const void *plans,
                                                    int n_src, long T_in, int K, int S, int L, int U, int ndir, float *y,
                                                    void *ws, size_t ws_bytes, int halo, double *elev, double *azim,
                                                    long ang_stride, int nh, int nb, double *last, float *running_peak,
                                                    bas_stream_t stream, void *ev_begin, void *ev_end) {
    return stream_block_impl("bas_render_stream_block_profiled_f32", x, x_stride, packed, plans, n_src, T_in, K, S, L, U, ndir,
                             y, ws, ws_bytes, halo, elev, azim, ang_stride, nh, nb, last, running_peak, stream, ev_begin,
                             ev_end);
}
