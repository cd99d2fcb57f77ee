// a6 + a7 + a8 in one kernel (gfx950) for SMALL scenes - one source (BASELINE configs 2 and 3), a handful of sources, the
// real-time blocks of a stream: at most a few (tile of 2048, source) units per CU, where what counts is how long ONE unit
// takes.  Same arithmetic as bas_render_fz_kernel<1, false> (bas_fused.hip; interpolate_2d apply_hrtf.py:219-279, crossfade
// :442-443, FIR :445-446, overlap-add and mix :450-453) on the same tiles, but FOUR waves per tile instead of one:
//
//   staging   all four waves fetch the x window (3 float4 per lane instead of 9) and evaluate the tile's 6-7 chunk IRs, two
//             each instead of all of them in a chain (dealt and handed over as in the four-wave kernels);
//   FIR       lane = one of the tile's 64 rows in EVERY wave; wave w runs the row steps rp = w (mod 4) - a 128-tap segment
//             has rp = 0 .. 4 with half of the taps live in steps 0 and 4, so the four waves get one full step each - into
//             its own 98 accumulators (the FIR is linear in the taps: partial sums over disjoint tap ranges add up);
//   flush     when the tile is finished the waves' 32 x 2 outputs per row meet in LDS and wave 0 adds them in the fixed
//             order 0, 1, 2, 3 (deterministic) and writes the tile (y itself for one source, a slab part otherwise).
//
// One wave per SIMD runs this instruction stream at ~5 clocks per vector instruction, so a unit takes about a
// quarter of the one-wave kernel's 19 us plus two barriers.  No MFMA: this is a 1-D FIR (BASELINE.json north_star).
#include "bas_fused.h"

#ifndef FZ_NT_LOADS
#define FZ_NT_LOADS 1
#endif

#if FZ_ASM
__global__ __launch_bounds__(256, 2) void bas_render_fq_kernel(
    FzArgs A, const float *__restrict__ x,                   // [n_src] rows of T_in floats, stride A.x_stride
    float *__restrict__ slab,                                // [n_wg][parts_per_wg][2][tile]
    const float *__restrict__ packed,                        // table in phase-plane layout
    const unsigned *__restrict__ plans,                      // [n_src][n_chunks+1][2 ears][BAS_PLANS_WORDS]
    float *__restrict__ y, unsigned int *__restrict__ peak_bits) {   // direct output only: [2][T_out], max|y| bits
    constexpr int NW = 4;
    constexpr int THREADS = 64 * NW;
    constexpr int TILE = 2048;
    constexpr int ROWS = TILE / 32 + HD_HALO;                // rows of 32 inputs in the x window (68)
    constexpr int XR = ROWS + 1;                             // odd: conflict-free column-major image
    constexpr int NX = (ROWS * 8 + THREADS - 1) / THREADS;   // float4 of x per thread (3)
    constexpr int XFLOATS = 8 * XR * 4;
    constexpr int MAXEV = BAS_FQ_MAXEV;                      // chunk IRs one wave evaluates per unit
    constexpr int PL4 = 2 * BAS_PLANS_WORDS / 4;             // float4 per chunk IR's pair of plans (18)
    constexpr int NPV = (MAXEV * PL4 + 63) / 64;             // 16-byte plan pieces per lane
    static_assert(XR == 69, "bas_fir_asm.inc holds the row step for this x-image stride");
    extern __shared__ f32x4 lds4[];
    // LDS: [8][XR] float4 x image | [nslots][HD_SLOT] taps (h0_L, h0_R, d_L, d_R) | [NW][MAXEV * PL4] plans |
    //      [NW][64] float4 boundary IRs | [NW] flags | [3][32][64] float2 outputs of waves 1-3 at a flush
    f32x4 *xs4 = lds4;
    float *hd = reinterpret_cast<float *>(lds4) + XFLOATS;
    f32x4 *pl_base = reinterpret_cast<f32x4 *>(hd + ((A.nslots * HD_SLOT + 3) & ~3));
    f32x4 *bnd = pl_base + NW * (MAXEV * PL4);
    volatile unsigned *flags = reinterpret_cast<volatile unsigned *>(bnd + NW * 64);
    f32x2 *comb = reinterpret_cast<f32x2 *>(bnd + NW * 64 + 1);

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
    unsigned wmax_bits = 0u;                                 // direct output with a tail: max|y| wave 0 has stored (uniform)
    if (tid0 < NW) flags[tid0] = 0u;                         // hand-over flags of the boundary IRs: no pass has id 0
                                                             // (ordered before their first use by the first pass's barrier)
    f32x32 accA, accB, accP;                                 // half-rate partial sums (bas_fir.h), pinned to v[0:97] by the block
    f32x2 accB16 = f32x2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 32; ++i) accA[i] = accB[i] = accP[i] = 0.f;

    const long first_tile = unit0 / A.n_src;
    float *slab_wg = slab + (long)blockIdx.x * A.parts_per_wg * 2 * TILE;
    const __amdgpu_buffer_rsrc_t tab =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(packed), 0, (int)A.packed_bytes, 0x00020000);
    const unsigned L4 = 4u * (unsigned)A.L;

    // every wave holds a share of the tile's outputs (its row steps' taps): they meet in LDS, wave 0 writes the tile
    auto flush = [&](long t) {
        f32x2 acc[32];
#pragma unroll
        for (int p = 0; p < 16; ++p) {                       // y[2p] = A[p] + B[p-1];  y[2p+1] = P[p] - A[p] - B[p]
            const f32x2 a = f32x2{accA[2 * p], accA[2 * p + 1]}, b0 = f32x2{accB[2 * p], accB[2 * p + 1]};
            const f32x2 b1 = p < 15 ? f32x2{accB[2 * p + 2], accB[2 * p + 3]} : accB16;
            const f32x2 pp = f32x2{accP[2 * p], accP[2 * p + 1]};
            acc[2 * p] = a + b0;
            acc[2 * p + 1] = (pp - a) - b1;
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) accA[i] = accB[i] = accP[i] = 0.f;
        accB16 = f32x2{0.f, 0.f};
        if (wv > 0) {
            f32x2 *c = comb + (wv - 1) * 32 * 64 + lane0;
#pragma unroll
            for (int i = 0; i < 32; ++i) c[i * 64] = acc[i];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const f32x2 *c = comb + w * 32 * 64 + lane0;
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] += c[i * 64];
            }
            if (A.direct) {                                  // uniform
                const long n0 = t * TILE + 32 * lane0;       // this lane's first output
                const float lmax = fz_store_row_direct(acc, y, A.T_out, n0, A.accumulate);
                if (A.tail_mode) {
                    const unsigned b = fz_wave_max_bits(lmax);
                    wmax_bits = b > wmax_bits ? b : wmax_bits;
                } else if (peak_bits) {
                    bas_wave_peak_max(lmax, peak_bits);
                }
            } else {
                float *dst = slab_wg + (t - first_tile) * 2 * TILE + 32 * lane0;
                f32x4 *l4 = reinterpret_cast<f32x4 *>(dst);
                f32x4 *r4 = reinterpret_cast<f32x4 *>(dst + TILE);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    l4[i] = f32x4{acc[4 * i].x, acc[4 * i + 1].x, acc[4 * i + 2].x, acc[4 * i + 3].x};
                    r4[i] = f32x4{acc[4 * i].y, acc[4 * i + 1].y, acc[4 * i + 2].y, acc[4 * i + 3].y};
                }
            }
        }
        __syncthreads();                                     // (the next flush may write the meeting place again)
    };

    // scalar state of the walk over (tile, source, tap segment): advanced by counters, never re-divided
    long tile = first_tile;
    int s = (int)(unit0 - first_tile * A.n_src);
    int sg = 0;
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
    // this wave's share of the nslots + 1 chunk IRs: balanced contiguous ranges [slot_a, slot_b); it writes the slots
    // [slot_a, slot_b): slot i holds (IR i, IR i+1 - IR i), so its LAST slot needs the FIRST IR of the next wave, which that
    // wave leaves in LDS as soon as it has it (bnd / flags)
    const int n_ir = A.nslots + 1, base = n_ir / NW, rem = n_ir - base * NW;
    const int slot_a = wv * base + (wv < rem ? wv : rem);
    const int slot_b = slot_a + base + (wv < rem ? 1 : 0);
    const int n_ev = slot_a < slot_b ? slot_b - slot_a : 0;
    const bool need_next = n_ev > 0 && slot_b <= A.nslots;  // slot slot_b - 1 exists: it needs IR slot_b
    f32x4 *plw = pl_base + wv * (MAXEV * PL4);

    for (long pid = 0; pid < n_pass; ++pid) {
        // the thread index is made opaque once per pass: what derives from it is recomputed per pass instead of being held in
        // (or spilled from) registers across the row steps
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        if (tile != open_tile) {                             // the accumulators belong to another tile: hand them over
            if (open_tile >= 0) flush(open_tile);
            open_tile = tile;
        }
        const int seg0 = G.seg0, Lseg = G.Lseg, halo = G.halo, nrows = G.nrows;
        const long xbase = G.xbase;
        __builtin_amdgcn_s_setprio(3);                       // staging is latency bound: its few instructions go first
        // ---- global -> registers: read plans of this wave's chunk IRs (both ears) and the x window
        f32x4 pv[NPV], xv[NX];
        {
            const f32x4 *pl_src = reinterpret_cast<const f32x4 *>(plans) + (long)s * (A.n_chunks + 1) * PL4;
#pragma unroll
            for (int r = 0; r < NPV; ++r) {
                int p = lane + 64 * r;                       // 16-byte piece of the wave's n_ev * 18
                p = p < n_ev * PL4 ? p : 0;
                const int i = (p * 3641) >> 16;              // p / 18 for p < 1000
                const int c = clampi(G.c0 + slot_a + i, 0, A.n_chunks);
#if FZ_NT_LOADS
                pv[r] = __builtin_nontemporal_load(pl_src + (long)c * PL4 + (p - i * PL4));
#else
                pv[r] = pl_src[(long)c * PL4 + (p - i * PL4)];
#endif
            }
        }
        const long lo_l = -xbase, hi_l = A.T_in - xbase;     // offsets of the signal's first sample / one past its last
        const int x_lo = lo_l < -(1 << 30) ? -(1 << 30) : (lo_l > (1 << 30) ? (1 << 30) : (int)lo_l);
        const int x_hi = hi_l < -(1 << 30) ? -(1 << 30) : (hi_l > (1 << 30) ? (1 << 30) : (int)hi_l);
        const bool x_inside = x_lo <= 0 && x_hi >= 4 * NX * THREADS;   // whole window inside the signal
        {
            const float *xwin = x + (long)s * A.x_stride + xbase;
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                int i = 4 * (tid + j * THREADS);
                i = i < x_lo ? x_lo : i;
                i = i > x_hi - 4 ? x_hi - 4 : i;             // clamped into the row (T_in is a multiple of K >= 32)
#if FZ_NT_LOADS
                xv[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(xwin + i));   // streamed once
#else
                xv[j] = *reinterpret_cast<const f32x4 *>(xwin + i);
#endif
            }
        }
        __syncthreads();                                     // previous pass has finished reading LDS
        // ---- registers -> LDS: plans into this wave's own region, the x window as a column-major image
        const unsigned pass_id = (unsigned)pid + 1u;
#pragma unroll
        for (int r = 0; r < NPV; ++r)
            if (lane + 64 * r < n_ev * PL4) plw[lane + 64 * r] = pv[r];
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
            FzHalf ha, hb;
            f32x4 prev = f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 *pl_last = pl + (n_ev > 0 ? n_ev - 1 : 0) * PL4;
            if (n_ev > 0) fz_issue<0>(tab, pl, m4, L4, ha);  // (a wave without slots - few, long chunks - skips it all)
            for (int i = 0; i < n_ev; ++i) {
                const f32x4 *pl_next = pl + PL4 < pl_last ? pl + PL4 : pl_last;
                fz_issue<1>(tab, pl, m4, L4, hb);
                f32x4 h = fz_finish<0>(pl, ha, f32x4{0.f, 0.f, 0.f, 0.f});
                fz_issue<0>(tab, pl_next, m4, L4, ha);
                h = fz_finish<1>(pl, hb, h) * live;
                if (i == 0 && wv > 0) {                      // (uniform) the previous wave's last slot needs this IR
                    bnd[wv * 64 + lane] = h;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0 && !fz_inject(A)) flags[wv] = pass_id;
                }
                if (i > 0) {                                 // slot slot_a + i - 1, written once IR i is known
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
                pl = pl_next;
            }
            if (need_next) {                                 // (uniform) IR slot_b comes from the next wave's LDS copy
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
        __syncthreads();
        __builtin_amdgcn_s_setprio(0);

        // ---- FIR: the lane's row in every wave; this wave's share of the input rows rho' = 0..halo above/at it
        const int row_out = lane + halo;                     // window row holding the lane's outputs
        const int pos = G.mo0 + 32 * row_out;
        int sl = (int)((float)pos * A.invK);                 // chunk slot of that row (float estimate, corrected)
        int m_in = pos - sl * A.K;                           // offset of the row inside its chunk
        if (m_in < 0) { m_in += A.K; sl -= 1; }
        if (m_in >= A.K) { m_in -= A.K; sl += 1; }
        const f32x4 *xrow = xs4 + row_out;
#pragma unroll 1
        for (int rp = 0; rp <= halo; ++rp) {
            if ((rp & (NW - 1)) == wv) {                     // (uniform)
                float al;
                if (A.s_pow2) {
                    al = (float)(m_in & ~(A.S - 1)) * A.invK;
                } else {                                     // any multiple of 32: m_in / S by float estimate, corrected
                    int q = (int)((float)m_in * A.invS);
                    const int r = m_in - q * A.S;
                    if (r < 0) q -= 1;
                    if (r >= A.S) q += 1;
                    al = (float)(q * A.S) * A.invK;
                }
                // octet i holds taps t0 = 32 rp - 32 + 8 i .. + 7 of the segment, live for 0 <= t0 < Lseg (a multiple of 8)
                const int oct_lo = 4 - 4 * rp > 0 ? 4 - 4 * rp : 0;
                int oct_hi = (Lseg + 32 - 32 * rp) >> 3;     // >= 1 for rp <= halo
                oct_hi = oct_hi > 8 ? 8 : oct_hi;
                const unsigned mk = ((1u << oct_hi) - 1u) & ~((1u << oct_lo) - 1u);
                ffa_row_step_asm<XR>(accA, accB, accB16, accP, (unsigned)reinterpret_cast<uintptr_t>(xrow),
                                     (unsigned)reinterpret_cast<uintptr_t>(hd + sl * HD_SLOT + (32 * rp - 32) * 4), al, mk);
            }
            xrow -= 1;
            m_in -= 32;
            if (m_in < 0) {
                m_in += A.K;
                sl -= 1;
            }
        }

        // ---- next (tile, source, segment)
        int n_sg = sg + 1;
        long n_tile = tile;
        if (n_sg == nseg) {
            n_sg = 0;
            if (++s == A.n_src) {
                s = 0;
                ++n_tile;
            }
        }
        if (n_tile != tile || n_sg != sg) G = make_geo(n_tile, n_sg);
        tile = n_tile;
        sg = n_sg;
    }
    flush(open_tile);
    if (A.direct && A.tail_mode) bas_tail<THREADS>(fz_tail(A, y, peak_bits), __uint_as_float(wmax_bits));
}

#endif  // FZ_ASM

size_t bas_fq_lds_bytes(int nslots) {
    const int rows = 2048 / 32 + HD_HALO;
    return (size_t)(8 * (rows + 1) * 4 + ((nslots * HD_SLOT + 3) & ~3) + 4 * BAS_FQ_MAXEV * 2 * BAS_PLANS_WORDS + 4 * 64 * 4 + 4) * sizeof(float) +
           (size_t)3 * 32 * 64 * sizeof(f32x2);
}

hipError_t bas_fq_launch(const FzArgs &A, const float *x, float *slab, const float *packed, const unsigned *plans, float *y,
                         unsigned int *peak_bits, int n_wg, size_t lds_bytes, hipStream_t st, hipEvent_t eb, hipEvent_t ee) {
#if FZ_ASM
    hipError_t e = bas_allow_full_lds(reinterpret_cast<const void *>(bas_render_fq_kernel));
    if (e != hipSuccess) return e;
    if (eb) (void)hipEventRecord(eb, st);
    hipLaunchKernelGGL(bas_render_fq_kernel, dim3(n_wg), dim3(256), lds_bytes, st, A, x, slab, packed, plans, y, peak_bits);
    if (ee) (void)hipEventRecord(ee, st);
    return hipSuccess;
#else
    return hipErrorNotSupported;                             // (make cppstep: the kernel exists only around the assembly blocks)
#endif
}
