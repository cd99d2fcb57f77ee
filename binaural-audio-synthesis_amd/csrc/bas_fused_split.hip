// a6 + a7 + a8 in one kernel (gfx950), split roles: the same arithmetic as bas_render_fz_kernel<4, false> (bas_fused.hip;
// interpolate_2d of every chunk IR a tile needs apply_hrtf.py:219-279, crossfade :442-443, FIR :445-446, overlap-add and
// mix :450-453), same tiles, same order of accumulation inside a unit (the partial sums in the slabs group the sources
// differently: equal to that kernel's output up to float32 summation order) - but ONE workgroup of eight waves per CU:
//
//   waves 4-7 ("stagers")  fetch the x window and the read plans of unit u + 1, write the x image and evaluate the 18 chunk
//                          IRs into LDS buffer (u + 1) & 1 - latency-bound work that issues ~500 vector instructions;
//   waves 0-3 ("filters")  run the row steps of unit u on buffer u & 1 - 3 600 vector instructions per lane, no memory access but LDS;
//   one s_barrier per unit hands the finished buffer over and the drained one back.
//
// Why: with two identical workgroups per CU (bas_fused.hip) a SIMD holds one wave of each, and the VALU idles whenever both
// are in their staging phases at once - 21-23 % of the time, and neither start offsets, clock slots nor priorities keep the two
// in anti-phase.  Here every SIMD holds one stager and one filter by construction: the staging of the next
// unit always runs under the FIR of the current one, and the filter (no staging state in its registers) can afford a row-step
// block with 126 operand registers (bas_fir_asm.inc: ffa_unit_asm).  The price is LDS: two (x image, taps) buffers = 137 KB +
// plans, one workgroup per CU; and a wave alone on its SIMD pays ~6 clocks for every LDS read and wait.
// No MFMA: this is a 1-D FIR (BASELINE.json north_star).
#include "bas_fused.h"

#ifndef FZ_NT_LOADS
#define FZ_NT_LOADS 1
#endif

#ifndef FS_PSPLIT
#define FS_PSPLIT 1             // unit blocks of subchunks >= 32: the half-rate product P split once more (0: round 4's first form)
#endif
#ifndef FS_STAGER_PRIO
#define FS_STAGER_PRIO 3        // wave priority of the stagers (latency bound: their few instructions go first) ...
#endif
#ifndef FS_FILTER_PRIO
#define FS_FILTER_PRIO 0        // ... and of the filters
#endif

#ifdef BAS_STAMPS
// Diagnostic build only (make stamps): per wave, in 10 ns ticks: [0] work of the role (staging or FIR + flush), [1] waiting
// at the hand-over barrier, [6] lifetime, [7] units.
__device__ unsigned long long bas_fs_stamps[1024 * 8 * 8];
#define FS_NOW() __builtin_amdgcn_s_memrealtime()
#else
#define FS_NOW() 0ull
#endif

#if FZ_ASM
// UNITLEN = 128 / 104: every (unit, segment) pass is one whole segment of that many taps (L = 121 .. 128 and the lengths of
// several whole 128-tap segments, 249 .. 256, .., 505 .. 512; L = 97 .. 104 - the reference's default samples_to_keep is
// 100, apply_hrtf.py:595) and its five row steps run as ONE assembly block; 0: per-step blocks.
// (A template parameter, not a branch: with both assembly statements in one loop the compiler keeps the accumulators
// elsewhere and copies all 98 into and out of the pinned registers around every block.)
// NSUB = 2 / 4: subchunks of 16 / 8 samples (the reference accepts any divisor of the chunk, apply_hrtf.py:401-402): a row
// of 32 inputs meets two / four crossfaded tap sets; unit blocks only (ffa_unit2_asm, ffa_unit4_asm).
// PSPLIT (NSUB = 1, unit blocks only): the half-rate product P of the fast FIR is split once more (ffa_unitp_asm: 116
// accumulator registers, 3.3 % fewer vector instructions per unit; DESIGN.md 4.0, profiles/r04_ubench_fast_fir_level2.txt).
template <int UNITLEN, int NSUB = 1, int PSPLIT = 0>
__global__ __launch_bounds__(512, 1) void bas_render_fs_kernel(
    FzArgs A, const float *__restrict__ x,                   // [n_src] rows of T_in floats, stride A.x_stride
    float *__restrict__ slab,                                // [n_wg][parts_per_wg][2][tile]
    const float *__restrict__ packed,                        // table in phase-plane layout
    const unsigned *__restrict__ plans,                      // [n_src][n_chunks+1][2 ears][BAS_PLANS_WORDS]
    float *__restrict__ y, unsigned int *__restrict__ peak_bits) {   // direct output only: [2][T_out], max|y| bits
    constexpr int NW = 4;                                    // waves per role
    constexpr int THREADS = 64 * NW;                         // threads per role
    constexpr int TILE = 2048 * NW;
    constexpr int ROWS = TILE / 32 + HD_HALO;                // rows of 32 inputs in the x window
    constexpr int XR = ROWS + 1;                             // odd: conflict-free column-major image
    constexpr int NX = (ROWS * 8 + THREADS - 1) / THREADS;   // float4 of x per stager thread (9)
    constexpr int XFLOATS = 8 * XR * 4;
    constexpr int MAXEV = BAS_FS_MAXEV;                      // chunk IRs one stager evaluates per unit
    constexpr int PL4 = 2 * BAS_PLANS_WORDS / 4;             // float4 per chunk IR's pair of plans (18)
    constexpr int NPV = (MAXEV * PL4 + 63) / 64;             // 16-byte plan pieces per lane
    static_assert(XR == 261, "bas_fir_asm.inc holds the row step for this x-image stride");
    extern __shared__ f32x4 lds4[];
    // LDS: [2] x ([8][XR] float4 x image, [nslots][HD_SLOT] taps (h0_L, h0_R, d_L, d_R)) | [NW][MAXEV * PL4] plans |
    //      [NW][64] float4 boundary IRs | [NW] flags
    const int buf4 = (XFLOATS + ((A.nslots * HD_SLOT + 3) & ~3)) / 4;
    f32x4 *pl_base = lds4 + 2 * buf4;
    f32x4 *bnd = pl_base + NW * (MAXEV * PL4);
    volatile unsigned *flags = reinterpret_cast<volatile unsigned *>(bnd + NW * 64);

    const int tid0 = threadIdx.x;
    const int lane0 = tid0 & 63;
    const int wv8 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const bool stager = wv8 >= NW;                           // (wave-uniform)
    const int wv = wv8 & (NW - 1);                           // wave index inside its role

    const long unit0 = (long)blockIdx.x * A.units_per_wg;
    long unit1 = unit0 + A.units_per_wg;
    if (unit1 > A.units_total) unit1 = A.units_total;
    const int nseg = (A.Lp + RT_SEG - 1) / RT_SEG;
    const long n_pass = (unit1 - unit0) * nseg;
    if (!A.direct && peak_bits && blockIdx.x == 0 && tid0 == 0) *peak_bits = 0u;   // the reduce kernel maxes into it later
    if (n_pass <= 0) {
        if (A.direct && A.tail_mode) bas_tail<2 * THREADS>(fz_tail(A, y, peak_bits), 0.f);
        return;
    }
    unsigned wmax_bits = 0u;                                 // direct output with a tail: max|y| this wave has stored (uniform)
    if (tid0 < NW) flags[tid0] = 0u;                         // hand-over flags of the boundary IRs: no pass has id 0
    __syncthreads();

    const long first_tile = unit0 / A.n_src;
    // scalar state of the walk over (tile, source, tap segment): advanced by counters, never re-divided (both roles walk it)
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
    auto advance = [&]() {
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
    };
#ifdef BAS_STAMPS
    unsigned long long st_work = 0, st_wait = 0, st_clk = 0;
    const unsigned long long st_begin = FS_NOW(), st_begin_clk = __builtin_amdgcn_s_memtime();
#endif

    if (!stager) {
        // =========================================================================================================
        // filters: lane = one row of 32 outputs of the tile, the mix over sources in 98 pinned registers
        // =========================================================================================================
        if (FS_FILTER_PRIO) __builtin_amdgcn_s_setprio(FS_FILTER_PRIO);
        static_assert(!PSPLIT || (NSUB == 1 && UNITLEN != 0), "the P split exists for the unit blocks of subchunks >= 32");
        f32x32 accA, accB, accP;
        f32x2 accB16 = f32x2{0.f, 0.f};
        f32x16 accPA, accPB, accPP;                          // PSPLIT: P = (PA, PB[-1 .. 7], PP) at quarter rate
        f32x2 accPB8 = f32x2{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 32; ++i) accA[i] = accB[i] = accP[i] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) accPA[i] = accPB[i] = accPP[i] = 0.f;
        float *slab_wg = slab + (long)blockIdx.x * A.parts_per_wg * 2 * TILE;
        auto flush = [&](long t) {
            f32x2 acc[32];
            if constexpr (PSPLIT) {                          // P[2r] = PA[r] + PB[r-1];  P[2r+1] = PP[r] - PA[r] - PB[r]
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const f32x2 a = f32x2{accPA[2 * r], accPA[2 * r + 1]}, b0 = f32x2{accPB[2 * r], accPB[2 * r + 1]};
                    const f32x2 b1 = r < 7 ? f32x2{accPB[2 * r + 2], accPB[2 * r + 3]} : accPB8;
                    const f32x2 pp = f32x2{accPP[2 * r], accPP[2 * r + 1]};
                    const f32x2 pe = a + b0, po = (pp - a) - b1;
                    accP[4 * r] = pe.x; accP[4 * r + 1] = pe.y;
                    accP[4 * r + 2] = po.x; accP[4 * r + 3] = po.y;
                }
            }
#pragma unroll
            for (int p = 0; p < 16; ++p) {                   // y[2p] = A[p] + B[p-1];  y[2p+1] = P[p] - A[p] - B[p]
                const f32x2 a = f32x2{accA[2 * p], accA[2 * p + 1]}, b0 = f32x2{accB[2 * p], accB[2 * p + 1]};
                const f32x2 b1 = p < 15 ? f32x2{accB[2 * p + 2], accB[2 * p + 3]} : accB16;
                const f32x2 pp = f32x2{accP[2 * p], accP[2 * p + 1]};
                acc[2 * p] = a + b0;
                acc[2 * p + 1] = (pp - a) - b1;
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) accA[i] = accB[i] = accP[i] = 0.f;
            accB16 = f32x2{0.f, 0.f};
            if constexpr (PSPLIT) {
#pragma unroll
                for (int i = 0; i < 16; ++i) accPA[i] = accPB[i] = accPP[i] = 0.f;
                accPB8 = f32x2{0.f, 0.f};
            }
            if (A.direct) {                                  // uniform
                const long n0 = t * TILE + 2048 * wv + 32 * lane0;       // this lane's first output
                const float lmax = fz_store_row_direct(acc, y, A.T_out, n0, A.accumulate);
                if (A.tail_mode) {
                    const unsigned b = fz_wave_max_bits(lmax);
                    wmax_bits = b > wmax_bits ? b : wmax_bits;
                } else if (peak_bits) {
                    bas_wave_peak_max(lmax, peak_bits);
                }
                return;
            }
            float *dst = slab_wg + (t - first_tile) * 2 * TILE + 2048 * wv + 32 * lane0;
            f32x4 *l4 = reinterpret_cast<f32x4 *>(dst);
            f32x4 *r4 = reinterpret_cast<f32x4 *>(dst + TILE);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                l4[i] = f32x4{acc[4 * i].x, acc[4 * i + 1].x, acc[4 * i + 2].x, acc[4 * i + 3].x};
                r4[i] = f32x4{acc[4 * i].y, acc[4 * i + 1].y, acc[4 * i + 2].y, acc[4 * i + 3].y};
            }
        };
        long open_tile = -1;                                 // tile whose partial sums the accumulators hold
        // Unit blocks: the lane's five tap rows, crossfade weights and x row depend on the TILE (where its window starts
        // inside a chunk), not on the source: they are worked out when the tile changes (a uniform branch; ~100 vector
        // instructions, six of them quarter-rate integer multiplies, that round 4's first builds issued in front of every
        // unit - 3 % of a unit's vector instructions) and kept as LDS addresses; per unit they move to the other buffer in
        // place (six adds).  The block's operands live in registers across it anyway.
        long addr_tile = -1;                                 // (tile, segment) the addresses below belong to: tile * nseg + segment
        int addr_buf = 0;                                    // the LDS buffer the addresses below point into
        unsigned tapv[5] = {0u, 0u, 0u, 0u, 0u}, xrow4 = 0u;
        float alv[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, blv[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        const unsigned buf_bytes = 16u * (unsigned)buf4;
        for (long pid = 0; pid < n_pass; ++pid) {
            // the thread index is made opaque once per unit: what derives from it is recomputed per unit instead of being
            // held in (or spilled from) registers across the row steps (per-step blocks; the unit blocks: per tile)
            int tid = tid0;
            asm volatile("" : "+v"(tid));
            const int lane = tid & 63;
#ifdef BAS_STAMPS
            const unsigned long long t0 = FS_NOW();
#endif
            if (tile != open_tile) {                         // the accumulators belong to another tile: hand them over
                if (open_tile >= 0) flush(open_tile);        // (in front of the barrier: under the stagers' work)
                open_tile = tile;
            }
            auto weight = [&](int m) {                       // crossfade weight of the subchunk that holds offset m of its chunk
                if (A.s_pow2) return (float)(m & ~(A.S - 1)) * A.invK;
                int q = (int)((float)m * A.invS);            // any multiple of 32: m / S by float estimate, corrected
                const int r = m - q * A.S;
                if (r < 0) q -= 1;
                if (r >= A.S) q += 1;
                return (float)(q * A.S) * A.invK;
            };
            if constexpr (UNITLEN != 0) {
                const int cur_buf = (int)(pid & 1);
                const long akey = tile * nseg + sg;                          // (IRs of several segments: the window moves with the segment)
                if (akey == addr_tile && cur_buf != addr_buf) {              // (uniform) same rows, the other buffer: in place
                    const unsigned delta = cur_buf ? buf_bytes : 0u - buf_bytes;
#pragma unroll
                    for (int r = 0; r < 5; ++r) tapv[r] += delta;
                    xrow4 += delta;
                    addr_buf = cur_buf;
                }
                if (akey != addr_tile) {                     // (uniform; with one tap segment per unit the window depends on the tile alone)
                    addr_tile = akey;
                    addr_buf = cur_buf;
                    const unsigned bufb = (unsigned)reinterpret_cast<uintptr_t>(lds4 + cur_buf * buf4);
                    const int row_out = 64 * wv + lane + G.halo;             // window row holding the lane's outputs
                    const int pos = G.mo0 + 32 * row_out;
                    int sl = (int)((float)pos * A.invK);                     // chunk slot of that row (float estimate, corrected)
                    int m_in = pos - sl * A.K;                               // offset of the row inside its chunk
                    if (m_in < 0) { m_in += A.K; sl -= 1; }
                    if (m_in >= A.K) { m_in -= A.K; sl += 1; }
                    xrow4 = bufb + 16u * (unsigned)(row_out - 4);            // the x row of step 4 (step r reads 16 (4 - r) bytes above)
                    unsigned tap = bufb + 4u * (unsigned)(XFLOATS + sl * HD_SLOT - 32 * 4);   // tap row of step 0
#pragma unroll
                    for (int r = 0; r < 5; ++r) {
                        alv[r] = weight(m_in);
                        if constexpr (NSUB == 2) blv[r] = weight(m_in + 16);   // inputs 16-31 of the row: the next subchunk
                        tapv[r] = tap;
                        m_in -= 32;
                        const bool wrap = m_in < 0;                          // the next row up lies in the chunk before: one slot down
                        m_in += wrap ? A.K : 0;
                        tap += 32u * 16u - (wrap ? 4u * (unsigned)HD_SLOT : 0u);
                    }
                }
            }
#ifdef BAS_STAMPS
            const unsigned long long t1 = FS_NOW();
#endif
            __syncthreads();                                 // buffer pid & 1 holds this unit; the other one is free again
#ifdef BAS_STAMPS
            const unsigned long long t2 = FS_NOW();
#endif
            const f32x4 *xs4 = lds4 + (pid & 1) * buf4;
            const float *hd = reinterpret_cast<const float *>(xs4) + XFLOATS;
            const int Lseg = G.Lseg, halo = G.halo;
#ifdef BAS_STAMPS
            const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#endif
            if constexpr (UNITLEN != 0) {                    // a whole segment of UNITLEN taps: its five row steps in one block
                if constexpr (PSPLIT)
                    ffa_unitp_asm<XR, UNITLEN>(accA, accB, accPA, accPB, accPB8, accPP, accB16, xrow4, tapv, alv);
                else if constexpr (NSUB == 4)                // (the weights of a row's subchunks: al + u S / K, formed in the block)
                    ffa_unit4_asm<XR, UNITLEN>(accA, accB, accB16, accP, xrow4, tapv, alv, (float)A.S * A.invK);
                else if constexpr (NSUB == 2)
                    ffa_unit2_asm<XR, UNITLEN>(accA, accB, accB16, accP, xrow4, tapv, alv, blv);
                else
                    ffa_unit_asm<XR, UNITLEN>(accA, accB, accB16, accP, xrow4, tapv, alv);
                (void)hd; (void)Lseg; (void)halo;
            } else {
                const int row_out = 64 * wv + lane + halo;   // window row holding the lane's outputs
                const int pos = G.mo0 + 32 * row_out;
                int sl = (int)((float)pos * A.invK);         // chunk slot of that row (float estimate, corrected)
                int m_in = pos - sl * A.K;                   // offset of the row inside its chunk
                if (m_in < 0) { m_in += A.K; sl -= 1; }
                if (m_in >= A.K) { m_in -= A.K; sl += 1; }
                const f32x4 *xrow = xs4 + row_out;
#pragma unroll 1
                for (int rp = 0; rp <= halo; ++rp) {             // input rows rho' = 0..halo above/at the lane's output row
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
                    // octet i holds taps t0 = 32 rp - 32 + 8 i .. + 7 of the segment and is live for 0 <= t0 < Lseg (a multiple of 8):
                    // i in [lo, hi).  (Closed form: eight compare-and-or chains per row step were 60 scalar instructions, which a wave
                    // that has its SIMD to itself pays ~6 clocks each for.)
                    const int oct_lo = 4 - 4 * rp > 0 ? 4 - 4 * rp : 0;
                    int oct_hi = (Lseg + 32 - 32 * rp) >> 3;         // >= 1 for rp <= halo
                    oct_hi = oct_hi > 8 ? 8 : oct_hi;
                    const unsigned mk = ((1u << oct_hi) - 1u) & ~((1u << oct_lo) - 1u);
                    ffa_row_step_asm<XR>(accA, accB, accB16, accP, (unsigned)reinterpret_cast<uintptr_t>(xrow),
                                         (unsigned)reinterpret_cast<uintptr_t>(hd + sl * HD_SLOT + (32 * rp - 32) * 4), al, mk);
                    xrow -= 1;
                    m_in -= 32;
                    if (m_in < 0) {
                        m_in += A.K;
                        sl -= 1;
                    }
                }
            }
#ifdef BAS_STAMPS
            st_clk += __builtin_amdgcn_s_memtime() - c0;
            st_work += (t1 - t0) + (FS_NOW() - t2);
            st_wait += t2 - t1;
#endif
            advance();
        }
        flush(open_tile);
    } else {
        // =========================================================================================================
        // stagers: x window -> column-major LDS image, read plans -> chunk IRs -> (h0, d) rows of the unit's slots
        // =========================================================================================================
        __builtin_amdgcn_s_setprio(FS_STAGER_PRIO);
        const __amdgpu_buffer_rsrc_t tab =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(packed), 0, (int)A.packed_bytes, 0x00020000);
        const unsigned L4 = 4u * (unsigned)A.L;
        // this wave's share of the nslots + 1 chunk IRs: balanced contiguous ranges [slot_a, slot_b) - 5, 5, 4, 4 for the 18
        // IRs under a tile of 8192 at K = 512; it writes the slots [slot_a, slot_b): slot i holds (IR i, IR i+1 - IR i), so its
        // LAST slot needs the FIRST IR of the next wave, which that wave leaves in LDS as soon as it has it (bnd / flags)
        const int n_ir = A.nslots + 1, base = n_ir / NW, rem = n_ir - base * NW;
        const int slot_a = rfl(wv * base + (wv < rem ? wv : rem));     // (wave-uniform, and told so: the buffer loads below
        const int slot_b = rfl(slot_a + base + (wv < rem ? 1 : 0));    //  take scalar bases - a vector one costs a waterfall loop)
        const int n_ev = slot_a < slot_b ? slot_b - slot_a : 0;
        const bool need_next = n_ev > 0 && slot_b <= A.nslots;      // slot slot_b - 1 exists: it needs IR slot_b
        f32x4 *plw = pl_base + wv * (MAXEV * PL4);
        // Every vector instruction a stager issues takes a slot from the filter wave of its SIMD (the unit block runs at
        // 16 800 clocks per unit alone and at 19 900 beside its stager: tools/ubench_unit_block.hip), so nothing that depends
        // on the lane alone is recomputed per unit (a stager has registers to spare: no opaque thread index here), and the
        // windows that lie inside the signal - all but a tile's first and last - are fetched by buffer loads whose only
        // per-lane operand is 16 x lane: bases and strides are scalar.
        const int tid = tid0 - THREADS;
        const int lane = tid & 63;
        const int half = lane >> 5;
        const int tq = 4 * (lane & 31) + (lane < 32 ? 2 : 0);      // first of the two taps this lane stores
        const unsigned v16 = 16u * (unsigned)tid, l16 = 16u * (unsigned)lane;
        const bool all_live = A.L == A.Lp;                   // L a multiple of 8: every tap a lane stores exists
        for (long pid = 0; pid < n_pass; ++pid) {
#ifdef BAS_STAMPS
            const unsigned long long t0 = FS_NOW();
#endif
#ifndef FS_NO_STAGE          // ablation (never shipped): the filters run on whatever the LDS holds
            f32x4 *xs4 = lds4 + (pid & 1) * buf4;
            float *hd = reinterpret_cast<float *>(xs4) + XFLOATS;
            const int seg0 = G.seg0, Lseg = G.Lseg, nrows = G.nrows;
            const long xbase = G.xbase;
            // ---- global -> registers: read plans of this wave's chunk IRs (both ears) and the x window
            f32x4 pv[NPV], xv[NX];
            {
                const f32x4 *pl_src = reinterpret_cast<const f32x4 *>(plans) + (long)s * (A.n_chunks + 1) * PL4;
                const int c_first = rfl(G.c0 + slot_a);
                if (c_first >= 0 && c_first + n_ev - 1 <= A.n_chunks) {      // (uniform) the wave's n_ev x 18 pieces are one run
                    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<f32x4 *>(pl_src + (long)c_first * PL4), 0, n_ev * PL4 * 16, 0x00020000);
#pragma unroll
                    for (int r = 0; r < NPV; ++r)            // (pieces past the run read as zero and are not stored)
                        pv[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, (int)l16, 1024 * r, FZ_NT_LOADS ? 2 : 0));
                } else {
#pragma unroll
                    for (int r = 0; r < NPV; ++r) {
                        int p = lane + 64 * r;                   // 16-byte piece of the wave's n_ev * 18
                        p = p < n_ev * PL4 ? p : 0;
                        const int i = (p * 3641) >> 16;          // p / 18 for p < 1000
                        const int c = clampi(c_first + i, 0, A.n_chunks);
#if FZ_NT_LOADS
                        pv[r] = __builtin_nontemporal_load(pl_src + (long)c * PL4 + (p - i * PL4));
#else
                        pv[r] = pl_src[(long)c * PL4 + (p - i * PL4)];
#endif
                    }
                }
            }
            const long lo_l = -xbase, hi_l = A.T_in - xbase;     // offsets of the signal's first sample / one past its last
            const int x_lo = lo_l < -(1 << 30) ? -(1 << 30) : (lo_l > (1 << 30) ? (1 << 30) : (int)lo_l);
            const int x_hi = hi_l < -(1 << 30) ? -(1 << 30) : (hi_l > (1 << 30) ? (1 << 30) : (int)hi_l);
            const bool x_inside = x_lo <= 0 && x_hi >= 4 * NX * THREADS;   // whole window inside the signal
            {
                const float *xwin = x + (long)s * A.x_stride + xbase;
                if (x_inside) {
                    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xwin), 0,
                                                                                        16 * NX * THREADS, 0x00020000);
#pragma unroll
                    for (int j = 0; j < NX; ++j)             // streamed once
                        xv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)v16, 16 * THREADS * j, FZ_NT_LOADS ? 2 : 0));
                } else {
                    asm volatile("" ::: "memory");           // (keeps the two forms apart: merged, every load pays the clamps)
#pragma unroll
                    for (int j = 0; j < NX; ++j) {
                        int i = 4 * (tid + j * THREADS);
                        i = i < x_lo ? x_lo : i;
                        i = i > x_hi - 4 ? x_hi - 4 : i;     // clamped into the row (T_in is a multiple of K >= 32)
#if FZ_NT_LOADS
                        xv[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(xwin + i));
#else
                        xv[j] = *reinterpret_cast<const f32x4 *>(xwin + i);
#endif
                    }
                }
            }
            // ---- registers -> LDS: plans into this wave's own region, the x window as a column-major image
            const unsigned pass_id = (unsigned)pid + 1u;
#pragma unroll
            for (int r = 0; r < NPV; ++r)
                if (lane + 64 * r < n_ev * PL4) plw[lane + 64 * r] = pv[r];
            {
                f32x4 *xcol = xs4 + (tid & 7) * XR + (tid >> 3);            // quad i4 = tid + 256 j: column i4 & 7, row (tid >> 3) + 32 j
                if (!x_inside) {                             // uniform: only windows that overlap an end of the signal
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int j = 0; j < NX; ++j) {
                        const int e = 4 * (tid + j * THREADS);   // x_lo, x_hi are multiples of 4: all four in or out
                        if (!(e >= x_lo && e < x_hi)) xv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
                for (int j = 0; j < NX; ++j) {
                    // (rows: 256 + halo >= 257, i.e. at least 2056 quads: only the last of the nine can lie outside)
                    if ((j + 1) * THREADS <= (TILE / 32 + 1) * 8 || tid + j * THREADS < nrows * 8) xcol[32 * j] = xv[j];
                }
            }
            __builtin_amdgcn_wave_barrier();                 // other lanes of this wave read the plan words below

            // ---- chunk IRs from the table: lanes 0-31 four adjacent taps of the left ear, lanes 32-63 of the right.
            // Two halves of 8 reads are in flight at any time; slot i - 1 = (IR i-1, IR i - IR i-1) is stored as soon
            // as IR i is known: both ears of two taps per 16-byte LDS write.  The regrouping of the ears is linear, so it is
            // done once per IR and the difference is formed on the regrouped values.
            {
                const int m = seg0 + 4 * (lane & 31);
                const int m_c = m < A.L ? m : A.L - 1;       // idle lanes evaluate a valid tap and drop it
                const unsigned m4 = 4u * (unsigned)m_c;
                const f32x4 live = f32x4{m < A.L ? 1.f : 0.f, m + 1 < A.L ? 1.f : 0.f, m + 2 < A.L ? 1.f : 0.f,
                                         m + 3 < A.L ? 1.f : 0.f};             // taps >= L read as zero
                const f32x4 *pl = plw + half * (BAS_PLANS_WORDS / 4);
                f32x4 *dst = reinterpret_cast<f32x4 *>(hd) + slot_a * (HD_SLOT / 4) + tq;
                FzHalf ha, hb;
                f32x2 pa = f32x2{0.f, 0.f}, pb = pa;         // the previous IR, regrouped: (t_a L, t_a R), (t_b L, t_b R)
                const f32x4 *pl_last = pl + (n_ev > 0 ? n_ev - 1 : 0) * PL4;
                if (n_ev > 0) fz_issue<0>(tab, pl, m4, L4, ha);
                // a real loop without branches in its body: the load counters then carry across iterations and each
                // half is folded while the next one is in flight (the last iteration re-requests its own first half)
                for (int i = 0; i < n_ev; ++i) {
                    const f32x4 *pl_next = pl + PL4 < pl_last ? pl + PL4 : pl_last;
                    fz_issue<1>(tab, pl, m4, L4, hb);
                    f32x4 h = fz_finish<0>(pl, ha, f32x4{0.f, 0.f, 0.f, 0.f});
                    fz_issue<0>(tab, pl_next, m4, L4, ha);
                    h = fz_finish<1>(pl, hb, h);
                    if (!all_live) h *= live;                // (uniform)
                    if (i == 0 && wv > 0) {                  // (uniform) the previous wave's last slot needs this IR
                        bnd[wv * 64 + lane] = h;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0 && !fz_inject(A)) flags[wv] = pass_id;
                    }
                    f32x2 ca, cb;
                    fz_pair_ears(h, ca, cb);
                    if (i > 0) {                             // slot slot_a + i - 1, written once IR i is known
                        const f32x2 da = ca - pa, db = cb - pb;
                        if (tq < Lseg) {
                            dst[0] = f32x4{pa.x, pa.y, da.x, da.y};
                            dst[1] = f32x4{pb.x, pb.y, db.x, db.y};
                        }
                        dst += HD_SLOT / 4;
                    }
                    pa = ca;
                    pb = cb;
                    pl = pl_next;
                }
                if (need_next) {                             // (uniform) IR slot_b comes from the next wave's LDS copy
                    const bool got = fz_wait_handover(flags + wv + 1, pass_id, A);   // it stored that IR first thing
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    f32x4 nx = bnd[(wv + 1) * 64 + lane];
                    if (!got) nx = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
                    f32x2 ca, cb;
                    fz_pair_ears(nx, ca, cb);
                    const f32x2 da = ca - pa, db = cb - pb;
                    if (tq < Lseg) {
                        dst[0] = f32x4{pa.x, pa.y, da.x, da.y};
                        dst[1] = f32x4{pb.x, pb.y, db.x, db.y};
                    }
                }
            }
#endif
#ifdef BAS_STAMPS
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long t1 = FS_NOW();
#endif
            __syncthreads();                                 // this unit's buffer goes to the filters; theirs is free again
#ifdef BAS_STAMPS
            st_work += t1 - t0;
            st_wait += FS_NOW() - t1;
#endif
            advance();
        }
    }
    if (A.direct && A.tail_mode) bas_tail<2 * THREADS>(fz_tail(A, y, peak_bits), __uint_as_float(wmax_bits));
#ifdef BAS_STAMPS
    if (lane0 == 0 && blockIdx.x < 1024) {
        unsigned long long *d = bas_fs_stamps + (blockIdx.x * 8 + wv8) * 8;
        d[0] = st_work;
        d[1] = st_wait;
        d[2] = st_clk;                                       // filters: shader clocks inside the row steps
        d[3] = __builtin_amdgcn_s_memtime() - st_begin_clk;  // shader clocks of the wave's lifetime
        d[6] = FS_NOW() - st_begin;
        d[7] = (unsigned long long)n_pass;
    }
#endif
}

#ifdef BAS_STAMPS
extern "C" int bas_debug_read_fs_stamps(unsigned long long *host, size_t count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bas_fs_stamps), count * sizeof(unsigned long long));
}
#endif

#endif  // FZ_ASM

int bas_fs_unit_len(int Lp) {                               // segment lengths bas_fir_asm.inc holds a unit block for:
    // 104 taps; 128 taps - and IRs of SEVERAL whole 128-tap segments (L = 249 .. 256, 377 .. 384, 505 .. 512 - the default
    // samples_to_keep of the reference's loader, apply_hrtf.py:23 - ...): every (unit, segment) pass is one unit block
    return FS_UNIT_BLOCK && (Lp == 104 || (Lp > 0 && Lp % 128 == 0)) ? (Lp == 104 ? 104 : 128) : 0;
}

size_t bas_fs_lds_bytes(int nslots) {
    const int rows = 8192 / 32 + HD_HALO;
    return (size_t)(2 * (8 * (rows + 1) * 4 + ((nslots * HD_SLOT + 3) & ~3)) + 4 * BAS_FS_MAXEV * 2 * BAS_PLANS_WORDS + 4 * 64 * 4 + 4) *
           sizeof(float);
}

hipError_t bas_fs_launch(const FzArgs &A, const float *x, float *slab, const float *packed, const unsigned *plans, float *y,
                         unsigned int *peak_bits, int n_wg, size_t lds_bytes, hipStream_t st, hipEvent_t eb, hipEvent_t ee) {
#if FZ_ASM
    typedef void (*fs_fn)(FzArgs, const float *, float *, const float *, const unsigned *, float *, unsigned int *);
    const int ul = bas_fs_unit_len(A.Lp);
#if FS_PSPLIT
    fs_fn fn = ul == 128 ? bas_render_fs_kernel<128, 1, 1> : ul == 104 ? bas_render_fs_kernel<104, 1, 1> : bas_render_fs_kernel<0>;
#else
    fs_fn fn = ul == 128 ? bas_render_fs_kernel<128> : ul == 104 ? bas_render_fs_kernel<104> : bas_render_fs_kernel<0>;
#endif
    if (A.S == 16 || A.S == 8) {                             // (the plan gives these to this kernel only with a unit block)
        if (ul == 0) return hipErrorNotSupported;
        fn = A.S == 16 ? (ul == 128 ? bas_render_fs_kernel<128, 2> : bas_render_fs_kernel<104, 2>)
                       : (ul == 128 ? bas_render_fs_kernel<128, 4> : bas_render_fs_kernel<104, 4>);
    }
    hipError_t e = bas_allow_full_lds(reinterpret_cast<const void *>(fn));
    if (e != hipSuccess) return e;
    if (eb) (void)hipEventRecord(eb, st);
    hipLaunchKernelGGL(fn, dim3(n_wg), dim3(512), lds_bytes, st, A, x, slab, packed, plans, y, peak_bits);
    if (ee) (void)hipEventRecord(ee, st);
    return hipSuccess;
#else
    return hipErrorNotSupported;                             // (make cppstep: the kernel exists only around the assembly blocks)
#endif
}
