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

#define RT_THREADS 256
#define RT_TILE 2048            // outputs per tile: 64 lanes x 32
#define RT_SEG 128              // taps per LDS pass
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
};

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

__device__ __forceinline__ void fma2(f32x2 &acc, float xv, f32x2 g) {
    acc = __builtin_elementwise_fma(g, f32x2{xv, xv}, acc);
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

    for (long unit = unit0; unit < unit1; ++unit) {
        const long tile = unit / A.n_src;
        const int s = (int)(unit - tile * A.n_src);
        if (tile != cur_tile) {
            flush(cur_tile);
            cur_tile = tile;
        }
        const long n0 = tile * RT_TILE;
        const float *xsrc = A.x + (long)s * A.x_stride;
        const float *Hs = A.H + (long)s * (A.n_chunks + 1) * 2 * A.L;

        for (int seg0 = 0; seg0 < A.Lp; seg0 += RT_SEG) {
            const int Lseg = A.Lp - seg0 < RT_SEG ? A.Lp - seg0 : RT_SEG;   // multiple of 8
            const int Q = Lseg >> 3;                                        // 8-tap groups
            const long top = n0 - seg0;                                     // multiple of 32
            const long rho0 = (top - Lseg) >> 5;                            // floor: first row
            const long xbase = rho0 << 5;
            const int nrows = (int)((top >> 5) + RT_TILE / 32 - rho0);      // <= RT_ROWS

            __syncthreads();            // previous pass has finished reading LDS

            // ---- stage x: column-major image, xs4[c][r] = x[xbase + 32 r + 4 c .. +3]
            for (int i4 = tid; i4 < nrows * 8; i4 += RT_THREADS) {
                const long m = xbase + 4L * i4;
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m >= 0 && m < A.T_in) v = *reinterpret_cast<const f32x4 *>(xsrc + m);
                xs4[(i4 & 7) * RT_XR + (i4 >> 3)] = v;
            }

            // ---- stage the crossfaded IR of every row: thread = (ear, tap)
            if (tid < 2 * Lseg) {
                const int e = tid >= Lseg ? 1 : 0;
                const int kap = tid - e * Lseg;
                const int k = seg0 + kap;
                const bool live = k < A.L;
                const float *Hk = Hs + e * A.L + (live ? k : 0);
                // chunk / subchunk position of the first row (clamped into the signal)
                long m_r = xbase;
                long mc = m_r < 0 ? 0 : (m_r >= A.T_in ? A.T_in - 32 : m_r);
                int c = (int)(mc / A.K);
                int mo = (int)(mc - (long)c * A.K);         // offset inside the chunk
                int so = mo % A.S;                          // offset inside the subchunk
                float h0 = live ? Hk[(long)c * 2 * A.L] : 0.f;
                float h1 = live ? Hk[(long)(c + 1) * 2 * A.L] : 0.f;
                const float invK = 1.0f / (float)A.K;
                float *gdst = gs + kap * 2 + e;
                for (int r = 0; r < nrows; ++r) {
                    const float al = (float)(mo - so) * invK;               // apply_hrtf.py:442
                    gdst[r * RT_RS] = (1.0f - al) * h0 + al * h1;           // :443
                    if (m_r >= 0 && m_r + 32 < A.T_in) {    // advance inside the signal only
                        mo += 32;
                        so += 32;
                        if (so >= A.S) so = 0;
                        if (mo >= A.K) {
                            mo = 0;
                            ++c;
                            h0 = h1;
                            h1 = live ? Hk[(long)(c + 1) * 2 * A.L] : 0.f;
                        }
                    }
                    m_r += 32;
                }
            }
            __syncthreads();

            // ---- FIR: blocks q = 0..Q, inputs [n_t - seg0 - 8q, +8), taps seg0 + 8q-7 .. 8q+7
            const int off0 = (int)(top - xbase) + 8 * u;
            Win wa, wb;
            float x8[8];
            {
                const int r = lane + (off0 >> 5), cp = (off0 & 31) >> 3;
                load_x8(x8, xs4, cp, r);
                win_load(wa, gs + r * RT_RS);
                fir_first(acc, x8, wa);
            }
            // middle blocks: window (old,new) ping-pongs between wa and wb
            auto middle = [&](int q, Win &old, Win &nw) {
                const int off = off0 - 8 * q;
                const int r = lane + (off >> 5), cp = (off & 31) >> 3;
                const float *grow = gs + r * RT_RS + 16 * q;
                load_x8(x8, xs4, cp, r);
                win_load(nw, grow);
                if (cp == 3) win_load(old, grow - 16);      // crossed into the previous row
                fir_mid(acc, x8, old, nw);
            };
            auto last = [&](int q, Win &old) {
                const int off = off0 - 8 * q;
                const int r = lane + (off >> 5), cp = (off & 31) >> 3;
                load_x8(x8, xs4, cp, r);
                if (cp == 3) win_load(old, gs + r * RT_RS + 16 * q - 16);
                fir_last(acc, x8, old);
            };
            int q = 1;
            for (; q + 1 < Q; q += 2) {
                middle(q, wa, wb);
                middle(q + 1, wb, wa);
            }
            if (q < Q) {
                middle(q, wa, wb);
                last(q + 1, wb);
            } else {
                last(q, wa);
            }
        }
    }
    if (unit1 > unit0) flush(cur_tile);
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
        for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
        if ((threadIdx.x & 63) == 0) atomicMax(peak_bits, __float_as_uint(lmax));
    }
}

// ---------------------------------------------------------------------------
// Slab reduction (fixed order => deterministic) + optional fused max|y|
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bas_slab_reduce_kernel(const float *__restrict__ slab, int n_src,
                                                                int units_per_wg, int parts_per_wg,
                                                                int n_wg, long T_out,
                                                                float *__restrict__ y, int accumulate,
                                                                unsigned int *peak_bits) {
    float lmax = 0.f;
    const long n4 = (T_out + 3) / 4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
        const long n = i * 4;
        const long tile = n / RT_TILE;
        const int off = (int)(n - tile * RT_TILE);
        const long ulo = tile * n_src, uhi = ulo + n_src - 1;
        const int wlo = (int)(ulo / units_per_wg);
        int whi = (int)(uhi / units_per_wg);
        if (whi > n_wg - 1) whi = n_wg - 1;
        f32x4 sl = f32x4{0.f, 0.f, 0.f, 0.f}, sr = sl;
        for (int w = wlo; w <= whi; ++w) {
            const long first_tile = ((long)w * units_per_wg) / n_src;
            const float *p = slab + (((long)w * parts_per_wg + (tile - first_tile)) * 2) * RT_TILE + off;
            sl += *reinterpret_cast<const f32x4 *>(p);
            sr += *reinterpret_cast<const f32x4 *>(p + RT_TILE);
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
                y[n + j] = a;
                y[T_out + n + j] = b;
                lmax = fmaxf(lmax, fmaxf(fabsf(a), fabsf(b)));
            }
        }
    }
    if (peak_bits) {
        for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
        if ((threadIdx.x & 63) == 0) atomicMax(peak_bits, __float_as_uint(lmax));
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
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if ((threadIdx.x & 63) == 0) atomicMax(peak_bits, __float_as_uint(lmax));
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
        for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
        if ((threadIdx.x & 63) == 0) atomicMax(peak_bits, __float_as_uint(lmax));
    }
}

static int grid_for(long items, int cap) {
    long g = (items + 255) / 256;
    if (g < 1) g = 1;
    return (int)(g < cap ? g : cap);
}

static int device_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        (void)hipGetLastError();
        cus = 256;                                           // MI355X
    }
    return cus;
}

struct RenderPlan {
    bool fast;
    long n_tiles, units_total;
    int n_wg, units_per_wg, parts_per_wg;
    size_t slab_bytes;
};

static RenderPlan plan_render(int n_src, long T_in, int K, int S, int L, bool aligned) {
    RenderPlan p = {};
    p.fast = aligned && n_src > 0 && (S % 32 == 0);
    if (!p.fast) return p;
    const long T_out = T_in + L - 1;
    p.n_tiles = (T_out + RT_TILE - 1) / RT_TILE;
    p.units_total = p.n_tiles * n_src;
    long slots = 2L * device_cus();
    long wg = p.units_total < slots ? p.units_total : slots;
    p.units_per_wg = (int)((p.units_total + wg - 1) / wg);
    p.n_wg = (int)((p.units_total + p.units_per_wg - 1) / p.units_per_wg);
    p.parts_per_wg = (p.units_per_wg + n_src - 2) / n_src + 1;
    p.slab_bytes = (size_t)p.n_wg * p.parts_per_wg * 2 * RT_TILE * sizeof(float);
    return p;
}

extern "C" size_t bas_render_workspace_bytes(int n_src, long T_in, int K, int S, int L) {
    if (n_src <= 0 || T_in <= 0 || K <= 0 || S <= 0 || L <= 0) return 16;
    RenderPlan p = plan_render(n_src, T_in, K, S, L, true);
    return (p.fast ? p.slab_bytes : 0) + 16;
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
                         (reinterpret_cast<uintptr_t>(ws) % 16 == 0);
    const int live_src = T_in == 0 ? 0 : n_src;
    RenderPlan p = plan_render(live_src, T_in, K, S, L, aligned);
    if (!p.fast) {
        if (ev_begin) (void)hipEventRecord(ev_begin, st);
        hipLaunchKernelGGL(bas_render_generic_kernel, dim3(grid_for(T_out, 8192)), dim3(256), 0, st, x, x_stride,
                           H, live_src, T_in, K, S, L, n_chunks, T_out, y, accumulate, peak_bits);
        if (ev_end) (void)hipEventRecord(ev_end, st);
        return bas_check_launch("bas_render_mix_f32(generic)");
    }
    BAS_REQUIRE(ws && ws_bytes >= p.slab_bytes, BAS_E_WORKSPACE,
                "bas_render_mix_f32: workspace of %zu bytes needed, %zu given", p.slab_bytes, ws_bytes);
    RenderArgs A;
    A.x = x; A.x_stride = x_stride; A.H = H; A.n_src = live_src; A.T_in = T_in;
    A.K = K; A.S = S; A.L = L; A.Lp = (L + 7) & ~7; A.n_chunks = n_chunks;
    A.units_total = p.units_total; A.units_per_wg = p.units_per_wg; A.parts_per_wg = p.parts_per_wg;
    A.slab = reinterpret_cast<float *>(ws);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bas_render_rows32_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, RT_LDS_BYTES);
    if (e != hipSuccess) return bas_fail((int)e, "bas_render_mix_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    if (ev_begin) (void)hipEventRecord(ev_begin, st);
    hipLaunchKernelGGL(bas_render_rows32_kernel, dim3(p.n_wg), dim3(RT_THREADS), RT_LDS_BYTES, st, A);
    if (ev_end) (void)hipEventRecord(ev_end, st);
    int rc = bas_check_launch("bas_render_mix_f32(rows32)");
    if (rc) return rc;
    hipLaunchKernelGGL(bas_slab_reduce_kernel, dim3(grid_for((T_out + 3) / 4, 2048)), dim3(256), 0, st,
                       A.slab, live_src, p.units_per_wg, p.parts_per_wg, p.n_wg, T_out, y, accumulate,
                       peak_bits);
    return bas_check_launch("bas_render_mix_f32(reduce)");
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
    hipLaunchKernelGGL(bas_absmax_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, st, y, n,
                       reinterpret_cast<unsigned int *>(peak));
    int rc = bas_check_launch("bas_peak_normalize_f32(absmax)");
    if (rc || !apply) return rc;
    return bas_scale_by_peak_f32(y, n, peak, stream);
}

extern "C" int bas_scale_by_peak_f32(float *y, long n, const float *peak, bas_stream_t stream) {
    BAS_REQUIRE(peak && (y || n == 0), BAS_E_NULL, "bas_scale_by_peak_f32: null pointer");
    BAS_REQUIRE(n >= 0, BAS_E_SHAPE, "bas_scale_by_peak_f32: n < 0");
    if (n == 0) return 0;
    hipLaunchKernelGGL(bas_scale_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, bas_stream(stream), y, n, peak);
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
    hipLaunchKernelGGL(bas_mix_partials_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, st, parts, n_parts,
                       part_stride, n, y, reinterpret_cast<unsigned int *>(peak));
    return bas_check_launch("bas_mix_partials_f32");
}
