// Micro-benchmark: the hd FIR row step rebuilt piece by piece, to see which piece costs issue rate.
//   V0: register pattern only (as ubench_fir_pattern)      V1: + taps read from LDS per octet + forming FMA
//   V2: + x row read from LDS per row step                  V3: + run-time octet masks (branches)
// 4-wave workgroups, 71 KB of LDS each -> 2 workgroups per CU = 2 waves per SIMD, like bas_render_hd_kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define XR 261
#define SLOT 516

template <int I>
__device__ __forceinline__ void octet(f32x2 (&acc)[32], const float (&xr)[32], const f32x2 (&g)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int delta = 8 * I + j - 32;
#pragma unroll
        for (int o = 0; o < 32; ++o) {
            const int a = o - delta;
            if (a >= 0 && a < 32) acc[o] = __builtin_elementwise_fma(g[j], f32x2{xr[a], xr[a]}, acc[o]);
        }
    }
}

template <int V, int I>
__device__ __forceinline__ void oct(f32x2 (&acc)[32], const float (&xr)[32], f32x2 (&g)[8], const float *hdrow,
                                    float al, unsigned mask) {
    if (V >= 3 && !(mask & (1u << I))) return;
    if (V >= 1) {
        f32x4 hv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[j] = *reinterpret_cast<const f32x4 *>(hdrow + (8 * I + j) * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            g[j] = __builtin_elementwise_fma(f32x2{hv[j].z, hv[j].w}, f32x2{al, al}, f32x2{hv[j].x, hv[j].y});
    }
    octet<I>(acc, xr, g);
}

__device__ __forceinline__ void load_hv(f32x4 (&hv)[8], const float *hdrow, int i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) hv[j] = *reinterpret_cast<const f32x4 *>(hdrow + (8 * i + j) * 4);
}
template <int I>
__device__ __forceinline__ void oct_pipe(f32x2 (&acc)[32], const float (&xr)[32], f32x4 (&hv)[8], const float *hdrow,
                                         float al, unsigned mask) {
    if (!(mask & (1u << I))) return;
    f32x2 g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        g[j] = __builtin_elementwise_fma(f32x2{hv[j].z, hv[j].w}, f32x2{al, al}, f32x2{hv[j].x, hv[j].y});
    if (I < 7 && (mask & (2u << I))) load_hv(hv, hdrow, I + 1);
    octet<I>(acc, xr, g);
}

// in-place packed FMA, x broadcast from the low / high half of an aligned register pair
template <int HIHALF>
__device__ __forceinline__ void fma_asm(f32x2 &acc, f32x2 g, f32x2 xp) {
    if (HIHALF) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(g), "v"(xp));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(g), "v"(xp));
}
template <int I>
__device__ __forceinline__ void octet_asm(f32x2 (&acc)[32], const f32x2 (&xp)[16], const f32x2 (&g)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int delta = 8 * I + j - 32;
#pragma unroll
        for (int o = 0; o < 32; ++o) {
            const int a = o - delta;
            if (a >= 0 && a < 32) {
                if (a & 1) fma_asm<1>(acc[o], g[j], xp[a >> 1]);
                else fma_asm<0>(acc[o], g[j], xp[a >> 1]);
            }
        }
    }
}
template <int I, int HI>
__device__ __forceinline__ void oct_static_asm(f32x2 (&acc)[32], const f32x2 (&xp)[16], f32x4 (&hv)[8], const float *hdrow, float al) {
    if constexpr (I < HI) {
        f32x2 g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            g[j] = __builtin_elementwise_fma(f32x2{hv[j].z, hv[j].w}, f32x2{al, al}, f32x2{hv[j].x, hv[j].y});
        if constexpr (I + 1 < HI) load_hv(hv, hdrow, I + 1);
        __builtin_amdgcn_sched_barrier(0);
        octet_asm<I>(acc, xp, g);
        __builtin_amdgcn_sched_barrier(0);
        oct_static_asm<I + 1, HI>(acc, xp, hv, hdrow, al);
    }
}
template <int LO, int HI>
__device__ __forceinline__ void row_static_asm(f32x2 (&acc)[32], const f32x4 *xrow, const float *hdrow, float al) {
    f32x2 xp[16];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4 v = xrow[c * XR];
        xp[2 * c] = f32x2{v.x, v.y}; xp[2 * c + 1] = f32x2{v.z, v.w};
    }
    f32x4 hv[8];
    load_hv(hv, hdrow, LO);
    __builtin_amdgcn_sched_barrier(0);
    oct_static_asm<LO, HI>(acc, xp, hv, hdrow, al);
}

// static octet set [LO, HI): straight-line, next octet's taps loaded into the same registers once g is formed
template <int I, int HI>
__device__ __forceinline__ void oct_static(f32x2 (&acc)[32], const float (&xr)[32], f32x4 (&hv)[8], const float *hdrow, float al) {
    if constexpr (I < HI) {
        f32x2 g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            g[j] = __builtin_elementwise_fma(f32x2{hv[j].z, hv[j].w}, f32x2{al, al}, f32x2{hv[j].x, hv[j].y});
        if constexpr (I + 1 < HI) load_hv(hv, hdrow, I + 1);
        __builtin_amdgcn_sched_barrier(0);
        octet<I>(acc, xr, g);
        __builtin_amdgcn_sched_barrier(0);
        oct_static<I + 1, HI>(acc, xr, hv, hdrow, al);
    }
}
template <int LO, int HI>
__device__ __forceinline__ void row_static(f32x2 (&acc)[32], const f32x4 *xrow, const float *hdrow, float al) {
    float xr[32];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4 v = xrow[c * XR];
        xr[4 * c] = v.x; xr[4 * c + 1] = v.y; xr[4 * c + 2] = v.z; xr[4 * c + 3] = v.w;
    }
    f32x4 hv[8];
    load_hv(hv, hdrow, LO);
    __builtin_amdgcn_sched_barrier(0);
    oct_static<LO, HI>(acc, xr, hv, hdrow, al);
}

template <int V>
__global__ __launch_bounds__(256, 2) void k(float *out, const float *in, int iters, unsigned mask, int sl_div, const float *gx) {
    extern __shared__ f32x4 lds4[];
    float *hd = reinterpret_cast<float *>(lds4) + 8 * XR * 4;
    for (int i = threadIdx.x; i < 8 * XR; i += 256) lds4[i] = f32x4{in[i & 1023], 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < 18 * SLOT; i += 256) hd[i] = in[i & 1023];
    __syncthreads();
    f32x2 acc[32];
    float xr[32];
    f32x2 g[8];
#pragma unroll
    for (int i = 0; i < 32; ++i) { acc[i] = f32x2{0.f, 0.f}; xr[i] = in[threadIdx.x + i]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = f32x2{in[i], in[i + 8]};
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int it = 0; it < iters; ++it) {
        if (V >= 4) {                                            // pass structure: 5 row steps between barrier pairs
            const int rp = it % 5;
            mask = rp == 0 ? 0xf0u : rp == 4 ? 0x0fu : 0xffu;
            if (rp == 0) {
                __syncthreads();
                if (V == 6 || V == 8 || V == 10 || V == 12) {                          // a pass's worth of global loads -> LDS stores
#pragma unroll
                    for (int j = 0; j < 9; ++j) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(gx + (((long)blockIdx.x * iters + it) * 2340 + threadIdx.x + j * 256) * 4);
                        const int i4 = threadIdx.x + j * 256;
                        if (i4 < 260 * 8) lds4[(i4 & 7) * XR + (i4 >> 3)] = v;
                    }
                }
                __syncthreads();
                if (V >= 5) {
                    const unsigned t = (unsigned)(__builtin_amdgcn_s_memrealtime() >> 12);
                    if ((t & 1u) ^ (blockIdx.x >= (gridDim.x >> 1) ? 1u : 0u)) __builtin_amdgcn_s_setprio(2);
                    else __builtin_amdgcn_s_setprio(0);
                }
            }
        }
        const int row = 64 * wv + lane + 4 - (it & 3);
        const int sl = (32 * row) / sl_div;                       // per-lane chunk slot, as in the kernel
        const float *hdrow = hd + sl * SLOT + (it & 3) * 128;
        const float al = (float)(row & 15) * 0.0625f;
        if (V >= 2) {
            const f32x4 *xrow = lds4 + row;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 v = xrow[c * XR];
                xr[4 * c] = v.x; xr[4 * c + 1] = v.y; xr[4 * c + 2] = v.z; xr[4 * c + 3] = v.w;
            }
        }
        if (V >= 11) {
            const int rp = it % 5;
            if (rp == 0) row_static_asm<4, 8>(acc, lds4 + row, hdrow, al);
            else if (rp == 4) row_static_asm<0, 4>(acc, lds4 + row, hdrow, al);
            else row_static_asm<0, 8>(acc, lds4 + row, hdrow, al);
            continue;
        }
        if (V >= 9) {
            const int rp = it % 5;
            if (rp == 0) row_static<4, 8>(acc, lds4 + row, hdrow, al);
            else if (rp == 4) row_static<0, 4>(acc, lds4 + row, hdrow, al);
            else row_static<0, 8>(acc, lds4 + row, hdrow, al);
            continue;
        }
        if (V >= 7) {
            f32x4 hv[8];
            load_hv(hv, hdrow, __builtin_ctz(mask));
            oct_pipe<0>(acc, xr, hv, hdrow, al, mask); oct_pipe<1>(acc, xr, hv, hdrow, al, mask);
            oct_pipe<2>(acc, xr, hv, hdrow, al, mask); oct_pipe<3>(acc, xr, hv, hdrow, al, mask);
            oct_pipe<4>(acc, xr, hv, hdrow, al, mask); oct_pipe<5>(acc, xr, hv, hdrow, al, mask);
            oct_pipe<6>(acc, xr, hv, hdrow, al, mask); oct_pipe<7>(acc, xr, hv, hdrow, al, mask);
            continue;
        }
        oct<V, 0>(acc, xr, g, hdrow, al, mask); oct<V, 1>(acc, xr, g, hdrow, al, mask);
        oct<V, 2>(acc, xr, g, hdrow, al, mask); oct<V, 3>(acc, xr, g, hdrow, al, mask);
        oct<V, 4>(acc, xr, g, hdrow, al, mask); oct<V, 5>(acc, xr, g, hdrow, al, mask);
        oct<V, 6>(acc, xr, g, hdrow, al, mask); oct<V, 7>(acc, xr, g, hdrow, al, mask);
        if (V == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(g[i]));
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) r += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int V>
static void run(float *out, const float *in, int blocks = 512) {
    const int iters = 400;
    static float *gx = nullptr;
    if (!gx) { (void)hipMalloc(&gx, (size_t)512 * iters * 2340 * 16 + (1 << 20)); (void)hipMemset(gx, 0x3c, (size_t)512 * iters * 2340 * 16 + (1 << 20)); }
    const size_t lds = 8 * XR * 16 + 18 * SLOT * 4;
    (void)hipFuncSetAttribute((const void *)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), lds, 0, out, in, iters, 0xffu, 512, gx);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), lds, 0, out, in, iters, 0xffu, 512, gx);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double pk = (V >= 4 ? 4096.0 / 5 : 1024.0) * iters;
    double flops = pk * 4 * 64 * 4 * blocks;
    printf("V%d (%d WG/CU): %.3f ms, %.1f TFLOP/s (%.2f cycles per FIR v_pk_fma_f32 per SIMD at 2.0 GHz)  %s\n", V, blocks / 256, ms,
           flops / (ms * 1e-3) / 1e12, (ms * 1e-3 * 2.0e9) / (pk * (blocks / 256)), hipGetErrorString(hipGetLastError()));
}

int main() {
    float *out, *in;
    (void)hipMalloc(&out, 1 << 24);
    (void)hipMalloc(&in, 8192);
    {   // random operands: zero-filled inputs let the chip clock ~15 % higher (DVFS) and overstate the ceiling
        static float hbuf[2048];
        unsigned s = 12345u;
        for (int i = 0; i < 2048; ++i) { s = s * 1664525u + 1013904223u; hbuf[i] = ((s >> 8) * (1.0f / 8388608.0f) - 1.0f) * 0.01f; }
        (void)hipMemcpy(in, hbuf, 8192, hipMemcpyHostToDevice);
    }
    run<0>(out, in); run<1>(out, in); run<2>(out, in); run<3>(out, in); run<4>(out, in); run<5>(out, in); run<6>(out, in); run<9>(out, in); run<11>(out, in); run<12>(out, in);
    run<0>(out, in, 256); run<3>(out, in, 256); run<5>(out, in, 256); run<11>(out, in, 256);
    return 0;
}
