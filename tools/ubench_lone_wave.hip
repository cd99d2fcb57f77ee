// Micro-benchmark: the assembly row step of the fused kernels (bas_fir_asm.inc) run by ONE wave per SIMD and by two, to see
// what a wave that has its SIMD to itself pays for the non-VALU instructions of the block (LDS reads, waits, mask branches,
// alignment padding).  FIR_INC = a variant written by tools/gen_fir_asm.py (--nowait, --nobranch, --notaps, --nox, --noalign).
//   hipcc -O3 --offload-arch=gfx950 -DFIR_INC='"/tmp/v_full.inc"' tools/ubench_lone_wave.hip -o ubench_lone_full
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#include FIR_INC
#define XR 261
#define SLOT 516

__global__ __launch_bounds__(256, 2) void k(float *out, unsigned long long *cyc, int iters, unsigned mask, int lds_f4) {
    extern __shared__ f32x4 lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < lds_f4; i += 256) lds[i] = f32x4{1e-3f * (i & 7), 2e-3f, -1e-3f, 5e-4f};
    __syncthreads();
    f32x32 a, b, p;
    f32x2 b16 = f32x2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = b[i] = p[i] = 0.f;
    const float *hd = reinterpret_cast<const float *>(lds) + 8 * XR * 4;
    const unsigned xrow = (unsigned)reinterpret_cast<uintptr_t>(lds + 4 + tid);
    const unsigned tap = (unsigned)reinterpret_cast<uintptr_t>(hd + (1 + (tid >> 4)) * SLOT);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mask == 0) {                                         // one (tile, source) unit of a 128-tap segment: five row steps
        for (int it = 0; it < iters; it += 4) {              // (= four full ones), the masks and addresses of the kernels' loop
            unsigned xr = xrow, tp = tap - 32 * 16;
            float al = 0.25f;
#pragma unroll 1
            for (int rp = 0; rp <= 4; ++rp) {
                const int lo = rp == 0 ? 4 : 0;
                int hi = (128 + 32 - 32 * rp) >> 3;
                hi = hi > 8 ? 8 : hi;
                const unsigned mk = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
                ffa_row_step_asm<XR>(a, b, b16, p, xr, tp, al, mk);
                xr -= 16;
                tp += 32 * 16;
                al += 0.0625f;
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) ffa_row_step_asm<XR>(a, b, b16, p, xrow, tap, 0.25f, mask);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = b16.x + b16.y;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += a[i] + b[i] + p[i];
    out[blockIdx.x * 256 + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    const char *name = argc > 2 ? argv[2] : "";
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
        const int n_wg = 256 * per_cu;
        const size_t lds_bytes = per_cu == 1 ? 150 * 1024 : 75 * 1024;     // forces one / two workgroups per CU
        const int lds_f4 = 8 * XR + 17 * SLOT / 4;
        const unsigned mask = argc > 3 ? (unsigned)strtoul(argv[3], 0, 0) : 0xffu;
        float *out;
        unsigned long long *cyc;
        hipMalloc(&out, n_wg * 256 * sizeof(float));
        hipMalloc(&cyc, n_wg * 4 * sizeof(unsigned long long));
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(n_wg), dim3(256), lds_bytes, 0, out, cyc, iters, mask, lds_f4);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms);
        }
        std::vector<unsigned long long> h(n_wg * 4);
        hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2] / iters;
        // 896 VALU instructions per row step (848 packed FMAs, 32 packed adds, 16 adds)
        printf("%-22s %d wave(s) per SIMD: %8.0f clocks (s_memtime) per row step and wave = %5.2f per VALU instruction and SIMD; kernel %.3f ms (%.2f us per row step)\n",
               name, per_cu, med, med / 896.0 / per_cu, best, best * 1e3 / iters);
        hipFree(out);
        hipFree(cyc);
    }
    return 0;
}
