// Micro-benchmark of the UNIT block of the split-role kernel (bas_fir_asm.inc: ffa_unit_asm<261, 128> - the five row steps of
// one (tile, source) unit, 3 605 vector instructions, 296 ds_read_b128, 35 waits) run by the four filter waves of a workgroup,
// one per SIMD, on LDS data - alone, or with four partner waves that issue a stager's worth of vector instructions beside
// them (PARTNER_VALU per unit, in bursts between s_sleep).  FIR_INC = a variant written by tools/gen_fir_asm.py
// (--spread=N, --two-waits, --nowait, --notaps, --nox, --noalign).  Clocks per unit say what the filter stream costs with
// nothing else in the way; the kernel-level A/B (tools/ab_fir.py) decides.
//   hipcc -O3 --offload-arch=gfx950 -DFIR_INC='"/tmp/v.inc"' tools/ubench_unit_block.hip -o ubench_unit_v
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
#ifndef ULEN
#define ULEN 128
#endif

__global__ __launch_bounds__(512, 1) void k(float *out, unsigned long long *cyc, int iters, int partner_valu, int lds_f4) {
    extern __shared__ f32x4 lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < lds_f4; i += 512) lds[i] = f32x4{1e-3f * (i & 7), 2e-3f, -1e-3f, 5e-4f};
    __syncthreads();
    if (tid >= 256) {                                        // partner waves: bursts of independent vector instructions
        if (partner_valu <= 0) return;
        float a0 = tid, a1 = 1.f, a2 = 2.f, a3 = 3.f;
        for (int it = 0; it < iters; ++it) {
            for (int b = 0; b < 8; ++b) {
                for (int j = 0; j < partner_valu / 32; ++j) {
                    a0 = fmaf(a0, 1.0001f, a1); a1 = fmaf(a1, 0.9999f, a2); a2 = fmaf(a2, 1.0002f, a3); a3 = fmaf(a3, 0.9998f, a0);
                }
                __builtin_amdgcn_s_sleep(20);
            }
        }
        if (a0 + a1 + a2 + a3 == 12345.f) out[0] = a0;
        return;
    }
    f32x32 a, b, p;
    f32x2 b16 = f32x2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = b[i] = p[i] = 0.f;
    const float *hd = reinterpret_cast<const float *>(lds) + 8 * XR * 4;
    const unsigned xrow4 = (unsigned)reinterpret_cast<uintptr_t>(lds + tid);        // step r reads 16 (4 - r) bytes above
    unsigned tapv[5];
    float alv[5];
    for (int r = 0; r < 5; ++r) {
        tapv[r] = (unsigned)reinterpret_cast<uintptr_t>(hd + (1 + (tid >> 4)) * SLOT + (32 * r - 32) * 4);
        alv[r] = 0.25f + 0.0625f * r;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) ffa_unit_asm<XR, ULEN>(a, b, b16, p, xrow4, tapv, alv);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = b16.x + b16.y;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += a[i] + b[i] + p[i];
    out[blockIdx.x * 256 + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const char *name = argc > 2 ? argv[2] : "";
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int n_wg = 256;
    const int lds_f4 = 8 * XR + 20 * SLOT / 4;
    for (int partner = 0; partner <= 640; partner += 640) {
        float *out;
        unsigned long long *cyc;
        hipMalloc(&out, n_wg * 256 * sizeof(float));
        hipMalloc(&cyc, n_wg * 4 * sizeof(unsigned long long));
        hipMemset(cyc, 0, n_wg * 4 * sizeof(unsigned long long));
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(n_wg), dim3(512), 150 * 1024, 0, out, cyc, iters, partner, lds_f4);
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
        printf("%-26s partner VALU/unit %4d: %8.0f clocks (s_memtime) per unit and filter wave = %5.2f per vector instruction; kernel %.3f ms (%.2f us per unit)\n",
               name, partner, med, med / 3605.0, best, best * 1e3 / iters);
        hipFree(out);
        hipFree(cyc);
    }
    return 0;
}
