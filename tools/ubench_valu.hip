// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 (and DPP / SGPR-operand forms) on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 ubench_valu.hip -o ubench_valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define ITERS 4096
#define NACC 16

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float a, float b) {
    float s[NACC];
    f32x2 p[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { s[i] = threadIdx.x + i; p[i] = f32x2{(float)i, (float)threadIdx.x}; }
    f32x2 bb = f32x2{b, a};
    long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {          // plain v_fma_f32, VGPR operands
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(a), "v"(b));
        } else if (MODE == 1) {   // v_pk_fma_f32
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(bb), "v"(bb));
        } else if (MODE == 2) {   // v_pk_fma_f32 with op_sel broadcast (as the FIR uses)
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(bb), "v"(bb));
        } else if (MODE == 3) {   // v_fmac_f32 with SGPR operand
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s[i]) : "s"(a), "v"(b));
        } else if (MODE == 4) {   // v_fmac_f32 with DPP row_shr:1 on src0
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(s[i]) : "v"(a), "v"(b));
        } else if (MODE == 5) {   // v_pk_mul_f32
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(bb));
        } else if (MODE == 6) {   // v_fmac_f32 with DPP wave_shr:1
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(s[i]) : "v"(a), "v"(b));
        }
    }
    long long t1 = clock64();
    float r = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += s[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0);
}

template <int MODE>
void run(const char *name, int waves_per_simd, int fma_per_instr) {
    float *out;
    hipMalloc(&out, ((1 << 20) + 16) * sizeof(float));
    int blocks = 256 * waves_per_simd;      // 256 CUs, 4 waves/block -> waves_per_simd per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float cyc; hipMemcpy(&cyc, out + (1 << 20), 4, hipMemcpyDeviceToHost);
    double instr = (double)ITERS * NACC;
    double flops = 2.0 * fma_per_instr * 64 * instr * blocks * 4;
    printf("%-28s waves/SIMD=%d  %.2f clk/instr/wave(wall-clock64)  %.1f TFLOP/s  (%.3f ms)\n", name, waves_per_simd,
           cyc / instr, flops / (ms * 1e-3) / 1e12, ms);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("v_fma_f32", w, 1);
        run<1>("v_pk_fma_f32", w, 2);
        run<2>("v_pk_fma_f32 op_sel", w, 2);
        run<3>("v_fmac_f32 sgpr", w, 1);
        run<4>("v_fmac_f32 dpp row_shr", w, 1);
        run<6>("v_fmac_f32 dpp wave_shr", w, 1);
        run<5>("v_pk_mul_f32", w, 2);
    }
    return 0;
}
