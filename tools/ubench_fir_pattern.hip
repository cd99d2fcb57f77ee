// Micro-benchmark: the FIR's register pattern without any memory traffic:
// acc[o] (float2) += x[a] (scalar, op_sel broadcast) * g[j] (float2), a = o - delta, as in hd_octet_fma.
// Shows what the v_pk_fma_f32 stream alone sustains at 1 / 2 waves per SIMD with compiler-allocated registers,
// on ZERO and on RANDOM operands (the chip lowers its clock under a dense random-data FMA stream: DVFS), with
// the in-kernel clock read from s_memtime / s_memrealtime stamps around the loop (MI355X_MICROARCH.md, DVFS
// give-back item 6).  Evidence for DESIGN.md section 4.1; output committed as profiles/r02_ubench_fir_pattern.txt.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_fir_pattern.hip -o tools/ubench_fir_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

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

__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters, unsigned long long *stamps) {
    f32x2 acc[32];
    float xr[32];
    f32x2 g[8];
#pragma unroll
    for (int i = 0; i < 32; ++i) { acc[i] = f32x2{0.f, 0.f}; xr[i] = in[threadIdx.x + i]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = f32x2{in[i], in[i + 8]};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        octet<0>(acc, xr, g); octet<1>(acc, xr, g); octet<2>(acc, xr, g); octet<3>(acc, xr, g);
        octet<4>(acc, xr, g); octet<5>(acc, xr, g); octet<6>(acc, xr, g); octet<7>(acc, xr, g);
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(g[i]));      // keep the loop from being folded
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) r += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) {                                              // stamps go to their own buffer only
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 1.0;               // back-to-back launches per configuration
    float *out, *in;
    unsigned long long *stamps;
    (void)hipMalloc(&out, 1 << 24);
    (void)hipMalloc(&in, 4096);
    (void)hipMalloc(&stamps, 2 * 1024 * sizeof(unsigned long long));
    static float hbuf[1024];
    const int iters = 400;
    printf("# v_pk_fma_f32 stream of the FIR row step, no memory traffic; %g s of back-to-back launches per line\n", seconds);
    printf("# operands waves/SIMD ms/launch TFLOP/s in-kernel-clock-GHz(median over workgroups) cycles-per-pk-fma-per-SIMD\n");
    for (int random = 0; random <= 1; ++random) {
        unsigned s = 12345u;
        for (int i = 0; i < 1024; ++i) {
            s = s * 1664525u + 1013904223u;
            hbuf[i] = random ? ((s >> 8) * (1.0f / 8388608.0f) - 1.0f) * 0.01f : 0.0f;
        }
        (void)hipMemcpy(in, hbuf, 4096, hipMemcpyHostToDevice);
        for (int wps = 1; wps <= 2; ++wps) {
            const int blocks = 256 * wps;
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, in, iters, stamps);
            (void)hipDeviceSynchronize();
            float ms1 = 0.f;
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, in, iters, stamps);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            (void)hipEventElapsedTime(&ms1, e0, e1);
            int reps = (int)(seconds * 1e3 / (ms1 > 0.01f ? ms1 : 0.01f));
            if (reps < 3) reps = 3;
            (void)hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, in, iters, stamps);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            ms /= reps;
            std::vector<unsigned long long> st(2 * blocks);
            (void)hipMemcpy(st.data(), stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            std::vector<double> ghz(blocks);
            for (int b = 0; b < blocks; ++b) ghz[b] = st[2 * b + 1] ? (double)st[2 * b] / (double)st[2 * b + 1] * 0.1 : 0.0;
            std::sort(ghz.begin(), ghz.end());
            const double clk = ghz[blocks / 2];
            const double pk = 1024.0 * iters;                            // packed FMAs per lane
            const double flops = pk * 4 * 64 * 4 * blocks;
            printf("%-6s %d %.3f %.1f %.3f %.2f\n", random ? "random" : "zero", wps, ms, flops / (ms * 1e-3) / 1e12, clk,
                   (ms * 1e-3 * clk * 1e9) / (pk * wps));
        }
    }
    return 0;
}
