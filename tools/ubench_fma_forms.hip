// Micro-benchmark: which instruction form carries the FIR's multiply-adds fastest?  Same Toeplitz register pattern
// as tools/ubench_fir_pattern.hip (acc[o] += x[o - delta] * g[j], both ears), random operands, no memory traffic:
//   V0  v_pk_fma_f32, x broadcast through op_sel (what the kernels use)
//   V1  two v_fma_f32 per (output, tap): left and right ear separately
//   V2  v_pk_fma_f32 on "transposed" pairs: acc pair = two adjacent OUTPUTS of one ear, g pair = two adjacent taps
//       ... (not a drop-in: needs x pairs) - kept out; see DESIGN.md
// Prints ms, TFLOP/s, in-kernel clock and cycles per packed-FMA-equivalent per SIMD at 1 and 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize tools/ubench_fma_forms.hip -o tools/ubench_fma_forms
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int V, int I>
__device__ __forceinline__ void octet(f32x2 (&acc)[32], const float (&xr)[32], const f32x2 (&g)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int delta = 8 * I + j - 32;
#pragma unroll
        for (int o = 0; o < 32; ++o) {
            const int a = o - delta;
            if (a >= 0 && a < 32) {
                if (V == 0) {
                    acc[o] = __builtin_elementwise_fma(g[j], f32x2{xr[a], xr[a]}, acc[o]);
                } else {
                    float l = acc[o].x, r = acc[o].y;
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(l) : "v"(g[j].x), "v"(xr[a]));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(g[j].y), "v"(xr[a]));
                    acc[o] = f32x2{l, r};
                }
            }
        }
    }
}

template <int V>
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
        octet<V, 0>(acc, xr, g); octet<V, 1>(acc, xr, g); octet<V, 2>(acc, xr, g); octet<V, 3>(acc, xr, g);
        octet<V, 4>(acc, xr, g); octet<V, 5>(acc, xr, g); octet<V, 6>(acc, xr, g); octet<V, 7>(acc, xr, g);
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(g[i]));
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) r += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int V>
static void run(float *out, const float *in, unsigned long long *stamps, double seconds) {
    const int iters = 400;
    for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = 256 * wps;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, in, iters, stamps);
        (void)hipDeviceSynchronize();
        float ms1 = 0.f;
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, in, iters, stamps);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms1, e0, e1);
        int reps = (int)(seconds * 1e3 / (ms1 > 0.01f ? ms1 : 0.01f));
        if (reps < 3) reps = 3;
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, in, iters, stamps);
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
        const double pk = 1024.0 * iters;
        const double flops = pk * 4 * 64 * 4 * blocks;
        printf("V%d %d %.3f %.1f %.3f %.2f\n", V, wps, ms, flops / (ms * 1e-3) / 1e12, clk, (ms * 1e-3 * clk * 1e9) / (pk * wps));
    }
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 0.5;
    float *out, *in;
    unsigned long long *stamps;
    (void)hipMalloc(&out, 1 << 24);
    (void)hipMalloc(&in, 4096);
    (void)hipMalloc(&stamps, 2 * 1024 * sizeof(unsigned long long));
    static float hbuf[1024];
    unsigned s = 12345u;
    for (int i = 0; i < 1024; ++i) { s = s * 1664525u + 1013904223u; hbuf[i] = ((s >> 8) * (1.0f / 8388608.0f) - 1.0f) * 0.01f; }
    (void)hipMemcpy(in, hbuf, 4096, hipMemcpyHostToDevice);
    printf("# form waves/SIMD ms/launch TFLOP/s clock-GHz cycles-per-packed-FMA-equivalent-per-SIMD (random operands)\n");
    run<0>(out, in, stamps, seconds);
    run<1>(out, in, stamps, seconds);
    return 0;
}
