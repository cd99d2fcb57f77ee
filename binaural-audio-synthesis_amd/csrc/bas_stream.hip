// Carried state of block-wise rendering (gfx950).  The reference holds the whole signal in RAM
// (apply_hrtf.py:405-414); its chunk loop is causal (:431-453), so a stream is rendered as [halo | block] windows and
// everything that crosses a block boundary is moved by ONE launch here instead of a dozen framework-level copies:
//   * running peak: max|y| over the samples this block EMITS (both ears), the quantity of apply_hrtf.py:462 taken over
//     the stream so far (the window's first `halo` outputs were emitted by earlier blocks, its last L-1 are incomplete);
//   * the last `halo` input samples of every source move to the front of the input buffer;
//   * the trajectory angles at the chunk boundaries inside that halo move to the front of the angle buffers, and the
//     angles at the block's end are kept for finish().
#include "bas_internal.h"

__global__ __launch_bounds__(256) void bas_stream_epilogue_kernel(
    float *__restrict__ x, long x_stride, int n_src, int halo, long B,
    double *__restrict__ elev, double *__restrict__ azim, long ang_stride, int nh, int nb,
    double *__restrict__ last,                               // [2][n_src]: (elev, azim) at the block's end
    const float *__restrict__ y, long y_stride, unsigned int *__restrict__ peak_bits) {
    const long tid = blockIdx.x * 256L + threadIdx.x;
    const long nthreads = (long)gridDim.x * 256L;
    // ---- running peak over the emitted range [halo, halo + B) of both ears
    float lmax = 0.f;
    for (long i = tid; i < 2 * B; i += nthreads) {
        const long e = i >= B ? 1 : 0;
        lmax = fmaxf(lmax, fabsf(y[e * y_stride + halo + (i - e * B)]));
    }
    if (peak_bits) bas_block_peak_max(lmax, peak_bits);
    BasCarry C;
    C.x = x; C.x_stride = x_stride; C.n_src = n_src; C.halo = halo; C.B = B;
    C.elev = elev; C.azim = azim; C.ang_stride = ang_stride; C.nh = nh; C.nb = nb; C.last = last; C.running_peak = peak_bits;
    bas_carry_moves(C, tid, nthreads);                       // (the input halo, the halo's angles, the block's last angles)
}

extern "C" int bas_stream_epilogue_f32(float *x, long x_stride, int n_src, int halo, long B, double *elev, double *azim,
                                       long ang_stride, int nh, int nb, double *last, const float *y, long y_stride,
                                       float *running_peak, bas_stream_t stream) {
    BAS_REQUIRE(n_src >= 0 && halo >= 0 && B > 0 && nh >= 0 && nb >= 2, BAS_E_SHAPE,
                "bas_stream_epilogue_f32: need n_src>=0, halo>=0, B>0, nh>=0, nb>=2 (n_src=%d halo=%d B=%ld nh=%d nb=%d)",
                n_src, halo, B, nh, nb);
    BAS_REQUIRE(x_stride >= halo + B && ang_stride >= nh + nb && y_stride >= halo + B, BAS_E_SHAPE,
                "bas_stream_epilogue_f32: strides shorter than the window");
    BAS_REQUIRE(y && (n_src == 0 || (x && elev && azim && last)), BAS_E_NULL, "bas_stream_epilogue_f32: null pointer");
    long work = 2 * B > (long)n_src * halo ? 2 * B : (long)n_src * halo;
    hipLaunchKernelGGL(bas_stream_epilogue_kernel, dim3(bas_grid_for(work, 1024)), dim3(256), 0, bas_stream(stream), x,
                       x_stride, n_src, halo, B, elev, azim, ang_stride, nh, nb, last, y, y_stride,
                       reinterpret_cast<unsigned int *>(running_peak));
    return bas_check_launch("bas_stream_epilogue_f32");
}
