/* bas.h - C ABI of the MI355X (gfx950) binaural render library, libbas_hip.so.
 *
 * The reference (mbjd/binaural-audio-synthesis) is pure Python and has no FFI
 * boundary of its own: its boundary for the moving-source render path is three
 * Python functions in apply_hrtf.py.  This header is the native boundary this
 * build puts BEHIND those functions; every entry point names the reference
 * interface it replaces (file:line into the reference tree).  The ctypes stub a
 * reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *     marked "host"; nothing here allocates, frees or retains caller memory.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL =
 *     the default stream) and returns without synchronising, so calls can be
 *     captured into a hipGraph.
 *   - return value: 0 = ok, negative = BAS_E_* argument/shape error (nothing
 *     was enqueued), positive = hipError_t reported by the launch.
 *     bas_last_error() returns a thread-local description of the last failure.
 *   - re-entrant; no mutable global state besides that thread-local string.
 *   - float data is IEEE binary32, computed in binary32 on the device ("f32");
 *     delay bookkeeping (shift amounts) is computed in binary64.
 */
#ifndef BAS_H
#define BAS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BAS_ABI_VERSION 5

#define BAS_E_NULL      (-1)   /* a required pointer is NULL                    */
#define BAS_E_SHAPE     (-2)   /* inconsistent or unsupported sizes             */
#define BAS_E_ALIGN     (-3)   /* pointer/stride alignment requirement violated */
#define BAS_E_WORKSPACE (-4)   /* workspace too small                           */

#define BAS_WS_CONTROL_BYTES 2048   /* head of every workspace: the library's control block (see bas_render_mix_fused_f32) */

typedef void *bas_stream_t;    /* hipStream_t */

/* Library / ABI version (BAS_ABI_VERSION). */
int bas_version(void);

/* Thread-local text for the last non-zero return on this thread ("" if none). */
const char *bas_last_error(void);

/* ---- a1: table layout ----------------------------------------------------
 * Replaces the in-memory table produced by load_irs_and_delaydiffs
 * (apply_hrtf.py:23-46): irs_left/irs_right truncated to M = samples_to_keep*U
 * columns (:43-44).  Re-lays the row-major table
 *     irs   [2 ears][ndir][M]            (ear 0 = left)
 * as phase planes with guard floats at both ends of every plane
 *     packed[2 ears][ndir][U][L + 4],  packed[e][p][i % U][1 + i / U] = irs[e][p][i],
 *     packed[e][p][ph][0] = sample L-1 of the plane, packed[e][p][ph][L+1 .. L+3] = samples 0..2
 * (circular neighbours), L = M / U, so that the stride-U reads of a fractional
 * shift followed by decimation (apply_hrtf.py:156-165) are contiguous across
 * lanes and neither "one sample earlier" nor "the next three taps" need a wrap test.  bas_table_packed_floats() gives
 * the size of `packed` in floats (0 for invalid shapes); the layout belongs to the library build that packed it
 * (a build with -DBAS_PLANE_DOUBLE=1 keeps every plane's samples twice). */
size_t bas_table_packed_floats(int ndir, int M, int U);
int bas_table_pack_f32(const float *irs, int ndir, int M, int U, float *packed,
                       bas_stream_t stream);

/* ---- a4: delay_signal_float (apply_hrtf.py:127-165) ------------------------
 * y[i][j] = (1-f) x[i][(j*down - floor s_i) mod M] + f x[i][(j*down - ceil s_i) mod M],
 * f = s_i - floor s_i, j = 0..ceil(M/down)-1; circular like np.roll (:156-157);
 * decimation before the blend (:160-163).
 *   x [n][M] f32, shifts [n] f64 (samples), y [n][ceil(M/down)] f32. */
int bas_delay_signal_f32(const float *x, const double *shifts, int n, int M, int down,
                         float *y, bas_stream_t stream);

/* ---- a5: delay_compensated_interpolation_with_delaydiff (apply_hrtf.py:53-106)
 *   packed  table from bas_table_pack_f32
 *   diffs   [2 ears][ndir][ndir] f64   (diffs_left, diffs_right; :40-41)
 *   pq      [n][2] int32   (before, after)
 *   alpha   [n] f64
 *   out     [n][2 ears][return_upsampled ? M : L] f32   (:97-104)
 *   delays  [n][2 ears] f64, in non-upsampled samples (:106); may be NULL. */
int bas_ring_interp_f32(const float *packed, const double *diffs, const int32_t *pq,
                        const double *alpha, int n, int ndir, int L, int U,
                        int return_upsampled, float *out, double *delays,
                        bas_stream_t stream);

/* ---- a3 + elevation bracket on the device --------------------------------------
 * (elev, azim) radians -> the (idx, w) inputs of bas_interp2d_f32, float64 branch of
 * sphere.azim_to_interpolation_params (sphere.py:78-121) plus interpolate_2d's elevation
 * bracket and vertical weight (apply_hrtf.py:199-215, :261-266); azimuth wrapped,
 * elevation clamped like the reference.  Same arithmetic as the host's
 * sphere.interpolation_params_batch (bit-identical on finite inputs; tested).
 *   elev, azim [n] f64 (device); idx [n][4] int32, w [n][3] f64 (device)
 *   ring_elev[10] f64, ring_start[10], ring_count[10] int32: HOST arrays (the ten rings:
 *   deg2rad(-45..90), first direction index, number of azimuths); node_az [187] f32
 *   DEVICE array of the table's node azimuths (sphere.py:318 float32 values), ascending
 *   inside every ring with the ring's first node at azimuth 0, as sphere.py:124-319 lists
 *   them: the search for "the last node <= azim" (sphere.py:103) starts from azim's
 *   position on an evenly spaced ring and walks to the exact node. */
int bas_traj_params_f64(const double *elev, const double *azim, long n, const double *ring_elev,
                        const int32_t *ring_start, const int32_t *ring_count, const float *node_az,
                        int32_t *idx, double *w, bas_stream_t stream);

/* The same with the reference's OTHER numeric branch selectable.  Which one the reference takes
 * depends on the scalar type its trajectory function returns (NumPy >= 2 promotion rules):
 *   BAS_BRANCH_F64      azimuth is an np.float64: node comparisons (sphere.py:103-104) and the weight
 *                       a = (azim - az_b) / (az_a - az_b) (:119) in binary64 against the float32 node
 *                       values (= bas_traj_params_f64);
 *   BAS_BRANCH_PYFLOAT  azimuth is a Python float (what the reference's own presets circle_horizontal,
 *                       circle_askew and spiral return, apply_hrtf.py:585-586, :593): the azimuth is
 *                       wrapped in binary64 (:86), then rounded to binary32; comparisons and the weight
 *                       are binary32 (the weight is returned widened to f64).
 * The elevation bracket and the vertical weight (apply_hrtf.py:199-215, :261-266) are binary64 in
 * both branches, as in the reference. */
#define BAS_BRANCH_F64     0
#define BAS_BRANCH_PYFLOAT 1
int bas_traj_params_branch_f64(const double *elev, const double *azim, long n,
                               const double *ring_elev, const int32_t *ring_start,
                               const int32_t *ring_count, const float *node_az, int32_t *idx,
                               double *w, int branch, bas_stream_t stream);

/* ---- a6: interpolate_2d (apply_hrtf.py:171-281), batched --------------------
 * The angle -> (indices, weights) step (sphere.py:78-121 and the elevation
 * bracket apply_hrtf.py:199-215, :261-266) is host logic; this entry point does
 * all table arithmetic (:219-279).
 *   idx [n][4] int32 = (top_before, top_after, bot_before, bot_after)
 *   w   [n][3] f64   = (top_alpha, bot_alpha, a)
 *   H   [n][2 ears][L] f32
 *   ws / ws_bytes: 16-byte aligned scratch of bas_interp2d_workspace_bytes(n) bytes
 *      (per-(query, ear) read plans handed from the plan kernel to the eval kernel). */
size_t bas_interp2d_workspace_bytes(int n);
int bas_interp2d_f32(const float *packed, const double *diffs, const int32_t *idx,
                     const double *w, int n, int ndir, int L, int U, float *H, void *ws,
                     size_t ws_bytes, bas_stream_t stream);

/* ---- a7/a8: make_signal_move_2d inner loops (apply_hrtf.py:431-453) ---------
 * For every source s, input sample m and tap k:
 *   y[e][m+k] += x[s][m] * ((1-al) H[s][c][e][k] + al H[s][c+1][e][k]),
 *   c = m / K,  al = ((m % K) / S) * S / K            (:442-443, :445-446, :450-453)
 * i.e. per-subchunk crossfaded IR, direct-form FIR (what scipy.signal.convolve
 * resolves to at these sizes), overlap-add, summed over sources.
 *   x  [n_src] rows of T_in floats, row stride x_stride floats; T_in % K == 0
 *      (the caller zero-pads as apply_hrtf.py:405-406 does).  The fast kernels need x 16-byte
 *      aligned and x_stride % 4 == 0 (else the plain kernel runs); when T_in % 4 != 0 they read
 *      every row up to the next multiple of 4 floats, which x_stride then covers - also for
 *      the last row, so allocate n_src * x_stride floats.
 *   H  [n_src][T_in/K + 1][2][L] f32 (bas_interp2d_f32 output, chunk IRs at
 *      t = 0, K, .., T_in; :429, :435)
 *   y  [2][T_in + L - 1] f32; overwritten, or added to when accumulate != 0
 *   peak (may be NULL): device float receiving max|y| of the result (:462),
 *      fused into the final pass
 *   ws / ws_bytes: scratch of at least bas_render_workspace_bytes(...) bytes; its first BAS_WS_CONTROL_BYTES are the library's
 *      control block (see bas_render_mix_fused_f32: zero them once after allocating), which this entry point leaves alone. */
size_t bas_render_workspace_bytes(int n_src, long T_in, int K, int S, int L);

/* Name of the FIR kernel bas_render_mix_f32 launches for these sizes with aligned
 * operands ("bas_render_hd_kernel", "bas_render_rows32_kernel" or
 * "bas_render_generic_kernel"); for profiling tools.  The hd kernel (fast path) serves
 * chunk sizes from about 72 samples up with any subchunk size >= 4: multiples of 32 (and
 * 16 / 8 / 4 with K % 32 == 0) at full speed, other sizes through a multi-part row step
 * (2 to 5 times the arithmetic); rows32 serves smaller chunks with subchunks that are
 * multiples of 32; everything else runs the plain generic kernel. */
const char *bas_render_kernel_name(int n_src, long T_in, int K, int S, int L);

int bas_render_mix_f32(const float *x, long x_stride, const float *H, int n_src, long T_in,
                       int K, int S, int L, float *y, int accumulate, float *peak,
                       void *ws, size_t ws_bytes, bas_stream_t stream);

/* Same call; additionally records the caller's hipEvent_t pair (passed as void*, either
 * may be NULL) on `stream` immediately before and after the FIR kernel, so a
 * benchmark can time the dominant kernel live with HIP events. */
int bas_render_mix_profiled_f32(const float *x, long x_stride, const float *H, int n_src,
                                long T_in, int K, int S, int L, float *y, int accumulate,
                                float *peak, void *ws, size_t ws_bytes, bas_stream_t stream,
                                void *ev_begin, void *ev_end);

/* ---- a6 + a7 fused: chunk IRs evaluated inside the FIR kernel ------------------
 * bas_interp2d_plan_f32 runs only the per-(query, ear) plan step of bas_interp2d_f32
 * (delays, shift splits, the 16 folded blend weights; apply_hrtf.py:219-279) and
 * leaves n*2 read plans of 144 bytes in `plans` (bas_interp2d_workspace_bytes(n) bytes,
 * 16-byte aligned; query order [n_src][T_in/K + 1]; needs U >= 4).
 * bas_render_mix_fused_f32 is bas_render_mix_f32 with H replaced by (packed table, plans):
 * the workgroups evaluate the chunk IRs they need while staging (plans staged in LDS,
 * table samples by buffer loads), so the [n][2][L] IR array never exists in HBM.
 * Served for chunk sizes K >= 448 or so (K % 32 == 0) with subchunks that are multiples
 * of 32, for K >= 256 when the scene has at least two workgroups' worth of
 * (8192-output tile, source) units per CU, and for subchunks of 16 and 8 samples (the reference accepts any
 * divisor of the chunk, apply_hrtf.py:401-402) in scenes with more than one such unit per CU and
 * L = 97 .. 104 or 121 .. 128: check bas_render_fused_supported (1 = yes) and
 * otherwise use bas_interp2d_f32 + bas_render_mix_f32.  Scenes with few sources get
 * smaller tiles (2048 outputs) and more workgroups.  ndir = directions in the table (187).
 * ws / ws_bytes: scratch of bas_render_fused_workspace_bytes(...) bytes, 16-byte aligned.  Its first BAS_WS_CONTROL_BYTES (2048) are the
 *    library's control block (arrival counters of the kernel tails, the device-side error record): zero them ONCE after
 *    allocating the workspace (hipMemset); every call leaves the counters zero, so the workspace can be reused call after
 *    call and inside hipGraph replays.  One workspace serves one stream at a time.
 * x must be 16-byte aligned with x_stride % 4 == 0 (BAS_E_ALIGN otherwise).
 * normalize != 0: the peak rule of make_signal_move_2d (apply_hrtf.py:462-464: m = max|y|; if m > 1: y /= m) is applied
 *    to y before the call's work on `stream` ends - inside the tail of the last kernel (no launch of its own; the workgroups
 *    that finish last share the rescale) whenever y is 16-byte aligned, else by bas_scale_by_peak_f32.  `peak` (may be NULL)
 *    receives max|y| BEFORE the rule either way.
 * n == 0 (no queries / no sources) is served by every entry point: the per-query arrays may then be NULL. */
int bas_interp2d_plan_f32(const double *diffs, const int32_t *idx, const double *w, int n,
                          int ndir, int L, int U, void *plans, size_t plans_bytes,
                          bas_stream_t stream);
/* bas_traj_params_branch_f64 + bas_interp2d_plan_f32 in ONE launch (a3 and the plan step of a6: sphere.py:78-121,
 * apply_hrtf.py:199-279): the first two waves of a block do the angle arithmetic of its 128 queries, the (query, ear)
 * threads of all four take the parameters from LDS - no (idx, w) round trip through HBM, one launch less (what every
 * render from angles runs since round 4; one source x 10 s is 863 queries, the headline scene 221 k).
 * Same plans, bit for bit, as the two calls.  elev / azim [n] f64 device; ring_* host, node_az device, branch as in
 * bas_traj_params_branch_f64. */
int bas_interp2d_plan_angles_f32(const double *diffs, const double *elev, const double *azim, int n,
                                 const double *ring_elev, const int32_t *ring_start,
                                 const int32_t *ring_count, const float *node_az, int branch, int ndir,
                                 int L, int U, void *plans, size_t plans_bytes, bas_stream_t stream);
int bas_render_fused_supported(int n_src, long T_in, int K, int S, int L);
/* Name of the kernel bas_render_mix_fused_f32 launches for these operands, for profiling tools ("" when the sizes are
 * not served): "bas_render_fs_kernel<128>" / "<104>" / "<0>" - one workgroup of four filter and four stager waves per
 * CU, two LDS buffers (scenes with more than one (tile of 8192, source) unit per CU; <128>: L = 121 .. 128 and the lengths of
 * several whole 128-tap segments - 249 .. 256, 377 .. 384, 505 .. 512, .. -, <104>: L = 97 .. 104: the five row steps of
 * a (unit, segment) pass as one assembly block; "<128,2>" / "<104,2>", "<128,4>" / "<104,4>": the same
 * for subchunks of 16 / 8 samples - two / four crossfaded tap sets per row of 32 inputs); "bas_render_fq_kernel" - four waves per tile of 2048
 * outputs, staging and row steps dealt over them (small scenes: one source, a handful, real-time blocks) - or
 * "bas_render_fz_kernel<4,0>" / "<1,0>" / "<4,1>": every wave stages and filters (two workgroups of four waves per CU on
 * tiles of 8192 outputs / eight one-wave workgroups on tiles of 2048; <4,1>: chunk sizes below ~448, h-only LDS rows). */
const char *bas_render_fused_kernel_name(int n_src, long T_in, int K, int S, int L);
size_t bas_render_fused_workspace_bytes(int n_src, long T_in, int K, int S, int L);
int bas_render_mix_fused_f32(const float *x, long x_stride, const float *packed,
                             const void *plans, int n_src, long T_in, int K, int S, int L,
                             int U, int ndir, float *y, int accumulate, float *peak, int normalize,
                             void *ws, size_t ws_bytes, bas_stream_t stream);

/* Same call; additionally records the caller's hipEvent_t pair (void*, either may be NULL) on `stream` immediately
 * before and after the FIR kernel (benchmarks time the dominant kernel live with HIP events). */
int bas_render_mix_fused_profiled_f32(const float *x, long x_stride, const float *packed,
                                      const void *plans, int n_src, long T_in, int K, int S, int L,
                                      int U, int ndir, float *y, int accumulate, float *peak,
                                      int normalize, void *ws, size_t ws_bytes, bas_stream_t stream,
                                      void *ev_begin, void *ev_end);

/* The two halves of bas_render_mix_fused_f32 as entry points of their own, same arguments: _fir_ launches the FIR kernel
 * (partial tiles into the workspace - or y itself, complete with peak and rule, for scenes whose tiles are each finished
 * by one workgroup, e.g. a single source), _reduce_ the fixed-order sum of the partial tiles into y with max|y| and the
 * peak rule.  Called back to back on one stream they ARE bas_render_mix_fused_f32; apart, a caller can put other work
 * of its own between them or on a second stream beside either (the read plans of its next block, the carry of its
 * previous one).  _reduce_ reads only the sizes, y, accumulate, peak, normalize and the workspace. */
int bas_render_fused_fir_f32(const float *x, long x_stride, const float *packed, const void *plans,
                             int n_src, long T_in, int K, int S, int L, int U, int ndir, float *y,
                             int accumulate, float *peak, int normalize, void *ws, size_t ws_bytes,
                             bas_stream_t stream);
int bas_render_fused_reduce_f32(const float *x, long x_stride, const float *packed, const void *plans,
                                int n_src, long T_in, int K, int S, int L, int U, int ndir, float *y,
                                int accumulate, float *peak, int normalize, void *ws, size_t ws_bytes,
                                bas_stream_t stream);

/* Device-side error record of a workspace (its control block): synchronises `stream`, returns 0 when no kernel that
 * used this workspace has reported anything since the last call, else a positive code (hipErrorLaunchTimeOut; text in
 * bas_last_error) and clears the record.  What can be reported: a stager wave that never received its neighbour's
 * boundary chunk IR inside a fused kernel (the affected outputs then hold NaN, never plausible audio), a kernel tail
 * whose late workgroups never saw the others arrive (peak rule not applied).  Neither has ever been observed; both used
 * to fall through silently.  Costs a stream synchronisation: call it where the caller synchronises anyway. */
int bas_render_status(void *ws, size_t ws_bytes, bas_stream_t stream);

/* ---- a7 (vii): peak normalisation (apply_hrtf.py:462-464) -------------------
 * m = max|y| over n floats; *peak = m (device float, may be NULL when apply);
 * if apply and m > 1: y /= m. */
int bas_peak_normalize_f32(float *y, long n, float *peak, int apply, bas_stream_t stream);

/* y /= *peak if *peak > 1 (second half of the rule, for a peak already known,
 * e.g. reduced across GPUs). */
int bas_scale_by_peak_f32(float *y, long n, const float *peak, bas_stream_t stream);

/* ---- multi-GPU combine (no reference counterpart; DESIGN.md "Multi-GPU") ------
 * y[i] = parts[0][i] + parts[1][i] + ... in that fixed order (deterministic),
 * parts[p] at parts + p*part_stride, n floats each; *peak = max|y| (may be NULL).
 * Used on the root rank after the RCCL gather of the per-GPU partial mixes. */
int bas_mix_partials_f32(const float *parts, int n_parts, long part_stride, long n, float *y,
                         float *peak, bas_stream_t stream);

/* The same sum, max|y| and - with normalize != 0 - the peak rule (apply_hrtf.py:462-464) in ONE launch: what the root
 * rank runs on the gathered partial mixes (bas_mix_partials_f32 + bas_scale_by_peak_f32 are a memset and two launches).
 * y and ws 16-byte aligned; ws: bas_mix_workspace_bytes() bytes whose first BAS_WS_CONTROL_BYTES are zero when first
 * used (the control block, as for bas_render_mix_fused_f32; a fused-render workspace may be passed). */
size_t bas_mix_workspace_bytes(void);
int bas_mix_finish_f32(const float *parts, int n_parts, long part_stride, long n, float *y, float *peak,
                       int normalize, void *ws, size_t ws_bytes, bas_stream_t stream);

/* ---- streaming: carried state of block-wise rendering -------
 * No reference counterpart: the reference renders one whole signal held in RAM
 * (apply_hrtf.py:405-414); its chunk loop is causal (:431-453), so a stream is rendered
 * as windows [halo | block] with halo = (L-1) rounded up to chunks.  After a window's
 * render this ONE launch moves everything that crosses the block boundary:
 *   running_peak = max(running_peak, max|y[e][halo .. halo+B)|)   (the peak of
 *       apply_hrtf.py:462 over the samples EMITTED so far; may be NULL)
 *   x[s][0 .. halo)  = x[s][B .. B+halo)           (input halo, every source)
 *   last[0][s], last[1][s] = elev/azim[s][nh+nb-1] (the angles at the block's end)
 *   elev/azim[s][0 .. nh) = elev/azim[s][nb-1 .. nb-1+nh)   (halo chunk boundaries)
 * x [n_src] rows, stride x_stride >= halo+B; elev/azim f64 [n_src] rows of nh+nb angles,
 * stride ang_stride; last f64 [2][n_src]; y [2] rows of the window's output, stride
 * y_stride; nh = halo / K, nb = B / K + 1. */
int bas_stream_epilogue_f32(float *x, long x_stride, int n_src, int halo, long B, double *elev,
                            double *azim, long ang_stride, int nh, int nb, double *last,
                            const float *y, long y_stride, float *running_peak,
                            bas_stream_t stream);

/* One block of a stream in one call (behind read plans of the window's chunk boundaries,
 * bas_interp2d_plan_angles_f32 over elev/azim [n_src][nh+nb]): the fused render of the
 * window x[s][0 .. T_in), T_in = halo + B, into y [2][T_in+L-1] - overwritten, no peak of
 * the window, no peak rule: a stream's samples leave before apply_hrtf.py:462-464 could
 * know its peak - followed by everything bas_stream_epilogue_f32 does (same arguments,
 * y_stride = T_in+L-1).  Where the scene's FIR kernel leaves slabs, the carried state and
 * the running peak ride in the reduce kernel (no launch of their own: 4.4 us of a
 * 256 x 512 real-time block's 30); where it writes y itself, the epilogue kernel is
 * launched.  Same results either way.  Sizes must be served by the fused kernels
 * (bas_render_fused_supported(n_src, T_in, K, S, L)); workspace as for
 * bas_render_mix_fused_f32.  x is WRITTEN (its first halo samples).
 * _profiled: HIP events around the FIR kernel (bench.py). */
int bas_render_stream_block_f32(float *x, long x_stride, const float *packed, const void *plans,
                                int n_src, long T_in, int K, int S, int L, int U, int ndir,
                                float *y, void *ws, size_t ws_bytes, int halo, double *elev,
                                double *azim, long ang_stride, int nh, int nb, double *last,
                                float *running_peak, bas_stream_t stream);
int bas_render_stream_block_profiled_f32(float *x, long x_stride, const float *packed,
                                         const void *plans, int n_src, long T_in, int K, int S,
                                         int L, int U, int ndir, float *y, void *ws,
                                         size_t ws_bytes, int halo, double *elev, double *azim,
                                         long ang_stride, int nh, int nb, double *last,
                                         float *running_peak, bas_stream_t stream,
                                         void *ev_begin, void *ev_end);

/* ---- table builder: the heavy parts of upsample_irs.m ---------
 * PARITY UNPINNED (no Octave, no IRCAM data in the build: upsample_irs.py's header).  All
 * arrays float64 on the device; h = the 2 Lh + 1 taps of the resampling filter Octave's
 * resample(x, p, 1) designs (upsample_irs.py: octave_resample_filter), made on the host.
 *
 * bas_resample_up_f64: y[r][j] = sum_k h[j + Lh - p k] x[r][k], j < lx p: `rows` signals
 *   of lx samples resampled by p (upsample_irs.m:37-44: the 2 x 187 HRIRs).
 * bas_delaydiffs_f64: diffs [n_dir][n_dir] (overwritten) = the antisymmetric matrix of
 *   delay differences (upsample_irs.m:15-32, :58-101): for every pair i < j the
 *   cross-correlation of irs[i], irs[j] ([n_dir][n_taps]), resampled by p, its FIRST
 *   maximum refined by a parabola; diffs[i][j] = peak / p - (n_taps - 1), diffs[j][i] =
 *   -diffs[i][j], zero diagonal.  status: ONE 64-bit word on the device (overwritten,
 *   8-byte aligned): 0, or - where a pair hit one of the reference's preconditions -
 *   ((n_dir^2 - (i n_dir + j)) << 2) | code for the failing pair with the smallest (i, j)
 *   (the same one whichever workgroup ran first); code 1: peak at the edge of the
 *   correlation's support (:70), 2: the middle point is not the first maximum (:92-93),
 *   3: three collinear points (:98).  Failing pairs' entries stay zero.  One ear per call. */
int bas_resample_up_f64(const double *x, int rows, int lx, const double *h, int Lh, int p,
                        double *y, bas_stream_t stream);
int bas_delaydiffs_f64(const double *irs, int n_dir, int n_taps, const double *h, int Lh, int p,
                       double *diffs, unsigned long long *status, bas_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BAS_H */
