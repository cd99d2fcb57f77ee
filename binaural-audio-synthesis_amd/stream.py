"""Block-wise (streaming) rendering with carried state.

The reference renders one whole signal held in RAM (apply_hrtf.py:405-414) but its chunk
loop is causal (:431-453): output sample n depends on inputs n-L+1 .. n and on the chunk
IRs around them.  StreamRenderer therefore keeps, per source, the last `halo` input samples
(halo = (L-1) rounded up to a multiple of the chunk size) and the trajectory angles of the
chunk boundaries inside that halo; every call renders [halo | new block] with the same
kernels and emits exactly the outputs the new block completes.  Concatenating the emitted
blocks (plus `finish()`) reproduces the whole-signal render sample for sample
(tests/test_gpu_parity.py::test_streaming_equals_whole).

The reference's peak rule (apply_hrtf.py:462-464) is global over the finished signal and
cannot be applied to samples already handed out; the stream returns un-normalised audio
and tracks the running peak (`peak`), so a caller can scale afterwards exactly as the
reference would.  Long streams (BASELINE config 5: 1 h at 48 kHz, 1024 sources) never
materialise more than one block of inputs, chunk IRs and outputs.

One block = three launches on persistent buffers (bas_render_stream_block_f32): angles ->
read plans, the fused chunk-IR / FIR / mix kernel, and its slab reduce, which also takes the
running peak over the emitted samples and makes every carry copy (scenes whose FIR kernel
writes y itself, and shapes the fused kernels do not serve, end in bas_stream_epilogue_f32
instead).  With graph=True those are replayed as ONE hipGraph launch.  `prepare(B)` lays the buffers out and
captures the graph BEFORE streaming starts (capture synchronises the device and must not
race with allocations of other threads: keep it out of the real-time phase); without it the
first block of a size runs as plain launches and the second one captures.

Why the halo is a whole number of CHUNKS (K) rather than L-1 rounded to 32: the kernels
take windows whose first sample lies on a chunk boundary (the crossfade position of an input
is its offset inside its chunk, apply_hrtf.py:442), and their work is quantised by output
tiles of 2048 samples anyway - a 512-sample block with a 512-sample halo is ONE tile per
source, exactly as it would be with a 128-sample halo.
"""
from . import _hip
from .apply_hrtf import as_device_table, plan_angles_device, render_angles_device


def tile_filling_block(about, chunksize, ir_length, tile=8192):
    """The largest block length <= `about` (a multiple of the chunk size) whose window - [halo | block] inputs, L - 1 more
    outputs - ends on a tile boundary of the big scenes' FIR kernel (8192 outputs per (tile, source) unit) or just before it.
    A window that spills a few samples into one more tile pays for the whole tile: 2^18-sample blocks with K = 512, L = 128
    are 32.08 tiles, rendered as 33 (+ 3 %: 3 661 against 3 788 x real time for BASELINE config 5); 261 120 are 31.95.
    Returns `about` rounded down to chunks where no whole tile fits."""
    K, L = int(chunksize), int(ir_length)
    halo = -(-(L - 1) // K) * K if L > 1 else 0
    about = int(about) // K * K
    tiles = (halo + about + L - 1) // tile
    best = (tiles * tile - (L - 1) - halo) // K * K
    return best if tiles >= 1 and best >= K else max(about, K)


class StreamRenderer:
    one_call = True      # bas_render_stream_block_f32 where the fused kernels serve the block (False: render + epilogue launch; A/B, tests)

    def __init__(self, tbl, n_src, chunksize, subchunksize, graph=True, copy_out=True):
        """graph: replay each block as one captured hipGraph (captured by prepare(), else on the second block of a
        size).  copy_out: process() returns a fresh tensor (True) or a view of the renderer's output buffer that
        the next process() call overwrites (False: no copy kernel; for callers that consume each block at once)."""
        import torch
        assert chunksize % subchunksize == 0, 'subchunksize does not divide chunksize evenly'
        self.tbl = as_device_table(tbl)
        self.n_src, self.K, self.S = int(n_src), int(chunksize), int(subchunksize)
        L = self.tbl.L
        self.halo = -(-(L - 1) // self.K) * self.K if L > 1 else 0
        self.nh = self.halo // self.K                     # chunk boundaries carried with the halo
        dev = self.tbl.device
        self.graph_enabled, self.copy_out = bool(graph), bool(copy_out)
        # input staging buffer [n_src, halo + capacity]: columns [0, halo) carry the previous inputs, a block
        # is rendered in place behind them (input_view() lets a producer write there directly: no copy)
        self._xbuf = torch.zeros((self.n_src, self.halo), dtype=torch.float32, device=dev)
        self._B = None                                    # block size the per-block buffers are laid out for
        self._graph = None
        self._halo_params = None                          # (elev, azim) [n_src, nh] of the halo's boundaries across a re-layout
        self._started = False                             # a block has been rendered
        # the angles at the END of the last block, for finish(): their own buffer, so that a change of block size
        # (which re-allocates the per-block angle buffers) cannot lose them
        self._last = torch.zeros((2, self.n_src), dtype=torch.float64, device=dev)
        self._peak_dev = torch.zeros((1,), dtype=torch.float32, device=dev)
        self.samples_in = 0
        self._finished = False
        self._events = None                               # (begin, end) raw hipEvent_t around the FIR kernel of plain-launch blocks (bench.py)

    # ---- buffers ---------------------------------------------------------------------------------------
    def _reserve(self, B):
        import torch
        cap = self._xbuf.shape[1] - self.halo
        if cap < B:
            grown = torch.zeros((self.n_src, (self.halo + B + 3) // 4 * 4), dtype=torch.float32, device=self._xbuf.device)
            grown[:, :self.halo] = self._xbuf[:, :self.halo]
            self._xbuf = grown
            self._graph = None                            # the captured pointers are gone

    def _layout(self, B):
        """Per-block buffers for blocks of B samples (kept until another size arrives)."""
        import torch
        if self._B == B:
            return
        dev, n, nh = self.tbl.device, self.n_src, self.nh
        nb = B // self.K + 1
        if self._B is not None and self._started:         # carry the halo's angles into the new layout
            self._halo_params = (self._elev_all[:, :nh].clone(), self._azim_all[:, :nh].clone())
        self._reserve(B)
        self._B, self._nb, self._graph, self._blocks_in_layout = B, nb, None, 0
        # trajectory at the chunk boundaries t0-halo .. t0+B: [halo part carried | this block's part].  Zero is a
        # valid direction: before the first block the halo holds silence and its chunk IRs only multiply zeros.
        self._elev_all = torch.zeros((n, nh + nb), dtype=torch.float64, device=dev)
        self._azim_all = torch.zeros((n, nh + nb), dtype=torch.float64, device=dev)
        if self._halo_params is not None:
            self._elev_all[:, :nh], self._azim_all[:, :nh] = self._halo_params
            self._halo_params = None
        self._idx = torch.empty((n * (nh + nb), 4), dtype=torch.int32, device=dev)
        self._w = torch.empty((n * (nh + nb), 3), dtype=torch.float64, device=dev)
        self._y = torch.empty((2, self.halo + B + self.tbl.L - 1), dtype=torch.float32, device=dev)
        lib = _hip.lib()
        t_in = self.halo + B
        with _hip.on_device(dev):
            wb = max(lib.bas_render_workspace_bytes(n, t_in, self.K, self.S, self.tbl.L),
                     lib.bas_render_fused_workspace_bytes(n, t_in, self.K, self.S, self.tbl.L))
        self._ws = _hip.new_workspace(wb, dev)
        self._ws_plans = torch.empty((lib.bas_interp2d_workspace_bytes(n * (nh + nb)),), dtype=torch.uint8, device=dev)

    def input_view(self, B):
        """Device view [n_src, B] of the renderer's own input buffer.  A producer (decoder, H2D copy,
        another kernel) that writes the next block here and passes this view to process() saves the
        staging copy of the block; the view is valid until the next input_view() call with a larger B.
        Growing the buffer allocates and drops the captured graph: size it once, before prepare()."""
        self._reserve(B)
        return self._xbuf[:, self.halo:self.halo + B]

    def trajectory_views(self, B):
        """Device views (elev, azim), float64 [n_src, B/K + 1], of the renderer's own trajectory buffers for
        blocks of B samples (strided: they sit behind the carried halo boundaries): a producer that fills them in
        place and passes them to process() saves two copies."""
        self._layout(B)
        return self._elev_all[:, self.nh:], self._azim_all[:, self.nh:]

    # ---- one block -------------------------------------------------------------------------------------
    def _block_body(self):
        """The stream-ordered work of one block on the per-block buffers (captured into the hipGraph)."""
        B, nb, nh, halo = self._B, self._nb, self.nh, self.halo
        dev = self.tbl.device
        x = self._xbuf[:, :halo + B]
        tbl, n = self.tbl, self.n_src
        lib = _hip.lib()
        with _hip.on_device(dev):
            one_call = self.one_call and bool(lib.bas_render_fused_supported(n, halo + B, self.K, self.S, tbl.L)) and tbl.upsampling >= 4 \
                and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
        if one_call:
            # read plans (a3 inside), then ONE call: chunk IRs + FIR + mix, and behind the sums of its reduce kernel the
            # running peak over the emitted samples + the carry of the last `halo` inputs and of the angles at their chunk
            # boundaries (t0+B-halo .. t0+B-K) + the angles at t0+B for finish()
            plan_angles_device(tbl, self._elev_all, self._azim_all, self._ws_plans)
            args = (_hip.ptr(self._xbuf), self._xbuf.stride(0), _hip.ptr(tbl.packed), _hip.ptr(self._ws_plans), n, halo + B,
                    self.K, self.S, tbl.L, tbl.upsampling, tbl.ndir, _hip.ptr(self._y), _hip.ptr(self._ws), self._ws.numel(),
                    halo, _hip.ptr(self._elev_all), _hip.ptr(self._azim_all), self._elev_all.stride(0), nh, nb,
                    _hip.ptr(self._last), _hip.ptr(self._peak_dev), _hip.current_stream(dev))
            with _hip.on_device(dev):
                if self._events is None:
                    _hip.call("bas_render_stream_block_f32", *args)
                else:
                    _hip.call("bas_render_stream_block_profiled_f32", *args, self._events[0], self._events[1])
            return
        # other shapes: a3, read plans, chunk IRs + FIR + mix (or the stored-IR path), then the epilogue launch
        render_angles_device(x, self.K, self.S, self.tbl, self._elev_all, self._azim_all, normalize="none",
                             out=self._y, ws=self._ws, ws_plans=self._ws_plans, params=(self._idx, self._w),
                             events=self._events, want_peak=False)   # (the epilogue tracks the peak of the EMITTED samples)
        with _hip.on_device(dev):
            _hip.call("bas_stream_epilogue_f32", _hip.ptr(self._xbuf), self._xbuf.stride(0), self.n_src, halo, B,
                      _hip.ptr(self._elev_all), _hip.ptr(self._azim_all), self._elev_all.stride(0), nh, nb,
                      _hip.ptr(self._last), _hip.ptr(self._y), self._y.stride(0), _hip.ptr(self._peak_dev),
                      _hip.current_stream(dev))

    def _capture(self):
        import torch
        assert self._events is None, "HIP events of a profiling caller cannot be captured into the block's graph"
        g = torch.cuda.CUDAGraph()
        # thread_local: allocations or copies of OTHER threads (a decoder filling input_view()) do not invalidate
        # the capture; the capture's own allocations come from the graph's private pool
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._block_body()
        self._graph = g

    def prepare(self, B):
        """Lay out the buffers for blocks of B samples, run one block on silence as a warm-up (first-use costs of the
        kernels) and, with graph=True, capture the block's hipGraph - all BEFORE streaming starts, so that no
        process() call ever pays for a capture (which synchronises the device for milliseconds).  The carried
        state (input halo, halo angles, end angles, running peak, sample count) is left exactly as it was.
        Call again after a change of block size or after input_view() had to grow.  The warm-up block is rendered
        into the renderer's own output buffer: a block view handed out by process() with copy_out=False is overwritten
        by it - consume such a view before calling prepare() mid-stream."""
        import torch
        assert not self._finished, "stream already finished"
        assert B % self.K == 0 and B > 0, 'block length must be a positive multiple of the chunk size'
        self._layout(B)
        halo, nh = self.halo, self.nh
        keep = (self._xbuf[:, :halo + B].clone(), self._elev_all.clone(), self._azim_all.clone(), self._last.clone(),
                self._peak_dev.clone())
        self._xbuf[:, halo:halo + B].zero_()
        self._block_body()                                # plain launches: warm-up
        if self.graph_enabled and self._graph is None:
            self._capture()                               # (records the launches, does not execute them)
        torch.cuda.synchronize(self.tbl.device)
        self._xbuf[:, :halo + B].copy_(keep[0])
        self._elev_all.copy_(keep[1])
        self._azim_all.copy_(keep[2])
        self._last.copy_(keep[3])
        self._peak_dev.copy_(keep[4])
        self._blocks_in_layout = max(self._blocks_in_layout, 1)

    def process(self, block, elev, azim):
        """block: [n_src, B] (B a multiple of the chunk size); elev/azim: float64 [n_src, B/K + 1],
        the trajectory at t = t0, t0+K, .., t0+B of this block (radians; numpy arrays or device tensors).
        Returns the B stereo samples this block completes as a device tensor (B, 2), un-normalised."""
        import torch
        assert not self._finished, "stream already finished"
        blk = torch.as_tensor(block)
        assert blk.dim() == 2 and blk.shape[0] == self.n_src, 'block must be [n_src, B]'
        B = blk.shape[1]
        assert B % self.K == 0 and B > 0, 'block length must be a positive multiple of the chunk size'
        self._layout(B)
        nb = self._nb
        for src, dst in ((elev, self._elev_all[:, self.nh:]), (azim, self._azim_all[:, self.nh:])):
            t = torch.as_tensor(src)
            if tuple(t.shape) != (self.n_src, nb):
                raise ValueError(f"elev/azim must have shape ({self.n_src}, {nb})")
            if not (t.is_cuda and t.data_ptr() == dst.data_ptr() and t.dtype == torch.float64 and t.stride() == dst.stride()):
                dst.copy_(t)                              # (H2D for host arrays; float64 kept exactly)
        x_dst = self._xbuf[:, self.halo:self.halo + B]
        in_place = blk.is_cuda and blk.dtype == torch.float32 and blk.stride() == x_dst.stride() and \
            blk.data_ptr() == x_dst.data_ptr()
        if not in_place:
            x_dst.copy_(blk)
        if self._graph is not None:
            self._graph.replay()
        elif not self.graph_enabled or self._blocks_in_layout == 0:
            self._block_body()                            # no prepare(): the first block of a size runs plain (warm-up)
        else:
            self._capture()                               # ... and the second one captures (a device synchronisation:
            self._graph.replay()                          # real-time callers use prepare() instead)
        self._started = True
        self._blocks_in_layout += 1
        self.samples_in += B
        out = self._y[:, self.halo:self.halo + B].t()
        return out.clone() if self.copy_out else out

    @property
    def peak(self):
        """max |sample| emitted so far (reads back one float)."""
        return float(self._peak_dev[0])

    def finish(self):
        """Emit the last L-1 samples (the tail the reference appends, apply_hrtf.py:410): render one
        silent chunk behind the stream.  The chunk boundary at the stream's end was supplied by the
        last process() call (kept in its own buffer across changes of block size); the one after it only
        multiplies silence."""
        import torch
        assert not self._finished, "stream already finished"
        if not self._started:
            raise RuntimeError("finish() before any block")
        L, nh = self.tbl.L, self.nh
        dev = self.tbl.device
        e_last, a_last = self._last[0].reshape(-1, 1), self._last[1].reshape(-1, 1)
        e_halo, a_halo = self._elev_all[:, :nh], self._azim_all[:, :nh]   # (a re-layout carries them over)
        elev = torch.cat([e_halo, e_last, e_last], dim=1).contiguous()
        azim = torch.cat([a_halo, a_last, a_last], dim=1).contiguous()
        x = torch.cat([self._xbuf[:, :self.halo], torch.zeros((self.n_src, self.K), dtype=torch.float32, device=dev)], dim=1)
        y, _ = render_angles_device(x, self.K, self.S, self.tbl, elev, azim, normalize="none", want_peak=False)
        out = y[:, self.halo:self.halo + L - 1]
        if out.numel():
            self._peak_dev = torch.maximum(self._peak_dev, out.abs().max().reshape(1))
        self._finished = True
        return out.t()
