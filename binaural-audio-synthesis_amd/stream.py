"""Block-wise (streaming) rendering with carried state.

The reference renders one whole signal held in RAM (apply_hrtf.py:405-414) but its chunk
loop is causal (:431-453): output sample n depends on inputs n-L+1 .. n and on the chunk
IRs around them.  StreamRenderer therefore keeps, per source, the last `halo` input samples
(halo = (L-1) rounded up to a multiple of the chunk size) and the interpolation parameters
of the chunk boundaries inside that halo; every call renders [halo | new block] with the
same kernels and emits exactly the outputs the new block completes.  Concatenating the
emitted blocks (plus `finish()`) reproduces the whole-signal render sample for sample
(tests/test_gpu_parity.py::test_streaming_equals_whole).

The reference's peak rule (apply_hrtf.py:462-464) is global over the finished signal and
cannot be applied to samples already handed out; the stream returns un-normalised audio
and tracks the running peak (`peak`), so a caller can scale afterwards exactly as the
reference would.  Long streams (BASELINE config 5: 1 h at 48 kHz, 1024 sources) never
materialise more than one block of inputs, chunk IRs and outputs.

All per-block state lives in buffers that are allocated once per block size, and one block
is a fixed sequence of stream-ordered calls on them (angles -> parameters, read plans, fused
FIR, carry copies, running peak).  From the second block of a size on that sequence is
replayed as ONE hipGraph launch (graph=True, the default): a real-time caller feeding
512-sample blocks pays one graph launch per block instead of a dozen Python-level launches.
"""
from . import sphere
from .apply_hrtf import as_device_table, render_angles_device


class StreamRenderer:
    def __init__(self, tbl, n_src, chunksize, subchunksize, graph=True, copy_out=True):
        """graph: replay each block as one captured hipGraph (from the second block of a given size on).
        copy_out: process() returns a fresh tensor (True) or a view of the renderer's output buffer that the
        next process() call overwrites (False: no copy kernel; for callers that consume each block at once)."""
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
        self._first = True
        self._peak_dev = torch.zeros((), dtype=torch.float32, device=dev)
        self.samples_in = 0
        self._finished = False

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
        from . import _hip
        if self._B == B:
            return
        dev, n, nh = self.tbl.device, self.n_src, self.nh
        nb = B // self.K + 1
        if self._B is not None and not self._first:       # carry the halo's angles into the new layout
            self._halo_params = (self._elev_all[:, :nh].clone(), self._azim_all[:, :nh].clone())
        self._reserve(B)
        self._B, self._nb, self._graph, self._blocks_in_layout = B, nb, None, 0
        # trajectory at the chunk boundaries t0-halo .. t0+B: [halo part carried | this block's part]
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
        wb = max(lib.bas_render_workspace_bytes(n, t_in, self.K, self.S, self.tbl.L),
                 lib.bas_render_fused_workspace_bytes(n, t_in, self.K, self.S, self.tbl.L))
        self._ws = torch.empty((wb,), dtype=torch.uint8, device=dev)
        self._ws_plans = torch.empty((lib.bas_interp2d_workspace_bytes(n * (nh + nb)),), dtype=torch.uint8, device=dev)

    def input_view(self, B):
        """Device view [n_src, B] of the renderer's own input buffer.  A producer (decoder, H2D copy,
        another kernel) that writes the next block here and passes this view to process() saves the
        staging copy of the block; the view is valid until the next input_view() call with a larger B."""
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
        import torch
        B, nb, nh, halo = self._B, self._nb, self.nh, self.halo
        if self._first:                                   # the halo holds silence, any valid direction will do
            self._elev_all[:, :nh] = self._elev_all[:, nh:nh + 1]
            self._azim_all[:, :nh] = self._azim_all[:, nh:nh + 1]
        x = self._xbuf[:, :halo + B]
        # a3, read plans, chunk IRs + FIR + mix (or the stored-IR path for other sizes)
        render_angles_device(x, self.K, self.S, self.tbl, self._elev_all, self._azim_all, normalize="none",
                             out=self._y, ws=self._ws, ws_plans=self._ws_plans, params=(self._idx, self._w))
        out = self._y[:, halo:halo + B]
        self._peak_dev.copy_(torch.maximum(self._peak_dev, out.abs().max()))
        # carry: last `halo` inputs and the angles at their chunk boundaries (t0+B-halo .. t0+B-K)
        if halo:
            tail = x[:, B:B + halo]
            self._xbuf[:, :halo] = tail.clone() if B < halo else tail            # ranges overlap only if B < halo
            ea, aa = self._elev_all[:, nb - 1:nb - 1 + nh], self._azim_all[:, nb - 1:nb - 1 + nh]
            overlap = nb - 1 < nh
            self._elev_all[:, :nh] = ea.clone() if overlap else ea
            self._azim_all[:, :nh] = aa.clone() if overlap else aa

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
        dev = self.tbl.device
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
        if self._first or not self.graph_enabled or (self._graph is None and self._blocks_in_layout == 0):
            self._block_body()                            # first block of a size: plain launches (also the warm-up)
            self._first = False
        else:
            if self._graph is None:                       # second consecutive block of this size: capture once
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._block_body()
                self._graph = g
            self._graph.replay()
        self._blocks_in_layout += 1
        self.samples_in += B
        out = self._y[:, self.halo:self.halo + B].t()
        return out.clone() if self.copy_out else out

    @property
    def peak(self):
        """max |sample| emitted so far (reads back one float)."""
        return float(self._peak_dev)

    def finish(self):
        """Emit the last L-1 samples (the tail the reference appends, apply_hrtf.py:410): render one
        silent chunk behind the stream.  The chunk boundary at the stream's end was supplied by the
        last process() call; the one after it only multiplies silence."""
        import torch
        assert not self._finished, "stream already finished"
        if self._first:
            raise RuntimeError("finish() before any block")
        L, nh, nb = self.tbl.L, self.nh, self._nb
        dev = self.tbl.device
        e_last, a_last = self._elev_all[:, nh + nb - 1:nh + nb], self._azim_all[:, nh + nb - 1:nh + nb]
        elev = torch.cat([self._elev_all[:, :nh], e_last, e_last], dim=1).contiguous()
        azim = torch.cat([self._azim_all[:, :nh], a_last, a_last], dim=1).contiguous()
        x = torch.cat([self._xbuf[:, :self.halo], torch.zeros((self.n_src, self.K), dtype=torch.float32, device=dev)], dim=1)
        y, _ = render_angles_device(x, self.K, self.S, self.tbl, elev, azim, normalize="none")
        out = y[:, self.halo:self.halo + L - 1]
        self._peak_dev = torch.maximum(self._peak_dev, out.abs().max()) if out.numel() else self._peak_dev
        self._finished = True
        return out.t()
