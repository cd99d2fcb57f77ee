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
"""
from . import sphere
from .apply_hrtf import as_device_table, render_params_device


class StreamRenderer:
    def __init__(self, tbl, n_src, chunksize, subchunksize):
        import torch
        assert chunksize % subchunksize == 0, 'subchunksize does not divide chunksize evenly'
        self.tbl = as_device_table(tbl)
        self.n_src, self.K, self.S = int(n_src), int(chunksize), int(subchunksize)
        L = self.tbl.L
        self.halo = -(-(L - 1) // self.K) * self.K if L > 1 else 0
        dev = self.tbl.device
        # input staging buffer [n_src, halo + capacity]: columns [0, halo) carry the previous inputs, a block
        # is rendered in place behind them (input_view() lets a producer write there directly: no copy)
        self._xbuf = torch.zeros((self.n_src, self.halo), dtype=torch.float32, device=dev)
        self._idx_halo = None          # parameters at the halo's chunk boundaries, [n_src, halo/K, 4|3]
        self._w_halo = None
        self._idx_last = self._w_last = None
        self._peak_dev = torch.zeros((), dtype=torch.float32, device=dev)
        self.samples_in = 0
        self._finished = False

    def _reserve(self, B):
        import torch
        cap = self._xbuf.shape[1] - self.halo
        if cap < B:
            grown = torch.zeros((self.n_src, (self.halo + B + 3) // 4 * 4), dtype=torch.float32, device=self._xbuf.device)
            grown[:, :self.halo] = self._xbuf[:, :self.halo]
            self._xbuf = grown

    def input_view(self, B):
        """Device view [n_src, B] of the renderer's own input buffer.  A producer (decoder, H2D copy,
        another kernel) that writes the next block here and passes this view to process() saves the
        staging copy of the block; the view is valid until the next input_view() call with a larger B."""
        self._reserve(B)
        return self._xbuf[:, self.halo:self.halo + B]

    def process(self, block, elev, azim):
        """block: [n_src, B] (B a multiple of the chunk size); elev/azim: float64 [n_src, B/K + 1],
        the trajectory at t = t0, t0+K, .., t0+B of this block (radians).  Returns the B stereo
        samples this block completes as a device tensor (B, 2), un-normalised."""
        import torch
        assert not self._finished, "stream already finished"
        blk = torch.as_tensor(block)
        assert blk.dim() == 2 and blk.shape[0] == self.n_src, 'block must be [n_src, B]'
        B = blk.shape[1]
        assert B % self.K == 0 and B > 0, 'block length must be a positive multiple of the chunk size'
        dev = self.tbl.device
        # angles -> (indices, weights): on the device when the trajectory already lives there (long
        # streams: no host work per block), else with the host's vectorised form; both are bit-identical
        if isinstance(elev, torch.Tensor) and elev.is_cuda:
            idx, w = sphere.interpolation_params_device(elev.to(torch.float64), torch.as_tensor(azim).to(dev, torch.float64))
        else:
            idx_h, w_h = sphere.interpolation_params_batch(elev, azim)
            idx, w = torch.from_numpy(idx_h).to(dev), torch.from_numpy(w_h).to(dev)
        nb = B // self.K + 1
        if tuple(idx.shape[:2]) != (self.n_src, nb):
            raise ValueError(f"elev/azim must have shape ({self.n_src}, {nb})")
        nh = self.halo // self.K
        if self._idx_halo is None:      # first block: the halo holds silence, any valid IR will do
            self._idx_halo = idx[:, :1].repeat(1, nh, 1)
            self._w_halo = w[:, :1].repeat(1, nh, 1)
        idx_all = torch.cat([self._idx_halo, idx], dim=1)              # boundaries t0-halo .. t0+B
        w_all = torch.cat([self._w_halo, w], dim=1)
        self._reserve(B)
        x = self._xbuf[:, :self.halo + B]
        in_place = blk.is_cuda and blk.dtype == torch.float32 and blk.stride() == x.stride() and \
            blk.data_ptr() == self._xbuf.data_ptr() + 4 * self.halo
        if not in_place:
            x[:, self.halo:] = blk.to(device=dev, dtype=torch.float32)
        y, _ = render_params_device(x, self.K, self.S, self.tbl, idx_all.reshape(-1, 4).contiguous(),
                                    w_all.reshape(-1, 3).contiguous(), normalize="none")
        out = y[:, self.halo:self.halo + B]
        # carry: last `halo` inputs and the parameters of their chunk boundaries (t0+B-halo .. t0+B-K)
        if self.halo:
            tail = x[:, B:B + self.halo]
            self._xbuf[:, :self.halo] = tail.clone() if B < self.halo else tail      # ranges overlap only if B < halo
            self._idx_halo = idx_all[:, nb - 1:nb - 1 + nh].clone()
            self._w_halo = w_all[:, nb - 1:nb - 1 + nh].clone()
        self._idx_last, self._w_last = idx[:, -1:].clone(), w[:, -1:].clone()   # boundary t0 + B
        self._peak_dev = torch.maximum(self._peak_dev, out.abs().max()) if out.numel() else self._peak_dev
        self.samples_in += B
        return out.t()

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
        if self._idx_last is None:
            raise RuntimeError("finish() before any block")
        L = self.tbl.L
        dev = self.tbl.device
        idx_all = torch.cat([self._idx_halo, self._idx_last, self._idx_last], dim=1)
        w_all = torch.cat([self._w_halo, self._w_last, self._w_last], dim=1)
        x = torch.cat([self._xbuf[:, :self.halo], torch.zeros((self.n_src, self.K), dtype=torch.float32, device=dev)], dim=1)
        y, _ = render_params_device(x, self.K, self.S, self.tbl, idx_all.reshape(-1, 4).contiguous(),
                                    w_all.reshape(-1, 3).contiguous(), normalize="none")
        out = y[:, self.halo:self.halo + L - 1]
        self._peak_dev = torch.maximum(self._peak_dev, out.abs().max()) if out.numel() else self._peak_dev
        self._finished = True
        return out.t()
