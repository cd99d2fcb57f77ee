"""Multi-GPU render: one process per GPU, sources sharded, one gather of the stereo mix.

The reference is a single-process loop; nothing here has a
reference counterpart.  Independent sources shard with no data-path exchange: rank g
renders its own sources into a partial stereo mix [2, T_out]; the only collective is
ONE gather of those partial mixes to the root rank (RCCL over xGMI when the backend
is "nccl": seven peers deliver on seven distinct links, so the gather is bounded by
one rank's 8*T_out bytes on one link), followed on the root by a fixed-order sum
(deterministic) with the peak fused, and the reference's peak rule
(apply_hrtf.py:462-464) on the final mix.
"""
import numpy as np


def shard_sources(n_src, world_size, rank, root_weight=1.0, root=0):
    """Contiguous block of source indices owned by `rank`.  root_weight = 1: balanced blocks.  root_weight < 1: rank
    `root` - which also receives the gather and runs the fixed-order sum of the partial mixes while the others are
    already rendering their next step - takes that fraction of an equal share and the others split the rest evenly
    (root_weight = 0.75 at 8 ranks of a 256-source scene: 24 sources on the root, 33 or 34 on the others)."""
    if not (0.0 <= root_weight <= 1.0):
        raise ValueError("root_weight must lie in [0, 1]")
    if world_size == 1 or root_weight == 1.0:
        base, extra = divmod(n_src, world_size)
        lo = rank * base + min(rank, extra)
        return range(lo, lo + base + (1 if rank < extra else 0))
    n_root = int(round(root_weight * n_src / world_size))
    n_root = max(min(n_root, n_src - (world_size - 1)), 0) if n_src >= world_size else 0
    base, extra = divmod(n_src - n_root, world_size - 1)
    sizes = []
    k = 0
    for r in range(world_size):
        if r == root:
            sizes.append(n_root)
        else:
            sizes.append(base + (1 if k < extra else 0))
            k += 1
    lo = sum(sizes[:rank])
    return range(lo, lo + sizes[rank])


_MIX_WS = {}


def _mix_workspace(device):
    """Control area of bas_mix_finish_f32, one per (device, stream)."""
    import torch
    from . import _hip
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _MIX_WS.get(key)
    if ws is None:
        ws = _MIX_WS[key] = _hip.new_workspace(_hip.lib().bas_mix_workspace_bytes(), device)
    return ws


def _hip_mix_partials(parts, normalize=False):
    """parts [P, 2, T] device tensor -> (y [2, T], peak [1]) via libbas_hip: fixed-order sum, max|y| and (normalize) the
    peak rule apply_hrtf.py:462-464 in ONE launch (bas_mix_finish_f32)."""
    import torch
    from . import _hip
    p, _, t = parts.shape
    y = torch.empty((2, t), dtype=torch.float32, device=parts.device)
    peak = torch.empty((1,), dtype=torch.float32, device=parts.device)
    with _hip.on_device(parts.device):
        ws = _mix_workspace(parts.device)
        _hip.call("bas_mix_finish_f32", _hip.ptr(parts), p, 2 * t, 2 * t, _hip.ptr(y), _hip.ptr(peak), int(bool(normalize)),
                  _hip.ptr(ws), ws.numel(), _hip.current_stream(parts.device))
    return y, peak


def _hip_scale_by_peak(y, peak):
    from . import _hip
    with _hip.on_device(y.device):
        _hip.call("bas_scale_by_peak_f32", _hip.ptr(y), y.numel(), _hip.ptr(peak), _hip.current_stream(y.device))
    return y


def gather_mix(partial, group=None, dst=0, mix_fn=None, scale_fn=None, normalize="mix", return_peak=False):
    """Combine per-rank partial mixes [2, T_out] on rank `dst`.

    Returns the final (T_out, 2) mix on rank dst and None elsewhere (with return_peak: the pair
    (mix, peak tensor [1] of the un-normalised mix)).  mix_fn / scale_fn default to the HIP
    library (sum, max|y| and the peak rule in one launch); CPU tests inject numpy stand-ins to
    exercise the collective under gloo.
    """
    import torch
    import torch.distributed as dist
    fused_rule = mix_fn is None and scale_fn is None           # the library's mix applies the rule in the same launch
    mix_fn = mix_fn or (lambda parts: _hip_mix_partials(parts, normalize == "mix"))
    scale_fn = scale_fn or _hip_scale_by_peak
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if not (0 <= dst < world):
        raise ValueError(f"dst={dst} is not a rank of a group of {world}")
    partial = partial.contiguous()                             # mix_fn and the collective assume dense [2, T]
    if not dist.is_initialized():
        y, peak = mix_fn(partial.unsqueeze(0))
    else:                                                      # (also for a one-rank group: the same code path at every size)
        rank = dist.get_rank(group)
        # gloo cannot gather device tensors: stage through host memory (debugging / rehearsal on boxes
        # without RCCL peers; the production backend "nccl" gathers device to device over xGMI)
        via_host = partial.is_cuda and dist.get_backend(group) == "gloo"
        send = partial.cpu() if via_host else partial
        if rank == dst:
            parts = torch.empty((world,) + tuple(send.shape), dtype=send.dtype, device=send.device)
            dist.gather(send, gather_list=list(parts.unbind(0)), dst=dst, group=group)
            y, peak = mix_fn(parts.to(partial.device) if via_host else parts)
        else:
            dist.gather(send, gather_list=None, dst=dst, group=group)
            return None
    if normalize == "mix" and not fused_rule:
        y = scale_fn(y, peak)
    return (y.t(), peak) if return_peak else y.t()


def render_sources_sharded(signals, chunksize, subchunksize, elev, azim, tbl, group=None, dst=0,
                           normalize="mix", render_fn=None, mix_fn=None, scale_fn=None):
    """Render THIS rank's sources (`signals`, `elev`, `azim` hold only the local shard)
    and gather the mix on rank dst.  render_fn(signals, K, S, elev, azim, tbl) must return
    the un-normalised local mix as (T_out, 2); default: the HIP renderer."""
    if render_fn is None:
        from .apply_hrtf import render_sources
        render_fn = lambda s, k, ss, e, a, t: render_sources(s, k, ss, e, a, t, normalize="none")   # noqa: E731
    local = render_fn(signals, chunksize, subchunksize, elev, azim, tbl)
    return gather_mix(local.t(), group=group, dst=dst, mix_fn=mix_fn, scale_fn=scale_fn, normalize=normalize)


# --------------------------------------------------------------------------
# by time: one long scene, every rank renders ALL sources over its own time range
# --------------------------------------------------------------------------
def shard_time(n_chunks, world_size, rank):
    """Chunk range [c0, c1) owned by `rank` (balanced, contiguous)."""
    base, extra = divmod(n_chunks, world_size)
    c0 = rank * base + min(rank, extra)
    return c0, c0 + base + (1 if rank < extra else 0)


def render_time_sharded(signals, chunksize, subchunksize, elev, azim, tbl, ir_length, group=None, dst=0,
                        normalize="mix", render_fn=None, scale_fn=None):
    """Every rank holds the whole (padded) scene description but renders only outputs
    [c0*K, c1*K) (the last rank also the L-1 tail): it reads its inputs with a halo of
    ceil((L-1)/K) chunks on the left, exactly like StreamRenderer, so slices need no seam
    addition.  The slices meet in ONE gather on rank dst (disjoint ranges: concatenation),
    followed by the peak rule over the whole signal.

    signals [n_src, N]; elev/azim float64 [n_src, n_chunks+1] for the padded length.
    render_fn(signals, K, S, elev, azim, tbl) -> un-normalised (T_out, 2); default HIP renderer.
    Returns (T_out, 2) on rank dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    if render_fn is None:
        from .apply_hrtf import render_sources
        render_fn = lambda s, k, ss, e, a, t: render_sources(s, k, ss, e, a, t, normalize="none")   # noqa: E731
    scale_fn = scale_fn or (lambda y, peak: _hip_scale_by_peak(y.contiguous(), peak))
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    K, L = chunksize, ir_length
    sig = torch.as_tensor(signals)
    n_src, n = sig.shape
    n_chunks = -(-n // K)
    halo_c = -(-(L - 1) // K) if L > 1 else 0
    c0, c1 = shard_time(n_chunks, world, rank)
    h0 = max(c0 - halo_c, 0)                                  # first chunk read
    x = sig[:, h0 * K:min(c1 * K, n)]
    if x.shape[1] < (c1 - h0) * K:                            # pad the scene's last chunk (apply_hrtf.py:405-406)
        x = torch.cat([x, torch.zeros((n_src, (c1 - h0) * K - x.shape[1]), dtype=x.dtype, device=x.device)], dim=1)
    e = np.asarray(elev)[:, h0:c1 + 1]
    a = np.asarray(azim)[:, h0:c1 + 1]
    y = render_fn(x, K, subchunksize, e, a, tbl)              # ((c1-h0)*K + L-1, 2), complete from (c0-h0)*K on
    lo = (c0 - h0) * K
    hi = (c1 - h0) * K + (L - 1 if rank == world - 1 else 0)
    mine = y[lo:hi].contiguous()
    if not dist.is_initialized():
        full = mine
    else:                                                     # (also for a one-rank group: the same code path at every size)
        sizes = [(shard_time(n_chunks, world, r)[1] - shard_time(n_chunks, world, r)[0]) * K +
                 (L - 1 if r == world - 1 else 0) for r in range(world)]
        pad = max(sizes)
        buf = torch.zeros((pad, 2), dtype=mine.dtype, device=mine.device)
        buf[:mine.shape[0]] = mine
        # gloo cannot gather device tensors: stage through host memory (rehearsal on boxes without RCCL peers;
        # the production backend "nccl" gathers device to device over xGMI)
        via_host = buf.is_cuda and dist.get_backend(group) == "gloo"
        send = buf.cpu() if via_host else buf
        if rank == dst:
            parts = [torch.empty_like(send) for _ in range(world)]
            dist.gather(send, gather_list=parts, dst=dst, group=group)
            full = torch.cat([parts[r][:sizes[r]] for r in range(world)], dim=0)
            if via_host:
                full = full.to(mine.device)
        else:
            dist.gather(send, gather_list=None, dst=dst, group=group)
            return None
    if normalize == "mix":
        peak = full.abs().max().reshape(1)
        full = scale_fn(full, peak)
    return full


# --------------------------------------------------------------------------
# streaming (BASELINE config 5): sources sharded, one gather per block
# --------------------------------------------------------------------------
class ShardedStreamRenderer:
    """StreamRenderer over the ranks of a process group: rank g streams its own block of sources
    (`sources`, from shard_sources) with its own carried state; each process()/finish() ends in one
    gather of the [2, B] partial block to rank dst and the fixed-order sum there.  Audio is returned
    un-normalised on rank dst (None elsewhere); `peak` is the running max |sample| of the MIX.

    stream_factory(tbl, n_local, K, S) defaults to StreamRenderer; mix_fn as in gather_mix (CPU tests
    inject stand-ins for both)."""

    def __init__(self, tbl, n_src_total, chunksize, subchunksize, group=None, dst=0, stream_factory=None,
                 mix_fn=None, root_weight=1.0):
        import torch.distributed as dist
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        if n_src_total < world:
            raise ValueError(f"{n_src_total} sources cannot be sharded over {world} ranks")
        self.sources = shard_sources(n_src_total, world, rank, root_weight=root_weight, root=dst)
        if stream_factory is None:
            from .stream import StreamRenderer
            stream_factory = StreamRenderer
        self.local = stream_factory(tbl, len(self.sources), chunksize, subchunksize)
        self.group, self.dst, self.mix_fn = group, dst, mix_fn
        self._peak = None
        # no process group: the local stream IS the mix (its `peak` is then the mix's: a stand-in without that
        # attribute goes through gather_mix, which measures the peak itself)
        has_peak = hasattr(type(self.local), "peak") or "peak" in getattr(self.local, "__dict__", {})
        self._solo = not dist.is_initialized() and mix_fn is None and has_peak

    def _combine(self, out_local):
        if self._solo:
            return out_local
        res = gather_mix(out_local.t(), group=self.group, dst=self.dst, mix_fn=self.mix_fn, normalize="none",
                         return_peak=True)
        if res is None:
            return None
        y, peak = res
        self._peak = peak if self._peak is None else self._peak.maximum(peak)
        return y

    def process(self, block_local, elev_local, azim_local):
        """block_local [len(self.sources), B] and the trajectories of THIS rank's sources."""
        return self._combine(self.local.process(block_local, elev_local, azim_local))

    def finish(self):
        return self._combine(self.local.finish())

    @property
    def peak(self):
        if self._solo:
            return self.local.peak
        return None if self._peak is None else float(self._peak.reshape(-1)[0])
