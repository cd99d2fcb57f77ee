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


def shard_sources(n_src, world_size, rank):
    """Contiguous, balanced block of source indices owned by `rank`."""
    base, extra = divmod(n_src, world_size)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def _hip_mix_partials(parts):
    """parts [P, 2, T] device tensor -> (y [2, T], peak [1]) via libbas_hip (fixed order)."""
    import torch
    from . import _hip
    p, _, t = parts.shape
    y = torch.empty((2, t), dtype=torch.float32, device=parts.device)
    peak = torch.empty((1,), dtype=torch.float32, device=parts.device)
    _hip.call("bas_mix_partials_f32", _hip.ptr(parts), p, 2 * t, 2 * t, _hip.ptr(y), _hip.ptr(peak),
              _hip.current_stream(parts.device))
    return y, peak


def _hip_scale_by_peak(y, peak):
    from . import _hip
    _hip.call("bas_scale_by_peak_f32", _hip.ptr(y), y.numel(), _hip.ptr(peak), _hip.current_stream(y.device))
    return y


def gather_mix(partial, group=None, dst=0, mix_fn=None, scale_fn=None, normalize="mix"):
    """Combine per-rank partial mixes [2, T_out] on rank `dst`.

    Returns the final (T_out, 2) mix on rank dst and None elsewhere.  mix_fn / scale_fn
    default to the HIP library; CPU tests inject numpy stand-ins to exercise the
    collective under gloo.
    """
    import torch
    import torch.distributed as dist
    mix_fn = mix_fn or _hip_mix_partials
    scale_fn = scale_fn or _hip_scale_by_peak
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        y, peak = mix_fn(partial.unsqueeze(0))
    else:
        rank = dist.get_rank(group)
        partial = partial.contiguous()
        if rank == dst:
            parts = torch.empty((world,) + tuple(partial.shape), dtype=partial.dtype, device=partial.device)
            dist.gather(partial, gather_list=list(parts.unbind(0)), dst=dst, group=group)
            y, peak = mix_fn(parts)
        else:
            dist.gather(partial, gather_list=None, dst=dst, group=group)
            return None
    if normalize == "mix":
        y = scale_fn(y, peak)
    return y.t()


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
