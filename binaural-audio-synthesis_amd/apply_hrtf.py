"""Host side of the MI355X renderer: the reference's apply_hrtf.py interface for the
path load_irs_and_delaydiffs -> interpolate_2d -> make_signal_move_2d.

Same function names, argument meaning and error behaviour as the reference
(file:line citations below are into the reference's apply_hrtf.py); audio, table
and parameters live in HBM as PyTorch-ROCm tensors and every arithmetic step is a
call into libbas_hip.so (include/bas.h).  There is no CPU fallback.

Differences a caller can see (all documented in DESIGN.md):
  * results are float32 end to end (the reference computes in float64 and casts
    the render to float32 at :459); parity bound 1e-5 norm-relative.
  * progress printing (:456-457) is off unless verbose=True.
  * batched forms (`interpolate_2d_batch`, `render_sources`) exist for many
    queries / many sources; the reference has none.
"""
import math

import numpy as np

from . import _hip
from . import sphere


# --------------------------------------------------------------------------
# a1: the table
# --------------------------------------------------------------------------
class irs_and_delaydiffs:
    """Device-resident table with the five attributes of the reference's struct
    (apply_hrtf.py:36-44) plus the packed forms the kernels read."""

    def __init__(self, upsampling, diffs_left, diffs_right, irs_left, irs_right, device=None):
        import torch
        device = _hip.require_gpu(device)
        self.upsampling = int(upsampling)                                   # :38
        il = np.ascontiguousarray(irs_left, dtype=np.float32)
        ir = np.ascontiguousarray(irs_right, dtype=np.float32)
        dl = np.ascontiguousarray(diffs_left, dtype=np.float64)
        dr = np.ascontiguousarray(diffs_right, dtype=np.float64)
        if il.shape != ir.shape or il.ndim != 2 or il.shape[1] % self.upsampling:
            raise ValueError("irs_left/irs_right must be (ndir, samples_to_keep*upsampling)")
        n = il.shape[0]
        if dl.shape != (n, n) or dr.shape != (n, n):
            raise ValueError("diffs_left/diffs_right must be (ndir, ndir)")
        self.device = device
        self.ndir, self.M = il.shape
        self.L = self.M // self.upsampling
        irs = torch.from_numpy(np.stack([il, ir])).to(device)               # [2][ndir][M]
        self.irs_left, self.irs_right = irs[0], irs[1]                      # :43-44
        self.diffs = torch.from_numpy(np.stack([dl, dr])).to(device)        # [2][ndir][ndir] f64
        self.diffs_left, self.diffs_right = self.diffs[0], self.diffs[1]    # :40-41
        n_packed = _hip.lib().bas_table_packed_floats(self.ndir, self.M, self.upsampling)
        self.packed = torch.empty((n_packed,), dtype=torch.float32, device=device)   # [2][ndir][U][1+L]
        with _hip.on_device(device):
            _hip.call("bas_table_pack_f32", _hip.ptr(irs), self.ndir, self.M, self.upsampling,
                      _hip.ptr(self.packed), _hip.current_stream(device))


def load_irs_and_delaydiffs(filename='irs_and_delaydiffs_compensated_6.mat', samples_to_keep=512, device=None):
    """Load the table written by upsample_irs.m and keep the first
    samples_to_keep*upsampling columns of every IR (apply_hrtf.py:23-46)."""
    import scipy.io
    m = scipy.io.loadmat(filename)['irs_and_delaydiffs']                    # :34
    rec = m[0][0]
    upsampling = int(rec['upsampling'][0][0])                               # :38
    keep = samples_to_keep * upsampling
    return irs_and_delaydiffs(upsampling, rec['diffs_left'], rec['diffs_right'],
                              rec['irs_left'][:, :keep], rec['irs_right'][:, :keep], device=device)


def _table_key(tbl):
    """What a cached device copy of a foreign table struct was made from: the identity and shape of its five
    fields.  Re-binding a field (e.g. re-truncating irs_left) changes the key; in-place edits of the same array
    cannot be seen and need `del tbl._bas_device_tables`."""
    return (int(tbl.upsampling),) + tuple((id(a), tuple(np.shape(a))) for a in
                                          (tbl.diffs_left, tbl.diffs_right, tbl.irs_left, tbl.irs_right))


def as_device_table(tbl, device=None):
    """Accept this module's table, or any object with the reference's five attributes
    holding numpy arrays (e.g. the reference's own class-as-struct).  The device copy of a
    foreign struct is cached on the object per (device, field identities): another GPU or a
    re-bound field gets its own copy."""
    if isinstance(tbl, irs_and_delaydiffs):
        return tbl
    device = _hip.require_gpu(device)
    key = _table_key(tbl)
    cache = getattr(tbl, "_bas_device_tables", None)            # {device: (key, device table)}
    hit = cache.get(str(device)) if cache is not None else None
    if hit is not None and hit[0] == key:
        return hit[1]
    made = irs_and_delaydiffs(tbl.upsampling, tbl.diffs_left, tbl.diffs_right,
                              tbl.irs_left, tbl.irs_right, device=device)
    if cache is None:
        cache = {}
        try:
            tbl._bas_device_tables = cache
        except (AttributeError, TypeError):
            pass
    # the arrays are kept alive with the entry so that their id()s cannot be reused by later allocations
    cache[str(device)] = (key, made, (tbl.diffs_left, tbl.diffs_right, tbl.irs_left, tbl.irs_right))
    return made


# --------------------------------------------------------------------------
# a4 / a5
# --------------------------------------------------------------------------
def delay_signal_float(in_sig, samples, downsample=1):
    """Fractional circular delay by linear blend of the two adjacent integer shifts
    (apply_hrtf.py:127-165).  in_sig: 1-D array; returns a float32 numpy array."""
    import torch
    device = _hip.require_gpu()
    x = torch.as_tensor(np.ascontiguousarray(in_sig, dtype=np.float32)).reshape(1, -1).to(device)
    m = x.shape[1]
    s = torch.tensor([float(samples)], dtype=torch.float64, device=device)
    y = torch.empty((1, (m + downsample - 1) // downsample), dtype=torch.float32, device=device)
    _hip.call("bas_delay_signal_f32", _hip.ptr(x), _hip.ptr(s), 1, m, int(downsample), _hip.ptr(y),
              _hip.current_stream(device))
    return y[0].cpu().numpy()


def delay_compensated_interpolation_with_delaydiff(irs_and_delaydiffs, before: int, after: int, alpha: float,
                                                   return_upsampled=False):
    """Delay-compensated interpolation along one ring, both ears (apply_hrtf.py:53-106).
    Returns (delay_l, delay_r, irs[2, M or L])."""
    import torch
    tbl = as_device_table(irs_and_delaydiffs)
    dev = tbl.device
    if not (0 <= int(before) < tbl.ndir and 0 <= int(after) < tbl.ndir):
        raise IndexError("HRTF database index out of range")
    pq = torch.tensor([[int(before), int(after)]], dtype=torch.int32, device=dev)
    al = torch.tensor([float(alpha)], dtype=torch.float64, device=dev)
    width = tbl.M if return_upsampled else tbl.L
    out = torch.empty((1, 2, width), dtype=torch.float32, device=dev)
    delays = torch.empty((1, 2), dtype=torch.float64, device=dev)
    with _hip.on_device(dev):
        _hip.call("bas_ring_interp_f32", _hip.ptr(tbl.packed), _hip.ptr(tbl.diffs), _hip.ptr(pq), _hip.ptr(al), 1,
                  tbl.ndir, tbl.L, tbl.upsampling, int(bool(return_upsampled)), _hip.ptr(out), _hip.ptr(delays),
                  _hip.current_stream(dev))
    d = delays.cpu().numpy()
    return (d[0, 0], d[0, 1], out[0].cpu().numpy())


def delay_compensated_interpolation(irs_and_delaydiffs, before: int, after: int, alpha: float):
    """Same as above but discard the delay differences (apply_hrtf.py:108-111)."""
    return delay_compensated_interpolation_with_delaydiff(irs_and_delaydiffs, before, after, alpha)[2]


def ring_easy_params(continuous_index):
    """(before, after, alpha) of delay_compensated_interpolation_easy (apply_hrtf.py:116-122), with the
    reference's hard-wired wrap of the horizontal ring (after == 97 -> 73)."""
    before = int(np.floor(continuous_index))
    after = int(np.ceil(continuous_index))
    alpha = continuous_index - before
    if after == 97:
        after = 73
    return before, after, alpha


def delay_compensated_interpolation_easy(irs_and_delaydiffs, continuous_index: float):
    """Ring interpolation addressed by one continuous database index (apply_hrtf.py:114-125)."""
    return delay_compensated_interpolation(irs_and_delaydiffs, *ring_easy_params(continuous_index))


# --------------------------------------------------------------------------
# a6
# --------------------------------------------------------------------------
def interpolate_2d_params(tbl, idx, w, out=None, validate=True, ws=None):
    """Batched table arithmetic of interpolate_2d (apply_hrtf.py:219-279) for
    precomputed parameters: idx int32 [n,4], w float64 [n,3] (numpy or device
    tensors).  Returns a device tensor [n, 2, L] float32 (`out` if given).
    validate=False skips the index range check (a host<->device sync)."""
    tbl = as_device_table(tbl)
    with _hip.on_device(tbl.device):
        return _interpolate_2d_params(tbl, idx, w, out, validate, ws)


def _interpolate_2d_params(tbl, idx, w, out=None, validate=True, ws=None):
    import torch
    tbl = as_device_table(tbl)
    dev = tbl.device
    idx_t = torch.as_tensor(idx, dtype=torch.int32).reshape(-1, 4).contiguous().to(dev)
    w_t = torch.as_tensor(w, dtype=torch.float64).reshape(-1, 3).contiguous().to(dev)
    n = idx_t.shape[0]
    if w_t.shape[0] != n:
        raise ValueError("idx and w disagree on the number of queries")
    if validate and n and (int(idx_t.min()) < 0 or int(idx_t.max()) >= tbl.ndir):
        raise IndexError("HRTF database index out of range")
    H = out if out is not None else torch.empty((n, 2, tbl.L), dtype=torch.float32, device=dev)
    ws_bytes = _hip.lib().bas_interp2d_workspace_bytes(n)
    if ws is None or ws.numel() < ws_bytes:
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    _hip.call("bas_interp2d_f32", _hip.ptr(tbl.packed), _hip.ptr(tbl.diffs), _hip.ptr(idx_t), _hip.ptr(w_t), n,
              tbl.ndir, tbl.L, tbl.upsampling, _hip.ptr(H), _hip.ptr(ws), ws.numel(), _hip.current_stream(dev))
    return H


def interpolate_2d_batch(tbl, elev, azim):
    """interpolate_2d for float64 arrays of angles (radians); device tensor [..., 2, L]."""
    idx, w = sphere.interpolation_params_batch(elev, azim)
    H = interpolate_2d_params(tbl, idx.reshape(-1, 4), w.reshape(-1, 3))
    return H.reshape(idx.shape[:-1] + (2, H.shape[-1]))


def interpolate_2d(irs_and_delaydiffs, elev, azim):
    """HRIR pair for a source at (elev, azim) radians (apply_hrtf.py:171-281).
    Returns a (2, L) float32 numpy array.  Elevation outside [-pi/4, pi/2] is clamped,
    azimuth is wrapped, like the reference."""
    idx, w = sphere.interpolation_params(elev, azim)
    H = interpolate_2d_params(irs_and_delaydiffs, np.array([idx], dtype=np.int32), np.array([w]))
    return H[0].cpu().numpy()


def interpolate_2d_deg(irs_and_delaydiffs, elev, azim):
    deg2rad = (2 * np.pi) / 360                                             # :167-169
    return interpolate_2d(irs_and_delaydiffs, deg2rad * elev, deg2rad * azim)


# --------------------------------------------------------------------------
# a7
# --------------------------------------------------------------------------
def padded_rows(n_rows, length, device):
    """Zero-filled float32 [n_rows, length] whose rows start 16-byte aligned (row stride rounded up to a
    multiple of 4 floats): what the fast FIR kernels want when the padded signal length is not a multiple of 4."""
    import torch
    stride = (length + 3) // 4 * 4
    return torch.zeros((n_rows, stride), dtype=torch.float32, device=device)[:, :length]


def render_lengths(n, chunksize, ir_length):
    in_length = int(0.5 + math.ceil(n / chunksize) * chunksize)             # :405
    return in_length, in_length + ir_length - 1                             # :410


@_hip.on_device_of("x")
def render_device(x, chunksize, subchunksize, H, tbl_L, normalize="mix", out=None, accumulate=False,
                  events=None, ws=None):
    """Core launch: x [n_src, T_in] device float32 (T_in % K == 0), H [n_src, n_chunks+1, 2, L].
    Returns (y [2, T_out] device float32, peak device scalar).  `events` = (begin, end) raw
    hipEvent_t handles recorded around the FIR kernel (bench.py); `ws` = reusable workspace."""
    import torch
    dev = x.device
    n_src, t_in = x.shape
    t_out = t_in + tbl_L - 1
    y = out if out is not None else torch.empty((2, t_out), dtype=torch.float32, device=dev)
    peak = torch.empty((1,), dtype=torch.float32, device=dev)
    ws_bytes = _hip.lib().bas_render_workspace_bytes(n_src, t_in, chunksize, subchunksize, tbl_L)
    if ws is None or ws.numel() < ws_bytes:
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    stream = _hip.current_stream(dev)
    args = (_hip.ptr(x), x.stride(0) if n_src else t_in, _hip.ptr(H), n_src, t_in, chunksize, subchunksize, tbl_L,
            _hip.ptr(y), int(bool(accumulate)), _hip.ptr(peak), _hip.ptr(ws), ws.numel(), stream)
    if events is None:
        _hip.call("bas_render_mix_f32", *args)
    else:
        _hip.call("bas_render_mix_profiled_f32", *args, events[0], events[1])
    if normalize == "mix":
        _hip.call("bas_scale_by_peak_f32", _hip.ptr(y), 2 * t_out, _hip.ptr(peak), stream)   # :462-464
    elif normalize != "none":
        raise ValueError("normalize must be 'mix' or 'none'")
    return y, peak


@_hip.on_device_of("x")
def render_params_device(x, chunksize, subchunksize, tbl, idx, w, normalize="mix", out=None, events=None,
                         ws=None, ws_plans=None, fused=None, angles=None, n_queries=None, want_peak=True, check=False,
                         plans_ready=False):
    """interpolate_2d + render for precomputed parameters: x [n_src, T_in] device float32,
    idx int32 [n_src*(n_chunks+1), 4], w float64 [.., 3] on the device.  Uses the fused kernel
    (chunk IRs evaluated inside the FIR kernel, never stored) when the sizes allow it, else
    bas_interp2d_f32 + bas_render_mix_f32.  Returns (y [2, T_out], peak).
    normalize="mix": the peak rule (apply_hrtf.py:462-464) inside the render call (the tail of its last kernel: no launch
    of its own).  want_peak=False with normalize="none": max|y| is not computed at all (returns peak None; streams track
    their own running peak).  ws: a workspace from _hip.new_workspace (zeroed control block).  check=True: ask the
    library for device-side errors afterwards (synchronises the stream: for callers that copy the result to the host anyway).
    plans_ready=True: ws_plans already holds this scene's read plans (plan_angles_device on another stream, ordered before
    this call by the caller): only the fused FIR is launched; needs n_queries and a shape the fused kernels serve."""
    import torch
    tbl = as_device_table(tbl)
    dev = x.device
    n_src, t_in = x.shape
    lib = _hip.lib()
    if normalize not in ("mix", "none"):
        raise ValueError("normalize must be 'mix' or 'none'")
    if fused is None or fused:                                # default: fused wherever the kernel serves the shape
        fused = bool(lib.bas_render_fused_supported(n_src, t_in, chunksize, subchunksize, tbl.L)) and \
            tbl.upsampling >= 4 and x.data_ptr() % 16 == 0 and x.stride(0) % 4 == 0
    n_q = idx.shape[0] if idx is not None else n_queries
    if plans_ready and not fused:
        raise ValueError("plans_ready: this shape is not served by the fused kernels (they are the ones that read plans)")
    if not fused:
        if idx is None:                                       # (angles given, shape not served by the fused kernel)
            idx, w = sphere.interpolation_params_device(angles[0], angles[1], branch=angles[2])
            idx, w = idx.reshape(-1, 4), w.reshape(-1, 3)
        H = interpolate_2d_params(tbl, idx, w, validate=False, ws=ws_plans)
        return render_device(x, chunksize, subchunksize, H.view(n_src, n_q // max(n_src, 1), 2, tbl.L), tbl.L,
                             normalize, out=out, events=events, ws=ws)
    t_out = t_in + tbl.L - 1
    y = out if out is not None else torch.empty((2, t_out), dtype=torch.float32, device=dev)
    peak = torch.empty((1,), dtype=torch.float32, device=dev) if (want_peak or normalize == "mix") else None
    pb = lib.bas_interp2d_workspace_bytes(n_q)
    if ws_plans is None or ws_plans.numel() < pb:
        ws_plans = torch.empty((pb,), dtype=torch.uint8, device=dev)
    wb = lib.bas_render_fused_workspace_bytes(n_src, t_in, chunksize, subchunksize, tbl.L)
    if ws is None or ws.numel() < wb:
        ws = _hip.new_workspace(wb, dev)
    stream = _hip.current_stream(dev)
    if plans_ready:
        if ws_plans.numel() < pb:
            raise ValueError("plans_ready: ws_plans is smaller than bas_interp2d_workspace_bytes(n_queries)")
    elif angles is None:
        _hip.call("bas_interp2d_plan_f32", _hip.ptr(tbl.diffs), _hip.ptr(idx), _hip.ptr(w), n_q, tbl.ndir, tbl.L,
                  tbl.upsampling, _hip.ptr(ws_plans), ws_plans.numel(), stream)
    else:                                                     # small batch: a3 + plans in one launch
        e, z, branch = angles
        ring_elev, ring_start, ring_count = sphere._ring_args()
        _hip.call("bas_interp2d_plan_angles_f32", _hip.ptr(tbl.diffs), _hip.ptr(e), _hip.ptr(z), n_q, ring_elev,
                  ring_start, ring_count, _hip.ptr(sphere.device_nodes(dev)), sphere.BRANCHES[branch], tbl.ndir, tbl.L,
                  tbl.upsampling, _hip.ptr(ws_plans), ws_plans.numel(), stream)
    args = (_hip.ptr(x), x.stride(0), _hip.ptr(tbl.packed), _hip.ptr(ws_plans), n_src, t_in, chunksize, subchunksize, tbl.L,
            tbl.upsampling, tbl.ndir, _hip.ptr(y), 0, _hip.ptr(peak), int(normalize == "mix"), _hip.ptr(ws), ws.numel(), stream)
    if events is None:
        _hip.call("bas_render_mix_fused_f32", *args)          # peak rule included (:462-464)
    else:
        _hip.call("bas_render_mix_fused_profiled_f32", *args, events[0], events[1])
    if check:
        _hip.check_status(ws, dev)
    return y, peak


def plan_angles_device(tbl, elev, azim, ws_plans, branch="f64"):
    """The first launch of render_angles_device alone, on the current stream: trajectory angles (float64 device tensors,
    one per source and chunk boundary) -> a3 -> the fused kernels' read plans in ws_plans (uint8 device tensor of at least
    bas_interp2d_workspace_bytes(elev.numel()) bytes).  For callers that render a SEQUENCE of scenes and let the plans of
    the next one be computed beside the FIR of the current one (bench.py --overlap-plans; a rank's share of a multi-GPU
    scene leaves CUs free): render_params_device(..., plans_ready=True) is the other half."""
    import torch
    tbl = as_device_table(tbl)
    if not (elev.is_cuda and azim.is_cuda and elev.dtype == torch.float64 and azim.dtype == torch.float64
            and elev.is_contiguous() and azim.is_contiguous() and elev.numel() == azim.numel()):
        raise ValueError("elev/azim must be contiguous float64 device tensors of equal size")
    if branch not in sphere.BRANCHES:
        raise ValueError("branch must be 'f64' or 'pyfloat'")
    n_q, dev = elev.numel(), elev.device
    if ws_plans.numel() < _hip.lib().bas_interp2d_workspace_bytes(n_q):
        raise ValueError("ws_plans is smaller than bas_interp2d_workspace_bytes(n_queries)")
    ring_elev, ring_start, ring_count = sphere._ring_args()
    _hip.call("bas_interp2d_plan_angles_f32", _hip.ptr(tbl.diffs), _hip.ptr(elev), _hip.ptr(azim), n_q, ring_elev,
              ring_start, ring_count, _hip.ptr(sphere.device_nodes(dev)), sphere.BRANCHES[branch], tbl.ndir, tbl.L,
              tbl.upsampling, _hip.ptr(ws_plans), ws_plans.numel(), _hip.current_stream(dev))
    return ws_plans


MERGED_A3_MAX_QUERIES = 1 << 30     # a3 rides inside the plan kernel (one launch less; see bas.h) for every batch size since round 4:
                                    # the angle arithmetic is done once per query by half of a block's waves and handed over through LDS


def render_angles_device(x, chunksize, subchunksize, tbl, elev, azim, normalize="mix", out=None, events=None,
                         ws=None, ws_plans=None, fused=None, params=None, branch="f64", want_peak=True, check=False,
                         plans_ready=False):
    """The whole device side of make_signal_move_2d for trajectories that live on the GPU: x [n_src, T_in] device
    float32 (T_in % K == 0), elev / azim float64 device tensors [n_src, T_in/K + 1] (radians at t = 0, K, .., T_in).
    bas_traj_params_f64 (a3 + the elevation bracket), then render_params_device (read plans + fused FIR where the
    sizes allow it).  `params` = optional (idx [n, 4] int32, w [n, 3] float64) buffers to write the parameters
    into (no allocation per call).  Returns (y [2, T_out], peak).
    a3 runs inside the plan kernel (bas_interp2d_plan_angles_f32: one launch less, no (idx, w) round trip; round 3's form,
    in which both ears' threads redid the angle arithmetic, lost above 65 536 queries - 53 us against 9 + 15 us for 221 k;
    round 4's does it once per query: equal there, 3 us ahead for a 32-source share: tools/ab_a3_merge.py).  `params` is
    only written by the two-launch form (MERGED_A3_MAX_QUERIES = 0)."""
    tbl = as_device_table(tbl)
    n_src, t_in = x.shape
    if elev.numel() != n_src * (t_in // chunksize + 1) or azim.numel() != elev.numel():
        raise ValueError("elev/azim must hold one angle per source and chunk boundary")
    import torch
    if not (elev.is_cuda and azim.is_cuda and elev.dtype == torch.float64 and azim.dtype == torch.float64):
        raise ValueError("elev/azim must be float64 device tensors")
    if elev.numel() <= MERGED_A3_MAX_QUERIES and elev.is_contiguous() and azim.is_contiguous():
        if branch not in sphere.BRANCHES:
            raise ValueError("branch must be 'f64' or 'pyfloat'")
        return render_params_device(x, chunksize, subchunksize, tbl, None, None, normalize, out=out, events=events, ws=ws,
                                    ws_plans=ws_plans, fused=fused, angles=(elev, azim, branch), n_queries=elev.numel(),
                                    want_peak=want_peak, check=check, plans_ready=plans_ready)
    if plans_ready:
        raise ValueError("plans_ready needs contiguous angle tensors (the merged a3 + plan launch)")
    idx, w = sphere.interpolation_params_device(elev, azim, out=params, branch=branch)
    return render_params_device(x, chunksize, subchunksize, tbl, idx.reshape(-1, 4), w.reshape(-1, 3), normalize,
                                out=out, events=events, ws=ws, ws_plans=ws_plans, fused=fused, want_peak=want_peak, check=check)


def _params_to_device(tbl, idx, w):
    import torch
    idx_t = torch.as_tensor(idx, dtype=torch.int32).reshape(-1, 4).contiguous().to(tbl.device)
    w_t = torch.as_tensor(w, dtype=torch.float64).reshape(-1, 3).contiguous().to(tbl.device)
    if idx_t.shape[0] and (int(idx_t.min()) < 0 or int(idx_t.max()) >= tbl.ndir):
        raise IndexError("HRTF database index out of range")
    return idx_t, w_t


def render_sources(signals, chunksize, subchunksize, elev, azim, tbl, normalize="mix", fused=None):
    """Render and mix many independently moving sources.

    signals: [n_src, N] (numpy or tensor); elev/azim: float64 [n_src, n_chunks+1]
    radians at t = 0, K, .., in_length (what make_signal_move_2d samples, :429/:435).
    Mix semantics (not defined by the reference): sum of the un-normalised
    renders, then the reference's peak rule once on the mix ("mix") or not at all
    ("none").  Returns a device tensor of shape (out_length, 2) - a transposed view of
    [2, out_length], the same memory order the reference returns (:459).
    """
    import torch
    tbl = as_device_table(tbl)
    dev = tbl.device
    assert chunksize % subchunksize == 0, 'subchunksize does not divide chunksize evenly'
    sig = torch.as_tensor(signals)
    assert sig.dim() == 2, 'signals must be [n_src, N]'
    n_src, n = sig.shape
    in_length, _ = render_lengths(n, chunksize, tbl.L)
    x = padded_rows(n_src, in_length, dev)                                   # :406
    x[:, :n] = sig.to(device=dev, dtype=torch.float32)
    n_q = in_length // chunksize + 1
    idx, w = sphere.interpolation_params_batch(elev, azim)
    if idx.shape[:-1] != (n_src, n_q):
        raise ValueError(f"elev/azim must have shape ({n_src}, {n_q})")
    idx_t, w_t = _params_to_device(tbl, idx, w)
    y, _ = render_params_device(x, chunksize, subchunksize, tbl, idx_t, w_t, normalize, fused=fused)
    return y.t()


def trajectory_branch(elev_azim_function):
    """Which of the reference's two numeric branches a trajectory function selects (sphere.py:98-105, :119 under
    NumPy >= 2): "pyfloat" when it returns a Python float / int azimuth for a Python-int time (the reference's own
    presets circle_horizontal, circle_askew, spiral: apply_hrtf.py:585-586, :593 - comparisons and the ring weight
    then run in float32), "f64" when it returns an np.float64 (presets built on np.sin / np.arctan), None for
    anything else (np.float32 ...: only the scalar path reproduces those)."""
    azim = elev_azim_function(0)[1]                                          # the reference's first call: t = 0 (:429)
    if type(azim) is np.float64:
        return "f64"
    if type(azim) in (float, int):
        return "pyfloat"
    return None


def make_signal_move_2d(in_signal, chunksize: int, subchunksize: int, elev_azim_function, irs_and_delaydiffs,
                        verbose=False, vectorized=False, branch="auto"):
    """Make `in_signal` sound as if its source moved along elev_azim_function
    (apply_hrtf.py:356-466): chunk IRs by interpolate_2d at t = 0, K, .., in_length,
    per-subchunk linear IR crossfade, direct FIR, overlap-add, float32, peak rule.

    in_signal: 1-D numpy array (returns numpy (out_length, 2) float32, :459) or 1-D
    device tensor (returns a device tensor of that shape).
    elev_azim_function(t_samples) -> (elev, azim) in radians; it is called with the
    same scalar arguments as the reference calls it, so grid-node decisions follow the
    dtype it returns exactly as in the reference (sphere.py).  vectorized=True calls it
    ONCE with the float64 array of all chunk times instead (it must broadcast) and turns
    the angles into interpolation parameters ON THE DEVICE (bas_traj_params_branch_f64):
    ~1000x less host time on long signals.  `branch` ("f64", "pyfloat" or "auto") names
    the reference's numeric branch to reproduce there: "auto" asks the function for its
    value at t = 0 and follows the scalar type it returns (trajectory_branch), so the
    reference's own presets render exactly as the scalar path renders them; a function
    returning anything but Python floats / np.float64, or one that raises when called with
    an array, falls back to the scalar path.  The branch is chosen ONCE, from the value at
    t = 0: a function whose return type changes with t is rendered in that one branch (the
    scalar path follows it call by call).  In the "pyfloat" branch the ring weight is the
    reference's float32 value; its complement (1 - alpha, apply_hrtf.py:90) is formed from
    the widened value on the device, in float32 in the reference: up to one float32 ulp
    apart on the `before` weight (parity-unpinned detail, far below the 1e-5 bar).
    """
    import torch
    is_tensor = isinstance(in_signal, torch.Tensor)
    assert len(in_signal.shape) == 1, 'only mono signals for now'            # :398
    chunks_per_subchunk = chunksize / subchunksize
    assert chunks_per_subchunk == np.floor(chunks_per_subchunk), 'subchunksize does not divide chunksize evenly'
    tbl = as_device_table(irs_and_delaydiffs)
    dev = tbl.device
    n = int(in_signal.shape[0])
    in_length, out_length = render_lengths(n, chunksize, tbl.L)
    times = range(0, in_length + 1, chunksize)                               # :429, :435
    if vectorized and branch == "auto":
        branch = trajectory_branch(elev_azim_function)
        vectorized = branch is not None
    if vectorized:
        if branch not in sphere.BRANCHES:
            raise ValueError("branch must be 'auto', 'f64' or 'pyfloat'")
        try:
            e, a = elev_azim_function(np.arange(0, in_length + 1, chunksize, dtype=np.float64))
            e, a = np.broadcast_arrays(np.asarray(e, dtype=np.float64), np.asarray(a, dtype=np.float64))
            if e.shape != (len(times),):
                e, a = np.broadcast_to(e, (len(times),)), np.broadcast_to(a, (len(times),))
        except Exception:                                                    # a function that does not broadcast over an array
            vectorized = False                                               # (math.*, an `if` on t): the scalar path below
    if vectorized:
        if not (np.isfinite(e).all() and np.isfinite(a).all()):
            raise ValueError("trajectory contains non-finite angles")
        ea = torch.from_numpy(np.stack([e, a])).to(dev)                      # one H2D copy for both
    if not vectorized:
        idx = np.empty((len(times), 4), dtype=np.int32)
        w = np.empty((len(times), 3), dtype=np.float64)
        for i, t in enumerate(times):
            idx[i], w[i] = sphere.interpolation_params(*elev_azim_function(t))
            if verbose:
                print(' {:.1f}%           '.format(100 * t / max(in_length, 1)), end='\r')
        idx_t, w_t = _params_to_device(tbl, idx, w)
    x = padded_rows(1, in_length, dev)                                       # :405-406
    src = in_signal if is_tensor else torch.from_numpy(np.ascontiguousarray(in_signal))
    x[0, :n] = src.to(device=dev, dtype=torch.float32)
    if vectorized:
        y, _ = render_angles_device(x, int(chunksize), int(subchunksize), tbl, ea[0].reshape(1, -1), ea[1].reshape(1, -1),
                                    "mix", branch=branch, check=not is_tensor)
    else:
        y, _ = render_params_device(x, int(chunksize), int(subchunksize), tbl, idx_t.reshape(-1, 4), w_t.reshape(-1, 3), "mix",
                                    check=not is_tensor)
    if verbose:
        print(' 100.0%      ')
    out = y.t()                                                              # (out_length, 2), F-ordered like :459
    assert out.shape[0] == out_length, 'wrong output length'
    return out if is_tensor else out.cpu().numpy()


def make_signal_move(in_signal, chunksize: int, index_function, irs_and_delaydiffs, verbose=False):
    """The reference's older 1-D renderer (apply_hrtf.py:294-353): one ring-interpolated IR per chunk
    (delay_compensated_interpolation_easy at the chunk's first sample, :331), no crossfade, direct FIR,
    overlap-add, float32, peak rule.  On the device this is bas_ring_interp_f32 for all chunks at once
    followed by bas_render_mix_f32 with subchunk = chunk (crossfade weight 0 throughout, so the extra
    IR the kernel's layout wants at t = in_length is never used: the last one is repeated).
    Returns (out_length, 2) float32 (numpy for numpy input, device tensor for a device tensor)."""
    import torch
    is_tensor = isinstance(in_signal, torch.Tensor)
    assert len(in_signal.shape) == 1, 'only mono signals for now'            # :306
    tbl = as_device_table(irs_and_delaydiffs)
    dev = tbl.device
    n = int(in_signal.shape[0])
    in_length, out_length = render_lengths(n, chunksize, tbl.L)              # :309-315
    n_chunks = in_length // chunksize
    pq = np.empty((n_chunks + 1, 2), dtype=np.int32)
    al = np.empty((n_chunks + 1,), dtype=np.float64)
    for c in range(n_chunks):                                                # :328, :331
        b, a, alpha = ring_easy_params(index_function(c * chunksize))
        if not (0 <= b < tbl.ndir and 0 <= a < tbl.ndir):
            raise IndexError("HRTF database index out of range")
        pq[c], al[c] = (b, a), alpha
        if verbose and c % 64 == 0:
            print(' {:.1f}%           '.format(100 * c * chunksize / max(in_length, 1)), end='\r')
    if n_chunks:
        pq[n_chunks], al[n_chunks] = pq[n_chunks - 1], al[n_chunks - 1]
    else:
        pq[0], al[0] = (0, 0), 0.0
    pq_t = torch.from_numpy(pq).to(dev)
    al_t = torch.from_numpy(al).to(dev)
    H = torch.empty((1, n_chunks + 1, 2, tbl.L), dtype=torch.float32, device=dev)
    with _hip.on_device(dev):
        _hip.call("bas_ring_interp_f32", _hip.ptr(tbl.packed), _hip.ptr(tbl.diffs), _hip.ptr(pq_t), _hip.ptr(al_t),
                  n_chunks + 1, tbl.ndir, tbl.L, tbl.upsampling, 0, _hip.ptr(H), None, _hip.current_stream(dev))
    x = padded_rows(1, in_length, dev)                                       # :309-310
    src = in_signal if is_tensor else torch.from_numpy(np.ascontiguousarray(in_signal))
    x[0, :n] = src.to(device=dev, dtype=torch.float32)
    y, _ = render_device(x, int(chunksize), int(chunksize), H, tbl.L, "mix")
    if verbose:
        print(' 100.0%      ')
    out = y.t()
    assert out.shape[0] == out_length, 'wrong output length'                 # :315
    return out if is_tensor else out.cpu().numpy()
