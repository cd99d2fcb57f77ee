"""Sphere geometry of the 187-direction HRIR grid and the angle -> (indices, weights) step.

Mirrors the part of the reference's sphere.py that is on the render path:
the module constant `index_elev_azim` (sphere.py:124-319, :350) and
`azim_to_interpolation_params` (sphere.py:78-121), plus a vectorised float64 form
that also does interpolate_2d's elevation bracket (apply_hrtf.py:199-215, :261-266)
for whole trajectories at once.

Branch decisions at grid nodes are parity critical (the interpolation jumps by
~1e-2 of the IR peak across a node): node angles are float32
values float32(deg) * float32(2*pi/360) exactly as the reference builds them, and
the scalar function keeps numpy's promotion rules (a Python-float azimuth is
handled in float32, an np.float64 azimuth in float64).  The vectorised form
implements the np.float64 branch.
"""
import bisect

import numpy as np

RING_ELEVS_DEG = (-45, -30, -15, 0, 15, 30, 45, 60, 75, 90)
RING_COUNTS = (24, 24, 24, 24, 24, 24, 24, 12, 6, 1)
RING_START = tuple(int(v) for v in np.concatenate([[0], np.cumsum(RING_COUNTS)[:-1]]))
N_DIRECTIONS = 187
POLE = 186
_TOL = 0.00001                                      # sphere.py:90


def get_index_elev_azim():
    """(187,3) float32 [index, elev_rad, azim_rad] (sphere.py:124-319)."""
    t = np.empty((N_DIRECTIONS, 3), dtype=np.float32)
    row = 0
    for elev, count in zip(RING_ELEVS_DEG, RING_COUNTS):
        step = 360 // count
        for i in range(count):
            t[row] = (row, elev, i * step)
            row += 1
    t[:, 1:3] *= (2 * np.pi / 360)                  # in-place float32 scaling, sphere.py:318
    return t


index_elev_azim = get_index_elev_azim()             # sphere.py:350
_RING_ELEV32 = np.array([index_elev_azim[s, 1] for s in RING_START], dtype=np.float32)
_AVAILABLE_ELEVS = np.deg2rad(np.array(RING_ELEVS_DEG))          # apply_hrtf.py:199 (float64)


def azim_to_interpolation_params(elev, azim):
    """(before, a, after) on the database ring at `elev` (sphere.py:78-121).

    `elev` must be a database elevation (ValueError otherwise, sphere.py:100-101);
    out-of-range elevations are clamped (:88), azimuth is wrapped (:86).
    """
    azim = azim % (2 * np.pi)
    assert azim >= 0
    elev = np.clip(elev, -np.pi / 4, np.pi / 2)
    if abs(elev - np.pi / 2) < _TOL:
        return (POLE, 0., POLE)
    hit = np.nonzero(np.abs(_RING_ELEV32 - elev) < _TOL)[0]
    if hit.size == 0:
        raise ValueError('ele must be one of the values in the database: '
                         '[-45,-30,-15,0,15,30,45,60,75,90] .* (2pi / 360)')
    start, count = RING_START[hit[0]], RING_COUNTS[hit[0]]
    nodes = index_elev_azim[start:start + count, 2]
    n_le = int(np.count_nonzero(nodes <= azim))     # nodes ascend, node 0 is azimuth 0
    before = start + n_le - 1
    after = start + n_le if n_le < count else start
    before_azim = index_elev_azim[before, 2]
    after_azim = index_elev_azim[after, 2]
    if after_azim < before_azim:
        after_azim = 2 * np.pi
    a = (azim - before_azim) / (after_azim - before_azim)
    return (before, a, after)


def elevation_bracket(elev):
    """(lower, higher) database elevations around elev, clamped (apply_hrtf.py:201-211)."""
    le = _AVAILABLE_ELEVS[_AVAILABLE_ELEVS <= elev]
    ge = _AVAILABLE_ELEVS[_AVAILABLE_ELEVS >= elev]
    lower = le.max() if le.size else -0.78539816339744828
    higher = ge.min() if ge.size else 1.5707963267948966
    assert higher >= lower, "something's messed up"
    return lower, higher


def _interpolation_params_numpy(elev, azim):
    """interpolation_params with the reference's own numpy expressions (any scalar types)."""
    lower, higher = elevation_bracket(elev)
    tb, ta, taf = azim_to_interpolation_params(higher, azim)      # apply_hrtf.py:214
    bb, ba, baf = azim_to_interpolation_params(lower, azim)       # :215
    if higher > lower:                                            # :261-266
        a = (elev - lower) / (higher - lower)
    else:
        a = 0
    assert 0 <= a <= 1, 'interpolation parameter somehow takes invalid value'
    return (tb, taf, bb, baf), (float(ta), float(ba), float(a))


_TWO_PI = 2 * np.pi
_AVAIL_LIST = [float(v) for v in _AVAILABLE_ELEVS]
_NODE_LISTS = [[float(v) for v in index_elev_azim[s:s + c, 2]] for s, c in zip(RING_START, RING_COUNTS)]   # f64(f32 node)


def _ring_denominators(start, count):
    """after_azim - before_azim exactly as sphere.py:119 evaluates it: both are float32 table entries (or the
    Python float 2*pi against a float32 entry, which numpy 2 also evaluates in float32), so the difference is
    rounded to float32 before the float64 division."""
    den = []
    for i in range(count):
        before_azim = index_elev_azim[start + i, 2]
        after_azim = index_elev_azim[start + i + 1, 2] if i + 1 < count else 2 * np.pi
        den.append(float(after_azim - before_azim))
    return den


_DEN_LISTS = [_ring_denominators(s, c) for s, c in zip(RING_START, RING_COUNTS)]


def _ring_params_f64(ring, azim):
    """azim_to_interpolation_params on ring number `ring` for a float64 azimuth already wrapped into
    [0, 2 pi): the np.float64 branch of sphere.py:103-119 in plain Python floats (IEEE binary64 either way;
    float32 nodes compare and subtract as their exact binary64 values, as numpy promotes them)."""
    if ring == 9:
        return POLE, 0., POLE                                    # sphere.py:92-93
    nodes = _NODE_LISTS[ring]
    n_le = bisect.bisect_right(nodes, azim)                      # nodes ascend, node 0 is azimuth 0
    start = RING_START[ring]
    after = start + n_le if n_le < len(nodes) else start         # sphere.py:104-117
    return start + n_le - 1, (azim - nodes[n_le - 1]) / _DEN_LISTS[ring][n_le - 1], after


def interpolation_params(elev, azim):
    """Scalar form of everything interpolate_2d derives from the angles:
    (top_before, top_after, bot_before, bot_after), (top_alpha, bot_alpha, a).

    An np.float64 azimuth (what numpy-based trajectory functions return) takes a plain-Python restatement of
    the reference's float64 branch - same IEEE operations, ~10x less interpreter time per chunk; anything
    else (a Python float azimuth computes in float32 under numpy 2) goes through the
    reference's own numpy expressions."""
    if type(azim) is not np.float64 or not (elev == elev):
        return _interpolation_params_numpy(elev, azim)
    azim = float(azim) % _TWO_PI                                  # sphere.py:86
    assert azim >= 0
    e = float(elev)
    lo = bisect.bisect_right(_AVAIL_LIST, e) - 1                  # apply_hrtf.py:201-211 (clamped)
    hi = bisect.bisect_left(_AVAIL_LIST, e)
    lo = 0 if lo < 0 else lo
    hi = 9 if hi > 9 else hi
    tb, ta, taf = _ring_params_f64(hi, azim)
    bb, ba, baf = _ring_params_f64(lo, azim)
    lower, higher = _AVAIL_LIST[lo], _AVAIL_LIST[hi]
    a = (e - lower) / (higher - lower) if higher > lower else 0.0   # :261-266
    assert 0 <= a <= 1, 'interpolation parameter somehow takes invalid value'
    return (tb, taf, bb, baf), (float(ta), float(ba), a)


# ---------------------------------------------------------------------------
# vectorised float64 form (np.float64 branch), for whole trajectories
# ---------------------------------------------------------------------------
_NODES64 = [index_elev_azim[s:s + c, 2].astype(np.float64) for s, c in zip(RING_START, RING_COUNTS)]
_NODES32 = [index_elev_azim[s:s + c, 2] for s, c in zip(RING_START, RING_COUNTS)]


BRANCHES = {"f64": 0, "pyfloat": 1}                 # BAS_BRANCH_F64 / BAS_BRANCH_PYFLOAT of include/bas.h


def _ring_params_batch(ring, azim_mod, branch="f64"):
    """before, after (int32) and a (float64) for ring indices `ring` (0..9) and wrapped azimuths.
    branch "pyfloat": the reference's arithmetic for a Python-float azimuth under NumPy >= 2 (sphere.py:98-105,
    :119 with a weak scalar): azimuth rounded to float32, float32 comparisons, float32 weight."""
    n = azim_mod.shape[0]
    before = np.empty(n, dtype=np.int32)
    after = np.empty(n, dtype=np.int32)
    a = np.empty(n, dtype=np.float64)
    for r in range(10):
        sel = np.nonzero(ring == r)[0]
        if sel.size == 0:
            continue
        if r == 9:                                  # pole: sphere.py:92-93
            before[sel] = POLE
            after[sel] = POLE
            a[sel] = 0.0
            continue
        az = azim_mod[sel]
        start, count = RING_START[r], RING_COUNTS[r]
        if branch == "pyfloat":
            az = az.astype(np.float32)              # the weak Python scalar takes the array's dtype
            j = np.searchsorted(_NODES32[r], az, side="right") - 1  # last node <= azim (float32 compare)
        else:
            j = np.searchsorted(_NODES64[r], az, side="right") - 1  # last node <= azim (float64 compare)
        wrap = j + 1 >= count
        ja = np.where(wrap, 0, j + 1)
        b32 = _NODES32[r][j]
        a32 = np.where(wrap, np.float32(2 * np.pi), _NODES32[r][ja]).astype(np.float32)
        den32 = a32 - b32                           # float32 subtraction, as sphere.py:119 evaluates it
        before[sel] = start + j
        after[sel] = start + ja
        if branch == "pyfloat":
            a[sel] = ((az - b32) / den32).astype(np.float64)        # float32 throughout
        else:
            a[sel] = (az - b32.astype(np.float64)) / den32.astype(np.float64)
    return before, a, after


def interpolation_params_batch(elev, azim, branch="f64"):
    """Vectorised interpolation_params for float64 arrays of any (equal) shape.

    Returns idx int32 [..., 4] = (top_before, top_after, bot_before, bot_after) and
    w float64 [..., 3] = (top_alpha, bot_alpha, a): the inputs of bas_interp2d_f32.
    branch: "f64" = what the reference computes when its trajectory function returns np.float64 azimuths,
    "pyfloat" = when it returns Python floats (the reference's own presets; see _ring_params_batch).
    """
    if branch not in BRANCHES:
        raise ValueError("branch must be 'f64' or 'pyfloat'")
    elev = np.asarray(elev, dtype=np.float64)
    azim = np.asarray(azim, dtype=np.float64)
    elev, azim = np.broadcast_arrays(elev, azim)
    shape = elev.shape
    e = elev.reshape(-1)
    z = azim.reshape(-1) % (2 * np.pi)
    if not (np.isfinite(e).all() and np.isfinite(z).all()):
        raise ValueError("trajectory contains non-finite angles")
    hi = np.searchsorted(_AVAILABLE_ELEVS, e, side="left")        # first elevation >= elev
    lo = np.searchsorted(_AVAILABLE_ELEVS, e, side="right") - 1   # last elevation <= elev
    hi = np.minimum(hi, 9)                                        # above +90: clamp (apply_hrtf.py:206-209)
    lo = np.maximum(lo, 0)                                        # below -45: clamp (:201-204)
    lower, higher = _AVAILABLE_ELEVS[lo], _AVAILABLE_ELEVS[hi]
    tb, ta, taf = _ring_params_batch(hi, z, branch)
    bb, ba, baf = _ring_params_batch(lo, z, branch)
    span = higher - lower
    a = np.where(span > 0, (e - lower) / np.where(span > 0, span, 1.0), 0.0)
    if ((a < 0) | (a > 1)).any():
        raise AssertionError('interpolation parameter somehow takes invalid value')
    idx = np.stack([tb, taf, bb, baf], axis=-1).astype(np.int32).reshape(shape + (4,))
    w = np.stack([ta, ba, a], axis=-1).reshape(shape + (3,))
    return idx, w


_DEVICE_NODES = {}
_RING_ARGS = None


def _ring_args():
    """The ten rings as host ctypes arrays (built once): what bas_traj_params_f64 takes by value."""
    global _RING_ARGS
    if _RING_ARGS is None:
        import ctypes
        ring_elev = (ctypes.c_double * 10)(*[float(v) for v in _AVAILABLE_ELEVS])
        ring_start = (ctypes.c_int32 * 10)(*RING_START)
        ring_count = (ctypes.c_int32 * 10)(*RING_COUNTS)
        _RING_ARGS = (ring_elev, ring_start, ring_count,
                      ctypes.addressof(ring_elev), ctypes.addressof(ring_start), ctypes.addressof(ring_count))
    return _RING_ARGS[3:]


def device_nodes(dev):
    """The table's 187 node azimuths (float32, sphere.py:318) on `dev`, uploaded once per device."""
    import torch
    nodes = _DEVICE_NODES.get(dev.index)
    if nodes is None:
        nodes = _DEVICE_NODES[dev.index] = torch.from_numpy(index_elev_azim[:, 2].copy()).to(dev)
    return nodes


def interpolation_params_device(elev, azim, out=None, branch="f64"):
    """interpolation_params_batch on the GPU (bas_traj_params_branch_f64): elev/azim are float64 device
    tensors of equal shape; returns device tensors idx int32 [..., 4], w float64 [..., 3] (written into
    `out` = (idx, w) when given: contiguous tensors of those shapes, no allocation per call).
    Non-finite angles are not diagnosed here (the host form raises ValueError)."""
    if branch not in BRANCHES:
        raise ValueError("branch must be 'f64' or 'pyfloat'")
    import torch
    from . import _hip
    assert elev.is_cuda and elev.dtype == torch.float64 and azim.shape == elev.shape and azim.dtype == torch.float64
    dev = elev.device
    nodes = device_nodes(dev)
    e = elev.contiguous().reshape(-1)
    z = azim.contiguous().reshape(-1)
    n = e.numel()
    if out is None:
        idx = torch.empty((n, 4), dtype=torch.int32, device=dev)
        w = torch.empty((n, 3), dtype=torch.float64, device=dev)
    else:
        idx, w = out
        assert idx.is_contiguous() and w.is_contiguous() and idx.numel() == 4 * n and w.numel() == 3 * n
        assert idx.dtype == torch.int32 and w.dtype == torch.float64 and idx.device == dev and w.device == dev
    ring_elev, ring_start, ring_count = _ring_args()
    with _hip.on_device(dev):
        _hip.call("bas_traj_params_branch_f64", _hip.ptr(e), _hip.ptr(z), n, ring_elev, ring_start, ring_count,
                  _hip.ptr(nodes), _hip.ptr(idx), _hip.ptr(w), BRANCHES[branch], _hip.current_stream(dev))
    return idx.reshape(tuple(elev.shape) + (4,)), w.reshape(tuple(elev.shape) + (3,))
