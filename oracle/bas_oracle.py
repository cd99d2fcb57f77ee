"""CPU oracle: a float64 numpy restatement of the reference's moving-source render path.

TEST INFRASTRUCTURE ONLY.  Nothing under the product package imports this
module; it may be used by tests/, by __graft_entry__.smoke() and by bench.py's
`cpu_baseline` leg, always as the checker / reported baseline, never as the
thing shipped or measured as the GPU path.

Parity status: PINNED.  The reference ships no tests or fixtures (SURVEY.md
section 4), so the oracle is pinned against outputs of the reference itself,
generated in the build container by tests/golden/make_golden.py (which imports
/root/reference/apply_hrtf.py unmodified) and committed as tests/golden/*.npz.
tests/test_oracle_golden.py checks every function below against them.

Each function cites the reference lines it restates.  The arithmetic follows the
reference operation for operation (so float64 results agree to the last bit on
the goldens) but is written from the closed forms of SURVEY.md section 7, with
modular index arithmetic instead of array rotation.
"""
import math

import numpy as np

TWO_PI = 2 * np.pi
RING_ELEVS_DEG = (-45, -30, -15, 0, 15, 30, 45, 60, 75, 90)
_RING_COUNTS = (24, 24, 24, 24, 24, 24, 24, 12, 6, 1)
POLE_INDEX = 186


# --------------------------------------------------------------------------
# a2: the direction table, sphere.py:124-319 / :350
# --------------------------------------------------------------------------
def ring_table():
    """(187,3) float32 rows [index, elev_rad, azim_rad].

    Degrees are stored as float32 and scaled in place by the float64 constant
    2*pi/360, i.e. float32(deg) * float32-rounded product (sphere.py:315-318).
    """
    rows = []
    for elev, count in zip(RING_ELEVS_DEG, _RING_COUNTS):
        for i in range(count):
            rows.append((len(rows), elev, i * (360 // count)))
    t = np.array(rows, dtype=np.float32)
    t[:, 1:3] *= (2 * np.pi / 360)
    return t


_TABLE = ring_table()


# --------------------------------------------------------------------------
# a3: azimuth -> (before, a, after) on one ring, sphere.py:78-121
# --------------------------------------------------------------------------
def azim_params(elev, azim):
    """Scalar restatement of sphere.azim_to_interpolation_params.

    numpy's promotion rules are kept on purpose: a Python-float `azim` is
    compared and divided in float32, an np.float64 `azim` in float64
    ("Branch decisions").
    """
    azim = azim % TWO_PI                                   # sphere.py:86
    elev = np.clip(elev, -np.pi / 4, np.pi / 2)            # :88
    if abs(elev - np.pi / 2) < 1e-5:                       # :92-93
        return (POLE_INDEX, 0., POLE_INDEX)
    on_ring = np.abs(_TABLE[:, 1] - elev) < 1e-5           # :98
    if not on_ring.any():                                  # :100-101
        raise ValueError("elev is not one of the database elevations")
    ring_idx = np.nonzero(on_ring)[0]
    ring_az = _TABLE[ring_idx, 2]
    le = ring_az <= azim                                   # :103
    before = int(ring_idx[le].max())
    gt = ~le                                               # :105 (az > azim)
    after = int(ring_idx[gt].min()) if gt.any() else int(ring_idx[0])   # :104-109
    az_b = _TABLE[before, 2]
    az_a = _TABLE[after, 2]
    if az_a < az_b:                                        # :115-117
        az_a = TWO_PI
    a = (azim - az_b) / (az_a - az_b)                      # :119
    return (before, a, after)


# --------------------------------------------------------------------------
# a4: fractional circular shift, apply_hrtf.py:127-165
# --------------------------------------------------------------------------
def frac_shift(x, s, step=1):
    """S(x,s)[n] = (1-f) x[(n-floor s) mod M] + f x[(n-ceil s) mod M], sampled at n = 0,step,2*step..

    Decimation happens before the blend (apply_hrtf.py:160-165).
    """
    x = np.asarray(x)
    m = x.size
    lo = int(np.floor(s))                                  # :149
    hi = int(np.ceil(s))                                   # :150
    f = s - lo                                             # :151
    n = np.arange(0, m, step)
    return (1 - f) * x[(n - lo) % m] + f * x[(n - hi) % m]  # :156-165


# --------------------------------------------------------------------------
# a5: ring interpolation, apply_hrtf.py:53-106
# --------------------------------------------------------------------------
def ring_interp(tbl, before, after, alpha, return_upsampled=False):
    u = tbl.upsampling
    step = 1 if return_upsampled else u                    # :97-102
    out, delays = [], []
    for irs, diffs in ((tbl.irs_left, tbl.diffs_left), (tbl.irs_right, tbl.diffs_right)):
        d = u * diffs[before, after]                       # :82-83
        q_nodelay = frac_shift(irs[after, :], -d)          # :86-87
        blend = (1 - alpha) * irs[before, :] + alpha * q_nodelay   # :90-91
        d_i = alpha * d                                    # :94-95
        out.append(frac_shift(blend, d_i, step))           # :98-102
        delays.append(d_i / u)                             # :106
    return (delays[0], delays[1], np.vstack(out))


# --------------------------------------------------------------------------
# a6: 2-D interpolation, apply_hrtf.py:171-281
# --------------------------------------------------------------------------
_ELEVS = np.deg2rad(np.array(RING_ELEVS_DEG))              # apply_hrtf.py:199


def elev_bracket(elev):
    """(lower, higher) database elevations around `elev`, clamped (apply_hrtf.py:201-211)."""
    below = _ELEVS[_ELEVS <= elev]
    above = _ELEVS[_ELEVS >= elev]
    lower = below.max() if below.size else -0.78539816339744828
    higher = above.min() if above.size else 1.5707963267948966
    return lower, higher


def interp2d_params(elev, azim):
    """(pt, qt, alpha_t, pb, qb, alpha_b, a): everything interpolate_2d derives from the angles."""
    lower, higher = elev_bracket(elev)
    pt, at, qt = azim_params(higher, azim)                 # :214
    pb, ab, qb = azim_params(lower, azim)                  # :215
    a = (elev - lower) / (higher - lower) if higher > lower else 0   # :261-265
    return pt, qt, at, pb, qb, ab, a


def interp2d_from_params(tbl, pt, qt, at, pb, qb, ab, a):
    u = tbl.upsampling
    dlt, drt, top = ring_interp(tbl, pt, qt, at, True)     # :219
    dlb, drb, bot = ring_interp(tbl, pb, qb, ab, True)     # :220
    out = []
    for e, (diffs, dt, db) in enumerate(((tbl.diffs_left, dlt, dlb), (tbl.diffs_right, drt, drb))):
        dv = u * (-dt + diffs[pt, pb] + db)                # :246-252
        bot_nodelay = frac_shift(bot[e, :], -dv)           # :254-255
        blend = (1 - a) * bot_nodelay + a * top[e, :]      # :268-269
        out.append(frac_shift(blend, (1 - a) * dv, u))     # :272-277
    return np.vstack(out)


def interp2d(tbl, elev, azim):
    return interp2d_from_params(tbl, *interp2d_params(elev, azim))


def interp2d_deg(tbl, elev, azim):                         # apply_hrtf.py:167-169
    k = (2 * np.pi) / 360
    return interp2d(tbl, k * elev, k * azim)


# --------------------------------------------------------------------------
# a7/a8: time-varying convolution with overlap-add, apply_hrtf.py:356-466
# --------------------------------------------------------------------------
def ir_length(tbl):
    return int(0.5 + tbl.irs_left.shape[1] / tbl.upsampling)   # :399


def render_lengths(n, chunksize, l):
    in_length = int(0.5 + math.ceil(n / chunksize) * chunksize)   # :405
    return in_length, in_length + l - 1                           # :410


def chunk_irs(tbl, chunksize, in_length, traj):
    """IRs at t = 0, K, .., in_length  ->  (n_chunks+1, 2, L)  (apply_hrtf.py:429, :435)."""
    return np.stack([interp2d(tbl, *traj(t)) for t in range(0, in_length + 1, chunksize)])


def render_from_irs(in_signal, chunksize, subchunksize, irs, normalize=True):
    """The reference's loop structure: per-subchunk crossfaded IR, direct FIR, overlap-add.

    `irs` is (n_chunks+1, 2, L) float64.  This is also the `cpu_baseline`
    workload of bench.py ("port" of apply_hrtf.py:431-464, one core).
    """
    x = np.asarray(in_signal)
    assert x.ndim == 1                                     # :398
    assert chunksize % subchunksize == 0                   # :401-402
    l = irs.shape[2]
    in_length, out_length = render_lengths(x.size, chunksize, l)
    x = np.concatenate([x.astype(np.float64), np.zeros(in_length - x.size)])   # :406
    out = np.zeros((2, out_length))                        # :413-414
    for c, i in enumerate(range(0, in_length, chunksize)): # :431
        h0, h1 = irs[c], irs[c + 1]                        # :434-435
        for j in range(0, chunksize, subchunksize):        # :438
            alpha = j / chunksize                          # :442
            h = (1 - alpha) * h0 + alpha * h1              # :443
            seg = x[i + j: i + j + subchunksize]
            lo = i + j
            out[0, lo: lo + subchunksize + l - 1] += np.convolve(seg, h[0])   # :445, :452
            out[1, lo: lo + subchunksize + l - 1] += np.convolve(seg, h[1])   # :446, :453
    out = out.astype(np.float32).T                         # :459  (out_length, 2)
    if normalize:
        out = peak_normalize(out)
    return out


def peak_normalize(out):
    m = np.max([out.max(), -(out.min())]) if out.size else 0.0   # :462
    if m > 1:                                              # :463-464
        out /= m
    return out


def render(in_signal, chunksize, subchunksize, traj, tbl):
    """Restatement of make_signal_move_2d (apply_hrtf.py:356-466)."""
    x = np.asarray(in_signal)
    assert x.ndim == 1
    in_length, _ = render_lengths(x.size, chunksize, ir_length(tbl))
    irs = chunk_irs(tbl, chunksize, in_length, traj)
    return render_from_irs(x, chunksize, subchunksize, irs)


def render_mix(signals, chunksize, subchunksize, irs_per_source, normalize=True):
    """Multi-source semantics of this build (not in the reference):
    mix = sum over sources of the un-normalised float64 renders, cast to float32,
    then the reference's peak rule applied once to the mix."""
    acc = None
    for x, irs in zip(signals, irs_per_source):
        x = np.asarray(x)
        l = irs.shape[2]
        in_length, out_length = render_lengths(x.size, chunksize, l)
        xp = np.concatenate([x.astype(np.float64), np.zeros(in_length - x.size)])
        out = np.zeros((2, out_length))
        for c, i in enumerate(range(0, in_length, chunksize)):
            for j in range(0, chunksize, subchunksize):
                alpha = j / chunksize
                h = (1 - alpha) * irs[c] + alpha * irs[c + 1]
                lo = i + j
                seg = xp[lo: lo + subchunksize]
                out[0, lo: lo + subchunksize + l - 1] += np.convolve(seg, h[0])
                out[1, lo: lo + subchunksize + l - 1] += np.convolve(seg, h[1])
        acc = out if acc is None else acc + out
    res = acc.astype(np.float32).T
    return peak_normalize(res) if normalize else res


# --------------------------------------------------------------------------
# legacy 1-D path: apply_hrtf.py:108-125, :294-353
# --------------------------------------------------------------------------
def ring_interp_irs(tbl, before, after, alpha):            # apply_hrtf.py:108-111
    return ring_interp(tbl, before, after, alpha)[2]


def ring_easy_params(continuous_index):
    """(before, after, alpha) of delay_compensated_interpolation_easy (apply_hrtf.py:114-125),
    including its hard-wired wrap of the horizontal ring (after 97 -> 73, :121-122)."""
    before = int(np.floor(continuous_index))               # :116
    after = int(np.ceil(continuous_index))                 # :117
    alpha = continuous_index - before                      # :118
    if after == 97:                                        # :121-122
        after = 73
    return before, after, alpha


def ring_easy(tbl, continuous_index):
    return ring_interp_irs(tbl, *ring_easy_params(continuous_index))


def render_1d(in_signal, chunksize, index_function, tbl, normalize=True):
    """Restatement of make_signal_move (apply_hrtf.py:294-353): ONE ring-interpolated IR per chunk
    (taken at the chunk's first sample, :331), no crossfade, direct FIR of the whole chunk, overlap-add."""
    x = np.asarray(in_signal)
    assert x.ndim == 1                                     # :306
    l = ir_length(tbl)                                     # :307
    in_length, out_length = render_lengths(x.size, chunksize, l)   # :309-315
    x = np.concatenate([x.astype(np.float64), np.zeros(in_length - x.size)])
    out = np.zeros((2, out_length))
    for i in range(0, in_length, chunksize):               # :328
        h = ring_easy(tbl, index_function(i))              # :331
        out[0, i: i + chunksize + l - 1] += np.convolve(x[i: i + chunksize], h[0])   # :334, :339
        out[1, i: i + chunksize + l - 1] += np.convolve(x[i: i + chunksize], h[1])   # :335, :340
    out = out.astype(np.float32).T                         # :347
    return peak_normalize(out) if normalize else out       # :349-351


def render_window(x_win, m_first, chunksize, subchunksize, ir_of, l, n0, n1):
    """Output samples [n0, n1) of ONE source's un-normalised render, straight from the definition of
    apply_hrtf.py:431-453 (float64): input sample m lies in chunk c = m // K, subchunk start
    j = ((m % K) // S) * S, and is filtered with g = (1 - j/K) H_c + (j/K) H_{c+1} (:442-446).
    x_win holds the input samples m_first .. m_first + len(x_win) - 1 (must cover max(n0-l+1, 0) .. n1-1
    as far as the signal exists); ir_of(c) -> (2, l) chunk IR.  Used for spot checks at sizes where the
    whole-signal restatement is too slow."""
    out = np.zeros((2, n1 - n0))
    x_win = np.asarray(x_win, dtype=np.float64)
    g_cache = {}
    for m in range(max(n0 - l + 1, m_first, 0), min(n1, m_first + x_win.size)):
        c = m // chunksize
        j = ((m % chunksize) // subchunksize) * subchunksize
        if (c, j) not in g_cache:
            alpha = j / chunksize
            g_cache[(c, j)] = (1 - alpha) * ir_of(c) + alpha * ir_of(c + 1)
        g = g_cache[(c, j)]
        lo, hi = max(n0, m), min(n1, m + l)                 # outputs this input sample reaches
        if hi > lo:
            out[:, lo - n0:hi - n0] += x_win[m - m_first] * g[:, lo - m:hi - m]
    return out
