"""Offline table builder: the struct `load_irs_and_delaydiffs` reads.

Restates upsample_irs.m of the reference (Octave): for every pair of directions the delay difference
of the two HRIRs (cross-correlation, band-limited x U upsampling, arg-max refined by a parabola,
upsample_irs.m:59-101), antisymmetrised (:31-32), and every HRIR resampled x U (:37-44); saved as the
MATLAB-v5 struct `irs_and_delaydiffs` with the five fields apply_hrtf.py:38-44 indexes.

PARITY UNPINNED.  Octave is not available in the build container and the IRCAM LISTEN data
(upsample_irs.m:1) cannot be fetched, so nothing here is checked against the reference's output.  The
band-limited resampler is scipy.signal.resample_poly with a Kaiser (beta = 5) window, the design family
Octave's `resample` uses; its coefficients are not guaranteed to match Octave's.  What IS tested
(tests/test_upsample_irs.py): antisymmetry and zero diagonal of the delay matrices, exact known answers
for shifted impulses, and that the written file loads through load_irs_and_delaydiffs's indexing.

This is an offline, run-once precompute on the host (17 391 pairs x 2 ears of 512-tap correlations, batched
per row of the pair matrix through FFTs); it is not part of the render path and has no GPU kernel.
"""
import numpy as np


def parabolic_interpolation(vec):
    """Abscissa of the vertex of the parabola through (-1, vec[0]), (0, vec[1]), (1, vec[2])
    (upsample_irs.m:88-101).  vec[1] must be the maximum."""
    vec = np.asarray(vec, dtype=np.float64)
    assert vec.shape == (3,)
    assert int(np.argmax(vec)) == 1
    c = vec[1]
    a = 0.5 * (vec[0] + vec[2] - 2 * c)
    b = 0.5 * (vec[2] - vec[0])
    assert a != 0
    return -b / (2 * a)


def _resample(x, upsampling):
    import scipy.signal
    return scipy.signal.resample_poly(np.asarray(x, dtype=np.float64), upsampling, 1, window=("kaiser", 5.0))


def delaydifference(signal_a, signal_b, upsampling):
    """Delay of b relative to a in (non-upsampled) samples, > 0 if b comes after a (upsample_irs.m:58-77)."""
    a = np.asarray(signal_a, dtype=np.float64).ravel()
    b = np.asarray(signal_b, dtype=np.float64).ravel()
    assert a.size == b.size
    n = a.size
    xc = np.convolve(a[::-1], b)                             # cross-correlation, length 2n-1 (:66)
    xc_up = _resample(xc, upsampling)
    k = int(np.argmax(xc_up))                                # 0-based peak (:69)
    k = min(max(k, 1), xc_up.size - 2)
    peak = k + parabolic_interpolation(xc_up[k - 1:k + 2])   # (:70), 0-based
    return peak / upsampling - (n - 1)                       # (:73-76) in 0-based indexing


def delaydifferences_from(h, i, upsampling):
    """delaydifference(h[i], h[j]) for every j > i at once (row i of the upper triangle, upsample_irs.m:22-28):
    one batched FFT cross-correlation (the reference uses fftconv too, :66), one polyphase resampling along the
    last axis, vectorised arg-max and parabola.  Same arithmetic per pair as `delaydifference` up to the FFT's
    rounding (tested equal to 1e-9 samples)."""
    import scipy.signal
    h = np.asarray(h, dtype=np.float64)
    n = h.shape[1]
    others = h[i + 1:]
    if others.shape[0] == 0:
        return np.zeros((0,))
    xc = scipy.signal.fftconvolve(np.broadcast_to(h[i, ::-1], others.shape), others, axes=1)   # (m, 2n-1)
    xc_up = scipy.signal.resample_poly(xc, upsampling, 1, axis=1, window=("kaiser", 5.0))
    k = np.clip(np.argmax(xc_up, axis=1), 1, xc_up.shape[1] - 2)
    rows = np.arange(xc_up.shape[0])
    lo, mid, hi = xc_up[rows, k - 1], xc_up[rows, k], xc_up[rows, k + 1]
    # the reference's preconditions (upsample_irs.m:90-98): the middle sample is the maximum, the parabola is not flat
    assert np.all(mid >= lo) and np.all(mid >= hi), "cross-correlation peak at the edge of its support"
    a = 0.5 * (lo + hi - 2 * mid)
    b = 0.5 * (hi - lo)
    assert np.all(a != 0), "three collinear points around a cross-correlation peak"
    peak = k - b / (2 * a)
    return peak / upsampling - (n - 1)


def upsample_irs(hrirs_left, hrirs_right, upsampling=8, progress=None):
    """hrirs_*: (n_dir, n_taps) arrays (the `content_m` matrices of the IRCAM structs).  Returns a dict
    with the five fields of the reference's struct (upsample_irs.m:46-51).  187 x 512 taps, U = 8: about
    half a minute on one host core (the per-pair Python loop of round 1 needed ~20 minutes)."""
    import scipy.signal
    hl = np.asarray(hrirs_left, dtype=np.float64)
    hr = np.asarray(hrirs_right, dtype=np.float64)
    assert hl.shape == hr.shape and hl.ndim == 2
    n_dir, n_taps = hl.shape
    dl = np.zeros((n_dir, n_dir))
    dr = np.zeros((n_dir, n_dir))
    for i in range(n_dir):                                   # upper triangle (:22-28), a row of pairs at a time
        dl[i, i + 1:] = delaydifferences_from(hl, i, upsampling)
        dr[i, i + 1:] = delaydifferences_from(hr, i, upsampling)
        if progress:
            progress(i, n_dir)
    dl = dl - dl.T                                           # antisymmetry (:31-32)
    dr = dr - dr.T
    irs_left = scipy.signal.resample_poly(hl, upsampling, 1, axis=1, window=("kaiser", 5.0))     # (:37-44)
    irs_right = scipy.signal.resample_poly(hr, upsampling, 1, axis=1, window=("kaiser", 5.0))
    assert irs_left.shape == (n_dir, n_taps * upsampling)
    return {"upsampling": float(upsampling), "diffs_left": dl, "diffs_right": dr,
            "irs_left": irs_left, "irs_right": irs_right}


def save(path, table):
    """Write the struct the way `save -6` does for the reference (upsample_irs.m:53): MATLAB v5."""
    import scipy.io
    scipy.io.savemat(path, {"irs_and_delaydiffs": table}, format="5")
