"""Offline table builder: the struct `load_irs_and_delaydiffs` reads.

Restates upsample_irs.m of the reference (Octave): for every pair of directions the delay difference
of the two HRIRs (cross-correlation, band-limited x U upsampling, arg-max refined by a parabola,
upsample_irs.m:59-101), antisymmetrised (:31-32), and every HRIR resampled x U (:37-44); saved as the
MATLAB-v5 struct `irs_and_delaydiffs` with the five fields apply_hrtf.py:38-44 indexes.

PARITY UNPINNED.  Octave is not available in the build container and the IRCAM LISTEN data
(upsample_irs.m:1) cannot be fetched, so nothing here is checked against the reference's output.  The
band-limited resampler `octave_resample` restates the algorithm of `resample` in the Octave Forge signal
package (the function upsample_irs.m:37-44, :66 calls; resample.m by E. Chassande-Mottin) step by step -
filter design, zero padding, upfirdn, group-delay trim - from the published algorithm as the builder
remembers it: the package is not installed here and cannot be fetched, so neither its source lines nor its
output could be compared (MATLAB's `resample` is a different design: firls with Kaiser beta 5 and
half-length 10 max(p, q)).  What IS tested (tests/test_upsample_irs.py): known answers of the resampler
(length, exact interpolation of the input samples, DC gain, a band-limited sinusoid, the filter's published
parameters), antisymmetry and zero diagonal of the delay matrices, exact known answers for shifted
impulses, and that the written file loads through load_irs_and_delaydiffs's indexing.

Two forms.  `upsample_irs` is the host restatement (numpy; 17 391 pairs x 2 ears of 512-tap correlations, batched per
row of the pair matrix through FFTs; ~10 s on one core) - the definition the tests hold the device form against.
`upsample_irs_device` runs the same two steps as HIP kernels (csrc/bas_table.hip: bas_delaydiffs_f64, bas_resample_up_f64;
float64, direct sums; the whole 187 x 512-tap table in tens of milliseconds) and needs the GPU library like every other
product path - no fallback.  Neither is part of the render path.
"""
import numpy as np


def parabolic_interpolation(vec):
    """Abscissa of the vertex of the parabola through (-1, vec[0]), (0, vec[1]), (1, vec[2])
    (upsample_irs.m:88-101).  vec[1] must be the maximum as Octave's `max` finds it (:92-93: the FIRST
    largest element, so a tie with vec[0] is rejected); ValueError otherwise (the reference asserts)."""
    vec = np.asarray(vec, dtype=np.float64)
    if vec.shape != (3,):
        raise ValueError("parabolic_interpolation: need three points")           # :91
    if int(np.argmax(vec)) != 1:
        raise ValueError("parabolic_interpolation: the middle point is not the (first) maximum")   # :92-93
    c = vec[1]
    a = 0.5 * (vec[0] + vec[2] - 2 * c)
    b = 0.5 * (vec[2] - vec[0])
    if a == 0:
        raise ValueError("parabolic_interpolation: three collinear points")      # :98
    return -b / (2 * a)


def octave_resample_filter(p, q):
    """The anti-aliasing / interpolation filter Octave's signal-package `resample(x, p, q)` designs when none is
    given: a Kaiser-windowed ideal low-pass.
        stop-band rejection 60 dB (log10_rejection = -3)       cutoff f_c = 1 / (2 max(p, q)) cycles per sample
        roll-off width f_c / 10                                  half-length L = ceil((60 - 8) / (28.714 * width))
        ideal filter  2 p f_c sinc(2 f_c t), t = -L .. L         Kaiser beta = 0.1102 (60 - 8.7) = 5.653  (rejection > 50 dB)
    (the two empirical formulas are Proakis & Manolakis, Digital Signal Processing, eqs. 7.62 / 7.63, which the
    package cites).  p = 8, q = 1: L = 290, 581 taps.  Returns (h, L)."""
    p, q = int(p), int(q)
    g = int(np.gcd(p, q))
    p, q = p // g, q // g
    rejection_db = 60.0
    fc = 1.0 / (2.0 * max(p, q))
    width = fc / 10.0
    L = int(np.ceil((rejection_db - 8.0) / (28.714 * width)))
    t = np.arange(-L, L + 1, dtype=np.float64)
    ideal = 2.0 * p * fc * np.sinc(2.0 * fc * t)
    beta = 0.1102 * (rejection_db - 8.7)
    return np.kaiser(2 * L + 1, beta) * ideal, L


def octave_resample(x, p, q=1, axis=-1):
    """`resample(x, p, q)` of the Octave signal package along `axis`: zero-stuff by p, filter with
    octave_resample_filter, keep every q-th sample, and trim the filter's group delay so that output sample j sits at
    input time j q / p.  Output length ceil(Lx p / q) exactly (upsample_irs.m:37-44 relies on 512 U).
    Steps as the package performs them: pad the filter in front with nz_pre = floor(q - mod(L, q)) zeros, offset =
    floor((L + nz_pre) / q), y = upfirdn(x, h_padded, p, q)[offset : offset + Ly] (samples the filter tail would
    supply beyond the data are zeros)."""
    import scipy.signal
    x = np.moveaxis(np.asarray(x, dtype=np.float64), axis, -1)
    p, q = int(p), int(q)
    g = int(np.gcd(p, q))
    p, q = p // g, q // g
    h, L = octave_resample_filter(p, q)
    lx = x.shape[-1]
    ly = int(np.ceil(lx * p / q))
    nz_pre = int(np.floor(q - np.mod(L, q)))
    hpad = np.concatenate([np.zeros(nz_pre), h])
    offset = int(np.floor((L + nz_pre) / q))
    full = scipy.signal.upfirdn(hpad, x, up=p, down=q, axis=-1)
    if full.shape[-1] < offset + ly:                          # (the package extends the filter with zeros instead)
        full = np.concatenate([full, np.zeros(full.shape[:-1] + (offset + ly - full.shape[-1],))], axis=-1)
    return np.moveaxis(full[..., offset:offset + ly], -1, axis)


def _resample(x, upsampling, axis=-1):
    return octave_resample(x, upsampling, 1, axis=axis)


def delaydifference(signal_a, signal_b, upsampling):
    """Delay of b relative to a in (non-upsampled) samples, > 0 if b comes after a (upsample_irs.m:58-77)."""
    a = np.asarray(signal_a, dtype=np.float64).ravel()
    b = np.asarray(signal_b, dtype=np.float64).ravel()
    if a.size != b.size:
        raise ValueError("delaydifference: signals of different length")     # :62
    n = a.size
    xc = np.convolve(a[::-1], b)                             # cross-correlation, length 2n-1 (:66)
    xc_up = _resample(xc, upsampling)
    k = int(np.argmax(xc_up))                                # 0-based peak (:69; first maximum, like Octave's max)
    if k < 1 or k > xc_up.size - 2:
        raise ValueError("delaydifference: cross-correlation peak at the edge of its support")   # (:70 indexes out of range)
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
    xc_up = _resample(xc, upsampling, axis=1)
    k = np.argmax(xc_up, axis=1)                             # first maximum of every row, like Octave's max (:69)
    if np.any(k < 1) or np.any(k > xc_up.shape[1] - 2):
        raise ValueError("delaydifferences_from: cross-correlation peak at the edge of its support")
    rows = np.arange(xc_up.shape[0])
    lo, mid, hi = xc_up[rows, k - 1], xc_up[rows, k], xc_up[rows, k + 1]
    # the reference's preconditions (upsample_irs.m:90-98): the middle sample is the first maximum (a tie with the
    # sample before it is rejected), the parabola is not flat - exceptions, not asserts: they must survive python -O
    if not (np.all(mid > lo) and np.all(mid >= hi)):
        raise ValueError("delaydifferences_from: the middle point is not the (first) maximum")
    a = 0.5 * (lo + hi - 2 * mid)
    b = 0.5 * (hi - lo)
    if np.any(a == 0):
        raise ValueError("delaydifferences_from: three collinear points around a cross-correlation peak")
    peak = k - b / (2 * a)
    return peak / upsampling - (n - 1)


def upsample_irs(hrirs_left, hrirs_right, upsampling=8, progress=None):
    """hrirs_*: (n_dir, n_taps) arrays (the `content_m` matrices of the IRCAM structs).  Returns a dict
    with the five fields of the reference's struct (upsample_irs.m:46-51).  187 x 512 taps, U = 8: about
    half a minute on one host core (the per-pair Python loop of round 1 needed ~20 minutes)."""
    import scipy.signal
    hl = np.asarray(hrirs_left, dtype=np.float64)
    hr = np.asarray(hrirs_right, dtype=np.float64)
    if hl.shape != hr.shape or hl.ndim != 2:
        raise ValueError("upsample_irs: need two (n_dir, n_taps) arrays of equal shape")
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
    irs_left = _resample(hl, upsampling, axis=1)             # (:37-44)
    irs_right = _resample(hr, upsampling, axis=1)
    if irs_left.shape != (n_dir, n_taps * upsampling):
        raise ValueError("upsample_irs: resampled length is not n_taps * upsampling")
    return {"upsampling": float(upsampling), "diffs_left": dl, "diffs_right": dr,
            "irs_left": irs_left, "irs_right": irs_right}


_DD_ERRORS = {1: "delaydifference: cross-correlation peak at the edge of its support",           # (upsample_irs.m:70)
              2: "parabolic_interpolation: the middle point is not the (first) maximum",       # (:92-93)
              3: "parabolic_interpolation: three collinear points"}                             # (:98)


def upsample_irs_device(hrirs_left, hrirs_right, upsampling=8, device=None):
    """upsample_irs on the GPU: the same dict (numpy float64 arrays), computed by bas_delaydiffs_f64 (every pair's
    cross-correlation, x U resampling, first maximum, parabola: upsample_irs.m:15-32, :58-101) and bas_resample_up_f64
    (:37-44).  The resampling filter is designed on the host (octave_resample_filter) and handed to the kernels.  The
    reference's preconditions raise ValueError naming the first pair that failed, as the host form does.  Raises if the
    HIP library or a GPU is missing (no host fallback)."""
    import torch
    from . import _hip
    hl = np.ascontiguousarray(hrirs_left, dtype=np.float64)
    hr = np.ascontiguousarray(hrirs_right, dtype=np.float64)
    if hl.shape != hr.shape or hl.ndim != 2:
        raise ValueError("upsample_irs: need two (n_dir, n_taps) arrays of equal shape")
    p = int(upsampling)
    if p < 1 or p != upsampling:
        raise ValueError("upsample_irs: the upsampling factor must be a positive integer")
    n_dir, n_taps = hl.shape
    dev = torch.device(device if device is not None else "cuda")
    _hip.require_gpu(dev)
    h, lh = octave_resample_filter(p, 1)
    with _hip.on_device(dev):
        stream = _hip.current_stream(dev)
        h_d = torch.from_numpy(h).to(dev)
        out = {"upsampling": float(p)}
        for ear, x in (("left", hl), ("right", hr)):
            x_d = torch.from_numpy(x).to(dev)
            diffs = torch.empty((n_dir, n_dir), dtype=torch.float64, device=dev)
            status = torch.empty((1,), dtype=torch.int64, device=dev)
            irs = torch.empty((n_dir, n_taps * p), dtype=torch.float64, device=dev)
            _hip.call("bas_delaydiffs_f64", _hip.ptr(x_d), n_dir, n_taps, _hip.ptr(h_d), lh, p, _hip.ptr(diffs),
                      _hip.ptr(status), stream)
            _hip.call("bas_resample_up_f64", _hip.ptr(x_d), n_dir, n_taps, _hip.ptr(h_d), lh, p, _hip.ptr(irs), stream)
            st = int(status.cpu()[0])                        # (synchronises)
            if st != 0:                                      # ((n_dir^2 - pair) << 2) | code of the smallest failing pair (bas.h)
                pair = n_dir * n_dir - (st >> 2)
                raise ValueError(f"{_DD_ERRORS.get(st & 3, 'delaydifference failed')} ({ear} ear, directions "
                                 f"{pair // n_dir} and {pair % n_dir})")
            out["diffs_" + ear] = diffs.cpu().numpy()
            out["irs_" + ear] = irs.cpu().numpy()
    return out


def save(path, table):
    """Write the struct the way `save -6` does for the reference (upsample_irs.m:53): MATLAB v5."""
    import scipy.io
    scipy.io.savemat(path, {"irs_and_delaydiffs": table}, format="5")


def load_ircam_hrirs(path):
    """The two `content_m` matrices (n_dir, n_taps) of an IRCAM LISTEN file such as IRC_1032_C_HRIR.mat - the structs
    `l_eq_hrir_S` / `r_eq_hrir_S` that upsample_irs.m:1, :25-26 reads (the raw files hold `l_hrir_S` / `r_hrir_S`, :2)."""
    import scipy.io
    m = scipy.io.loadmat(path)
    for lname, rname in (("l_eq_hrir_S", "r_eq_hrir_S"), ("l_hrir_S", "r_hrir_S")):
        if lname in m and rname in m:
            return (np.asarray(m[lname][0][0]["content_m"], dtype=np.float64),
                    np.asarray(m[rname][0][0]["content_m"], dtype=np.float64))
    raise ValueError(f"{path}: no l_eq_hrir_S / r_eq_hrir_S (or l_hrir_S / r_hrir_S) structs")


def build_table(hrir_mat, out_mat="irs_and_delaydiffs.mat", upsampling=8, on_host=False, device=None):
    """What running upsample_irs.m does: HRIR database in, `irs_and_delaydiffs` struct out (:53 saves it under that name;
    the reference's loader expects the file renamed, apply_hrtf.py:23).  On the GPU unless on_host (the numpy definition)."""
    hl, hr = load_ircam_hrirs(hrir_mat)
    table = upsample_irs(hl, hr, upsampling) if on_host else upsample_irs_device(hl, hr, upsampling, device=device)
    save(out_mat, table)
    return table


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Build the HRIR table load_irs_and_delaydiffs reads (the reference's upsample_irs.m).")
    ap.add_argument("hrir_mat", help="IRCAM LISTEN HRIR file, e.g. IRC_1032_C_HRIR.mat")
    ap.add_argument("out_mat", nargs="?", default="irs_and_delaydiffs.mat")
    ap.add_argument("--upsampling", type=int, default=8)
    ap.add_argument("--host", action="store_true", help="the numpy definition instead of the HIP kernels (~10 s)")
    a = ap.parse_args(argv)
    t = build_table(a.hrir_mat, a.out_mat, a.upsampling, on_host=a.host)
    print(f"{a.out_mat}: {t['irs_left'].shape[0]} directions x {t['irs_left'].shape[1]} samples, upsampling {int(t['upsampling'])}")


if __name__ == "__main__":
    main()
