"""Structural tests of the offline table builder (parity UNPINNED: no Octave, no IRCAM data;
see binaural-audio-synthesis_amd/upsample_irs.py)."""
import importlib

import numpy as np
import pytest
import scipy.io

up = importlib.import_module("binaural-audio-synthesis_amd.upsample_irs")


def _pulse(n, pos, width=3.0):
    t = np.arange(n) - pos
    return np.exp(-0.5 * (t / width) ** 2)


def test_parabolic_interpolation_known_answers():
    assert up.parabolic_interpolation([1.0, 2.0, 1.0]) == 0.0
    x0 = 0.3
    f = lambda x: 5.0 - 2.0 * (x - x0) ** 2                  # noqa: E731
    assert up.parabolic_interpolation([f(-1), f(0), f(1)]) == pytest.approx(x0, abs=1e-12)
    with pytest.raises(ValueError):
        up.parabolic_interpolation([3.0, 2.0, 1.0])
    with pytest.raises(ValueError):                          # Octave's max returns the FIRST largest element (:92-93)
        up.parabolic_interpolation([2.0, 2.0, 1.0])
    assert up.parabolic_interpolation([1.0, 2.0, 2.0]) == pytest.approx(0.5)   # a tie with the LAST point passes there too
    with pytest.raises(ValueError):
        up.parabolic_interpolation([2.0, 2.0, 2.0])


def test_octave_resample_filter_parameters():
    """The published design of the signal package's resample for p = 8, q = 1: 60 dB rejection, cutoff 1/16,
    roll-off 1/160 -> half-length ceil(52 / (28.714 / 160)) = 290 (581 taps), Kaiser beta 0.1102 (60 - 8.7)."""
    h, L = up.octave_resample_filter(8, 1)
    assert L == 290 and h.size == 581
    assert np.array_equal(h, h[::-1])                        # linear phase
    assert h[L] == pytest.approx(1.0, abs=1e-15)             # 2 p f_c sinc(0) = 1: input samples pass unchanged
    assert np.allclose(h[L + 8::8], 0.0, atol=1e-15)         # and every other multiple of p is a zero of the sinc
    beta = 0.1102 * (60 - 8.7)
    assert np.allclose(h, np.kaiser(581, beta) * np.sinc(np.arange(-290, 291) / 8.0), atol=1e-15)
    assert h.sum() == pytest.approx(8.0, rel=2e-3)           # pass-band gain p: unit gain per output phase
    h32, L32 = up.octave_resample_filter(3, 2)
    assert L32 == int(np.ceil(52 / (28.714 * (1 / 6) / 10)))
    h2, _ = up.octave_resample_filter(16, 2)                 # common factors are removed first
    assert np.array_equal(h2, h)


@pytest.mark.parametrize("p", [2, 8])
def test_octave_resample_known_answers(p):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(512)
    y = up.octave_resample(x, p, 1)
    assert y.shape == (512 * p,)                             # upsample_irs.m:37-44 relies on exactly 512 U
    assert np.allclose(y[::p], x, rtol=0, atol=1e-14)        # interpolation: the input samples are kept exactly
    # DC: away from the ends (half-length 290 / 8 ~ 36 input samples at p = 8) a constant stays a constant to the
    # pass-band ripple of a 60 dB design
    c = up.octave_resample(np.ones(400), p, 1)
    m = 40 * p
    assert np.allclose(c[m:-m], 1.0, atol=2e-3)
    # a band-limited sinusoid well inside the pass band comes out as the same sinusoid on the fine grid
    f = 0.11
    n = np.arange(600)
    s = up.octave_resample(np.sin(2 * np.pi * f * n + 0.4), p, 1)
    fine = np.sin(2 * np.pi * f * np.arange(600 * p) / p + 0.4)
    assert np.abs(s[m:-m] - fine[m:-m]).max() < 3e-3
    # above the cutoff of the ORIGINAL rate nothing can exist: a 0.45 cycles/sample tone is still reproduced (it is
    # below 0.5), its image at (1 - 0.45) / p cycles per fine sample is rejected by ~60 dB
    tone = up.octave_resample(np.cos(2 * np.pi * 0.45 * n), p, 1)[m:-m]
    spec = np.abs(np.fft.rfft(tone * np.hanning(tone.size)))
    freqs = np.fft.rfftfreq(tone.size)
    image = spec[np.abs(freqs - (1 - 0.45) / p) < 0.004].max()
    main = spec[np.abs(freqs - 0.45 / p) < 0.004].max()
    assert image < 4e-3 * main
    # batched along an axis == row by row; rational factors give ceil(Lx p / q) samples
    xb = rng.standard_normal((3, 100))
    assert np.array_equal(up.octave_resample(xb, p, 1, axis=1), np.stack([up.octave_resample(r, p, 1) for r in xb]))
    assert up.octave_resample(np.ones(101), 3, 2).shape == (152,)


@pytest.mark.parametrize("shift", [0.0, 1.0, -3.0, 2.5, -0.375, 7.125])
def test_delaydifference_of_shifted_pulses(shift):
    n = 128
    a, b = _pulse(n, 40.0), _pulse(n, 40.0 + shift)
    assert up.delaydifference(a, b, 8) == pytest.approx(shift, abs=5e-3)
    assert up.delaydifference(b, a, 8) == pytest.approx(-shift, abs=5e-3)


def test_table_structure_and_mat_layout(tmp_path):
    rng = np.random.default_rng(0)
    n_dir, n_taps = 6, 64
    pos = 20 + rng.uniform(-4, 4, size=(2, n_dir))
    hl = np.stack([_pulse(n_taps, p) for p in pos[0]])
    hr = np.stack([_pulse(n_taps, p) for p in pos[1]])
    t = up.upsample_irs(hl, hr, 8)
    for d, p in ((t["diffs_left"], pos[0]), (t["diffs_right"], pos[1])):
        assert d.shape == (n_dir, n_dir)
        assert np.allclose(d, -d.T) and np.all(np.diag(d) == 0)          # upsample_irs.m:31-32
        assert np.allclose(d, p[None, :] - p[:, None], atol=5e-3)        # mesh rule holds for pure delays
    assert t["irs_left"].shape == (n_dir, n_taps * 8)
    assert np.allclose(t["irs_left"][:, ::8], hl, atol=1e-3)             # resampling keeps the original samples
    path = str(tmp_path / "t.mat")
    up.save(path, t)
    rec = scipy.io.loadmat(path)["irs_and_delaydiffs"][0][0]             # indexing of apply_hrtf.py:38-44
    assert int(rec["upsampling"][0][0]) == 8
    assert rec["irs_right"][:, :16 * 8].shape == (n_dir, 128)
    assert np.array_equal(rec["diffs_left"], t["diffs_left"])


def test_batched_rows_equal_the_per_pair_function():
    """delaydifferences_from (one FFT batch per row of the pair matrix) against delaydifference pair by pair, and the
    reference's preconditions (upsample_irs.m:90-98) as assertions."""
    rng = np.random.default_rng(5)
    n_dir, n_taps = 7, 96
    h = np.stack([_pulse(n_taps, 30 + rng.uniform(-6, 6), width=2.0 + rng.uniform(0, 2)) for _ in range(n_dir)])
    h += 0.01 * rng.standard_normal(h.shape)
    for i in range(n_dir):
        got = up.delaydifferences_from(h, i, 8)
        want = np.array([up.delaydifference(h[i], h[j], 8) for j in range(i + 1, n_dir)])
        assert got.shape == want.shape and np.allclose(got, want, rtol=0, atol=1e-9)
    with pytest.raises(ValueError):                          # a flat correlation has no strict peak (:92-98)
        up.delaydifferences_from(np.zeros((2, 16)), 0, 8)
    with pytest.raises(ValueError):
        up.delaydifference(np.zeros(16), np.zeros(17), 8)


def test_full_size_table_builds_quickly():
    """187 directions x 512 taps, U = 8 (the real table's shape) on synthetic pulses: structure only."""
    import time
    rng = np.random.default_rng(9)
    pos = 40 + rng.uniform(-10, 10, size=(2, 187))
    hl = np.stack([_pulse(512, p) for p in pos[0]])
    hr = np.stack([_pulse(512, p) for p in pos[1]])
    t0 = time.perf_counter()
    t = up.upsample_irs(hl, hr, 8)
    assert time.perf_counter() - t0 < 240
    assert t["irs_left"].shape == (187, 4096) and t["diffs_right"].shape == (187, 187)
    assert np.allclose(t["diffs_left"], pos[0][None, :] - pos[0][:, None], atol=5e-3)


def test_build_table_from_an_ircam_shaped_file(tmp_path):
    """build_table / main: a file with the structs upsample_irs.m:1, :25-26 reads (l_eq_hrir_S.content_m, r_eq_hrir_S.content_m)
    in, the struct apply_hrtf.py:38-44 indexes out (host form here: no GPU)."""
    rng = np.random.default_rng(2)
    n_dir, n_taps = 5, 48
    hl = np.stack([_pulse(n_taps, 18 + rng.uniform(-3, 3)) for _ in range(n_dir)])
    hr = np.stack([_pulse(n_taps, 18 + rng.uniform(-3, 3)) for _ in range(n_dir)])
    src, dst = str(tmp_path / "IRC_test_C_HRIR.mat"), str(tmp_path / "out.mat")
    scipy.io.savemat(src, {"l_eq_hrir_S": {"content_m": hl, "sampling_hz": 44100.0},
                           "r_eq_hrir_S": {"content_m": hr, "sampling_hz": 44100.0}}, format="5")
    gl, gr = up.load_ircam_hrirs(src)
    assert np.array_equal(gl, hl) and np.array_equal(gr, hr)
    up.main([src, dst, "--upsampling", "4", "--host"])
    rec = scipy.io.loadmat(dst)["irs_and_delaydiffs"][0][0]
    want = up.upsample_irs(hl, hr, 4)
    assert int(rec["upsampling"][0][0]) == 4
    assert np.array_equal(rec["irs_left"], want["irs_left"]) and np.array_equal(rec["diffs_right"], want["diffs_right"])
    with pytest.raises(ValueError):
        scipy.io.savemat(src, {"something_else": hl})
        up.load_ircam_hrirs(src)
