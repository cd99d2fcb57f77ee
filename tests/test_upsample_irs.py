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
    with pytest.raises(AssertionError):
        up.parabolic_interpolation([3.0, 2.0, 1.0])


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
    with pytest.raises(AssertionError):                      # a flat correlation has no strict peak: a == 0 (:97)
        up.delaydifferences_from(np.zeros((2, 16)), 0, 8)


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
