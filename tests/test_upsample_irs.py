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
