"""GPU tests of the table builder's device form (csrc/bas_table.hip: bas_delaydiffs_f64,
bas_resample_up_f64) against the host restatement `upsample_irs.upsample_irs`.  PARITY UNPINNED for both (no Octave, no
IRCAM data: upsample_irs.py's header): what is pinned here is that the kernels compute what the host form defines - every
delay to 1e-9 samples (the host correlates through FFTs, the device sums directly), every resampled sample to 1e-12 - and the
structural properties and known answers of tests/test_upsample_irs.py on the device's own output."""
import importlib
import time

import numpy as np
import pytest

import binaural_audio_synthesis_amd as bas

up = importlib.import_module("binaural-audio-synthesis_amd.upsample_irs")
pytestmark = pytest.mark.gpu


def _pulse(n, pos, width=3.0):
    t = np.arange(n) - pos
    return np.exp(-0.5 * (t / width) ** 2)


def _noisy_pulses(rng, n_dir, n_taps, centre, spread):
    h = np.stack([_pulse(n_taps, centre + rng.uniform(-spread, spread), width=2.0 + rng.uniform(0, 2)) for _ in range(n_dir)])
    return h + 0.01 * rng.standard_normal(h.shape)


@pytest.mark.parametrize("n_dir,n_taps,p", [(7, 96, 8), (12, 64, 8), (5, 200, 4), (3, 33, 2)])
def test_device_table_equals_the_host_restatement(n_dir, n_taps, p):
    rng = np.random.default_rng(n_dir * 1000 + n_taps)
    hl = _noisy_pulses(rng, n_dir, n_taps, 0.4 * n_taps, 0.08 * n_taps)
    hr = _noisy_pulses(rng, n_dir, n_taps, 0.4 * n_taps, 0.08 * n_taps)
    want = up.upsample_irs(hl, hr, p)
    got = up.upsample_irs_device(hl, hr, p)
    assert got["upsampling"] == want["upsampling"] == float(p)
    for ear in ("left", "right"):
        d, dw = got["diffs_" + ear], want["diffs_" + ear]
        assert d.shape == dw.shape == (n_dir, n_dir) and d.dtype == np.float64
        assert np.abs(d - dw).max() <= 1e-9                              # samples
        assert np.array_equal(d, -d.T) and np.all(np.diag(d) == 0)       # upsample_irs.m:31-32, exactly
        r, rw = got["irs_" + ear], want["irs_" + ear]
        assert r.shape == rw.shape == (n_dir, n_taps * p)
        assert np.abs(r - rw).max() <= 1e-12 * max(1.0, np.abs(rw).max())
        src = hl if ear == "left" else hr
        assert np.allclose(r[:, ::p], src, rtol=0, atol=1e-14)           # the input samples are kept (h[Lh] = 1, zeros p apart)


@pytest.mark.parametrize("shift", [0.0, 1.0, -3.0, 2.5, -0.375, 7.125])
def test_device_delay_of_shifted_pulses(shift):
    """Known answers: two copies of one pulse `shift` samples apart (as tests/test_upsample_irs.py asks of the host form)."""
    n = 128
    h = np.stack([_pulse(n, 40.0), _pulse(n, 40.0 + shift)])
    t = up.upsample_irs_device(h, h[::-1].copy(), 8)
    assert t["diffs_left"][0, 1] == pytest.approx(shift, abs=5e-3) and t["diffs_left"][1, 0] == -t["diffs_left"][0, 1]
    assert t["diffs_right"][0, 1] == pytest.approx(-shift, abs=5e-3)


def test_device_preconditions_raise_like_the_host():
    """A flat correlation has no strict peak (upsample_irs.m:92-98): the host form raises ValueError, so does the device
    form, naming the pair; signals whose correlation peaks at the edge of its support (:70) likewise."""
    flat = np.zeros((3, 16))
    with pytest.raises(ValueError):
        up.upsample_irs(flat, flat, 8)
    with pytest.raises(ValueError, match="directions 0 and 1"):
        up.upsample_irs_device(flat, flat, 8)
    ok = np.stack([_pulse(32, 12.0), _pulse(32, 14.0), _pulse(32, 9.0)])
    bad = ok.copy()
    bad[2] = 0.0                                             # one silent direction: its pairs have flat correlations
    with pytest.raises(ValueError, match="left ear, directions 0 and 2"):
        up.upsample_irs_device(bad, ok, 8)
    with pytest.raises(ValueError, match="right ear"):
        up.upsample_irs_device(ok, bad, 8)
    with pytest.raises(ValueError):
        up.upsample_irs_device(ok, ok[:, :16], 8)
    with pytest.raises(ValueError):
        up.upsample_irs_device(ok, ok, 2.5)


def test_full_size_table_on_the_device_and_through_the_loader(tmp_path):
    """187 directions x 512 taps, U = 8 - the real table's shape - on synthetic pulses: the device form against the host
    form (all 2 x 17 391 delays, all 374 resampled HRIRs), its time, and the written file through the product loader."""
    rng = np.random.default_rng(9)
    pos = 40 + rng.uniform(-10, 10, size=(2, 187))
    hl = np.stack([_pulse(512, p) for p in pos[0]]) + 1e-3 * rng.standard_normal((187, 512))
    hr = np.stack([_pulse(512, p) for p in pos[1]]) + 1e-3 * rng.standard_normal((187, 512))
    up.upsample_irs_device(hl[:4], hr[:4], 8)                # (first use of the kernels)
    t0 = time.perf_counter()
    got = up.upsample_irs_device(hl, hr, 8)
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    want = up.upsample_irs(hl, hr, 8)
    t_host = time.perf_counter() - t0
    print(f"table builder, 187 x 512 taps, U = 8: device {t_dev * 1e3:.1f} ms (incl. copies), host numpy {t_host:.1f} s")
    assert t_dev < 5.0
    for ear in ("left", "right"):
        assert np.abs(got["diffs_" + ear] - want["diffs_" + ear]).max() <= 1e-9
        assert np.abs(got["irs_" + ear] - want["irs_" + ear]).max() <= 1e-12
    assert np.allclose(got["diffs_left"], pos[0][None, :] - pos[0][:, None], atol=2e-2)
    path = str(tmp_path / "table.mat")
    up.save(path, got)
    tbl = bas.load_irs_and_delaydiffs(path, samples_to_keep=128)
    assert tbl.upsampling == 8 and tuple(tbl.irs_left.shape) == (187, 1024)
    ir = bas.interpolate_2d(tbl, 0.1, 0.7)
    assert tuple(ir.shape) == (2, 128) and bool(np.isfinite(np.asarray(ir.cpu() if hasattr(ir, "cpu") else ir)).all())


def test_build_table_on_the_device_from_an_ircam_shaped_file(tmp_path):
    import scipy.io
    rng = np.random.default_rng(4)
    hl = _noisy_pulses(rng, 9, 80, 30.0, 5.0)
    hr = _noisy_pulses(rng, 9, 80, 30.0, 5.0)
    src, dst = str(tmp_path / "IRC_test_C_HRIR.mat"), str(tmp_path / "irs_and_delaydiffs.mat")
    scipy.io.savemat(src, {"l_eq_hrir_S": {"content_m": hl}, "r_eq_hrir_S": {"content_m": hr}}, format="5")
    up.main([src, dst])
    rec = scipy.io.loadmat(dst)["irs_and_delaydiffs"][0][0]
    want = up.upsample_irs(hl, hr, 8)
    assert np.abs(rec["diffs_left"] - want["diffs_left"]).max() <= 1e-9
    assert np.abs(rec["irs_right"] - want["irs_right"]).max() <= 1e-12
