"""Pins the CPU oracle (oracle/bas_oracle.py) to outputs of the unmodified reference.

The fixtures were produced by tests/golden/make_golden.py, which imports
/root/reference/apply_hrtf.py and sphere.py.  float64 results must agree
EXACTLY (the oracle performs the same IEEE operations in the same order);
float32 render outputs must be bit-identical too.
"""
import json

import numpy as np
import pytest

from conftest import golden
from oracle import bas_oracle as orc
import binaural_audio_synthesis_amd as bas


def test_ring_table_matches_reference_layout():
    t = orc.ring_table()
    assert t.shape == (187, 3) and t.dtype == np.float32
    assert t[72, 1] == 0 and t[72, 2] == 0 and t[186, 0] == 186
    assert t[73, 2] == np.float32(15) * np.float32(2 * np.pi / 360)


@pytest.mark.parametrize("kind", ["pyfloat", "f64"])
def test_azim_params(kind):
    g = golden("azim_params.npz")
    conv = float if kind == "pyfloat" else np.float64
    for i, (e, z) in enumerate(zip(g["elev"], g["azim"])):
        b, a, af = orc.azim_params(np.float64(e), conv(z))
        assert (b, af) == (g[f"before_{kind}"][i], g[f"after_{kind}"][i]), (i, e, z)
        assert float(a) == g[f"a_{kind}"][i]
        assert isinstance(a, np.float32) == bool(g[f"a_is_f32_{kind}"][i])
    # the dtype split documented in SURVEY.md section 7 is real on this numpy
    assert g["a_is_f32_pyfloat"].any() and not g["a_is_f32_f64"].any()


def test_azim_params_invalid_ring_raises():
    with pytest.raises(ValueError):
        orc.azim_params(np.float64(0.1), np.float64(1.0))


def test_frac_shift():
    g = golden("delay_signal.npz")
    x = g["x"]
    for i, s in enumerate(g["shifts"]):
        for down in (1, 8):
            want = g[f"y{i}_d{down}"]
            got = orc.frac_shift(x, float(s), down)
            assert got.shape == want.shape
            assert np.array_equal(got, want), (i, s, down)


def test_ring_interp(tables):
    g = golden("ring_interp.npz")
    tb = {0: tables["consistent"].truncated(128), 1: tables["adversarial"].truncated(128)}
    for k in range(int(g["n"])):
        kind, p, q, alpha, up = g[f"c{k}_meta"]
        dl, dr, irs = orc.ring_interp(tb[int(kind)], int(p), int(q), alpha, bool(up))
        assert np.array_equal(np.array([dl, dr]), g[f"c{k}_delays"])
        assert np.array_equal(irs, g[f"c{k}_irs"]), k


@pytest.mark.parametrize("tname", ["consistent", "adversarial"])
@pytest.mark.parametrize("l", [128, 100])
def test_interp2d(tables, tname, l):
    g = golden("interp2d.npz")
    tb = tables[tname].truncated(l)
    want = g[f"{tname}_{l}"]
    for i, (e, z) in enumerate(g["points"]):
        got = orc.interp2d(tb, np.float64(e), np.float64(z))
        assert got.shape == (2, l)
        assert np.array_equal(got, want[i]), (i, e, z)


RENDER_CASES = ["circle_512_32_128", "sweep_512_32_128", "spiral_512_32_128", "spiral_512_512_128",
                "askew_128_16_100", "spiral_512_32_100", "loud_512_32_128", "silent_512_32_128",
                "exact_multiple_256_8_128", "short_512_32_128"]


@pytest.mark.parametrize("name", RENDER_CASES)
def test_render(tables, name):
    g = golden(f"render_{name}.npz")
    meta = json.loads(str(g["meta"]))
    tb = tables[meta["table"]].truncated(meta["L"])
    traj = bas.synth.trajectory(meta["traj"], fs=meta["fs"], **meta["traj_kw"])
    got = orc.render(g["x"], meta["K"], meta["S"], traj, tb)
    want = g["y"]
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got, want)
    if name.startswith("loud"):
        assert np.abs(want).max() == 1.0          # the peak rule fired (apply_hrtf.py:462-464)
    else:
        assert np.abs(want).max() < 1.0


def test_render_pyfloat_trajectory(tables):
    """The reference CLI's own lambda returns a Python-float azimuth (float32 branch)."""
    g = golden("render_pyfloat_circle.npz")
    meta = json.loads(str(g["meta"]))
    k = 2 * np.pi / (meta["period_s"] * meta["fs"])
    traj = lambda t: (0, (k * t) % (2 * np.pi))     # noqa: E731
    got = orc.render(g["x"], meta["K"], meta["S"], traj, tables["consistent"].truncated(meta["L"]))
    assert np.array_equal(got, g["y"])


def test_render_mix_of_one_source_is_render(tables):
    g = golden("render_spiral_512_32_128.npz")
    meta = json.loads(str(g["meta"]))
    tb = tables["consistent"].truncated(128)
    traj = bas.synth.trajectory(meta["traj"], fs=meta["fs"], **meta["traj_kw"])
    in_length, _ = orc.render_lengths(g["x"].size, 512, 128)
    irs = orc.chunk_irs(tb, 512, in_length, traj)
    got = orc.render_mix([g["x"]], 512, 32, [irs])
    assert np.array_equal(got, g["y"])


# ---- legacy 1-D path (apply_hrtf.py:108-125, :294-353), goldens from tests/golden/make_golden_legacy.py
LEGACY_CASES = ["ring0_sweep_512_128", "ring0_wrap_256_100", "low_ring_500_128", "loud_512_128"]


@pytest.mark.parametrize("tname", ["consistent", "adversarial"])
def test_legacy_ring_easy(tables, tname):
    g = golden("legacy_ring_easy.npz")
    tb = tables[tname].truncated(128)
    for i, ci in enumerate(g["ci"]):
        assert np.array_equal(orc.ring_easy(tb, float(ci)), g[f"{tname}_easy"][i]), ci
    for i, (p, q, a) in enumerate(((72, 73, 0.3), (10, 11, 0.0), (186, 186, 0.5))):
        assert np.array_equal(orc.ring_interp_irs(tb, p, q, a), g[f"{tname}_plain"][i])
    assert orc.ring_easy_params(96.5) == (96, 73, 0.5)          # the hard-wired wrap, apply_hrtf.py:121-122


@pytest.mark.parametrize("name", LEGACY_CASES)
def test_legacy_render_1d(tables, name):
    g = golden(f"render1d_{name}.npz")
    meta = json.loads(str(g["meta"]))
    tb = tables[meta["table"]].truncated(meta["L"])
    got = orc.render_1d(g["x"], meta["K"], bas.synth.index_function(meta["index_function"], meta["n"]), tb)
    assert got.dtype == np.float32 and np.array_equal(got, g["y"])
    assert (np.abs(g["y"]).max() == 1.0) == name.startswith("loud")


def test_render_window_agrees_with_whole_render(tables):
    """The windowed definition used for full-size GPU spot checks equals the loop restatement."""
    g = golden("render_spiral_512_32_128.npz")
    meta = json.loads(str(g["meta"]))
    tb = tables["consistent"].truncated(128)
    traj = bas.synth.trajectory(meta["traj"], fs=meta["fs"], **meta["traj_kw"])
    x = g["x"]
    in_length, out_length = orc.render_lengths(x.size, 512, 128)
    irs = orc.chunk_irs(tb, 512, in_length, traj)
    xp = np.concatenate([x, np.zeros(in_length - x.size)])
    for n0, n1 in ((0, 70), (480, 560), (5000, 5100), (out_length - 90, out_length)):
        m0 = max(n0 - 127, 0)
        got = orc.render_window(xp[m0:min(n1, in_length)], m0, 512, 32, lambda c: irs[c], 128, n0, n1)
        assert np.abs(got.T - g["y"][n0:n1]).max() <= 1e-6 * np.abs(g["y"]).max()


@pytest.mark.parametrize("name", ["spiral_512_32_128", "askew_128_16_100", "loud_512_32_128", "short_512_32_128"])
def test_c_restatement_of_the_render_loops(tables, name):
    """oracle/bas_oracle_fir.c (the checker of the pure-C ABI caller) against the reference-generated goldens,
    fed with the numpy oracle's chunk IRs; 1e-6 norm-relative (numpy's convolve sums in another order)."""
    import ctypes
    import os
    import subprocess
    odir = os.path.dirname(os.path.abspath(orc.__file__))
    subprocess.check_call(["make", "-s", "-C", odir, "all"])
    lib = ctypes.CDLL(os.path.join(odir, "libbas_oracle_fir.so"))
    lib.bas_oracle_in_length.restype = ctypes.c_long
    lib.bas_oracle_in_length.argtypes = [ctypes.c_long, ctypes.c_int]
    vp = ctypes.c_void_p
    lib.bas_oracle_render_accumulate.argtypes = [vp, ctypes.c_long, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp]
    lib.bas_oracle_finish.argtypes = [vp, ctypes.c_long, ctypes.c_int, vp]
    g = golden(f"render_{name}.npz")
    meta = json.loads(str(g["meta"]))
    tb = tables[meta["table"]].truncated(meta["L"])
    traj = bas.synth.trajectory(meta["traj"], fs=meta["fs"], **meta["traj_kw"])
    x = np.ascontiguousarray(g["x"], dtype=np.float64)
    k, s_, l = meta["K"], meta["S"], meta["L"]
    in_length, out_length = orc.render_lengths(x.size, k, l)
    assert lib.bas_oracle_in_length(x.size, k) == in_length
    irs = np.ascontiguousarray(orc.chunk_irs(tb, k, in_length, traj), dtype=np.float64)
    acc = np.zeros((2, out_length))
    out = np.empty((out_length, 2), dtype=np.float32)
    lib.bas_oracle_render_accumulate(x.ctypes.data, x.size, k, s_, irs.ctypes.data, l, acc.ctypes.data)
    lib.bas_oracle_finish(acc.ctypes.data, out_length, 1, out.ctypes.data)
    assert np.abs(out - g["y"]).max() <= 1e-6 * max(np.abs(g["y"]).max(), 1e-30)
