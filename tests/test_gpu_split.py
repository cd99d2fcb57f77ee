"""GPU tests of the split-role fused kernel (bas_fused_split.hip: one workgroup of four filter and four stager waves per
CU, two LDS buffers; the whole unit of a 128-tap segment as one assembly block): parity against the oracle on small scenes
through the diagnostic build (which gives the kernel to any scene on request), on a scene big enough for the shipped
library to pick it, and against the two-workgroups-per-CU kernel it replaces for big scenes."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, rel_err
from oracle import bas_oracle as orc
import binaural_audio_synthesis_amd as bas

pytestmark = pytest.mark.gpu
REL = 1e-5


def _scene(l, n_src, n, k, seed=0):
    h = bas.synth.make_table("consistent", 0).truncated(l)
    sigs = np.stack([bas.synth.integer_noise(seed + 900 + i, n, 0.3 / n_src) for i in range(n_src)])
    in_length = -(-n // k) * k
    t = np.arange(0, in_length + 1, k, dtype=np.float64)
    elev = np.empty((n_src, t.size))
    azim = np.empty((n_src, t.size))
    for i in range(n_src):
        name = ("spiral", "circle_askew", "passing")[i % 3]
        elev[i], azim[i] = bas.synth.trajectory(name, period_s=0.05 + 0.011 * i, length_s=n / 44100, turns=1.0 + i % 7,
                                                phase=0.37 * i)(t)
    return h, sigs, elev, azim, in_length


def _oracle(h, sigs, elev, azim, k, s):
    irs = [np.stack([orc.interp2d(h, elev[i, c], azim[i, c]) for c in range(elev.shape[1])]) for i in range(sigs.shape[0])]
    return orc.render_mix(sigs, k, s, irs, normalize=False)


def _plan_code(lib, n_src, in_length, k, s, l):
    lib.bas_debug_fused_plan.argtypes = [ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return lib.bas_debug_fused_plan(n_src, in_length, k, s, l)


@pytest.mark.parametrize("n_src,n,k,s,l", [
    (3, 30000, 512, 32, 128),      # one unit per workgroup, every unit one whole 128-tap segment (the unit block)
    (40, 60000, 512, 32, 128),     # 320 units on 160 workgroups: two units each through the two LDS buffers
    (47, 70000, 512, 32, 121),     # L rounded up to 128 taps: the unit block with zero taps at the end
    (5, 40000, 512, 32, 100),      # 104 taps (the reference's default length): the unit block of 26 octets
    (44, 70000, 512, 32, 90),      # 96 taps: per-step blocks, masks 0xf0 0xff 0xff 0x0f, two units per workgroup
    (3, 30000, 512, 32, 300),      # three tap segments (128 + 128 + 48)
    (2, 50000, 1024, 64, 128),     # longer chunks: fewer slots
    (4, 30000, 576, 96, 128),      # subchunk size not a power of two
    (4, 30000, 448, 32, 128),      # 19-20 chunk slots under a tile: the largest LDS image that still fits twice
    (1, 60000, 512, 32, 128),      # one source: direct output, no slabs
    (41, 9000, 512, 32, 128),      # 41 parts per tile: the wide reduce kernel behind it
])
def test_split_kernel_vs_oracle_small_scenes(monkeypatch, n_src, n, k, s, l):
    """Diagnostic build, BAS_FZ_SPLIT=1 BAS_FZ_NW=4: the split-role kernel on scenes far smaller than the ones the shipped
    library gives it - same tolerance against the oracle as every other FIR kernel, and against the two-per-CU kernel."""
    h, sigs, elev, azim, in_length = _scene(l, n_src, n, k)
    want = _oracle(h, sigs, elev, azim, k, s)
    monkeypatch.setenv("BAS_FZ_NW", "4")
    with bas._hip.use_library(bas._hip.DIAG_LIB_PATH) as lib:
        d = bas.irs_and_delaydiffs(h.upsampling, h.diffs_left, h.diffs_right, h.irs_left, h.irs_right)
        monkeypatch.setenv("BAS_FZ_SPLIT", "1")
        code = _plan_code(lib, n_src, in_length, k, s, l)
        if k == 448 and not code & 32:
            pytest.skip("20 chunk slots: two LDS buffers do not fit, the plan keeps the two-per-CU kernel")
        assert code & 32 and code & 15 == 4, code
        got = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none", fused=True).cpu().numpy()
        monkeypatch.setenv("BAS_FZ_SPLIT", "0")
        assert not _plan_code(lib, n_src, in_length, k, s, l) & 32
        two = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none", fused=True).cpu().numpy()
    assert got.shape == want.shape and rel_err(got, want) <= REL, rel_err(got, want)
    assert rel_err(got, two) <= 5e-6, rel_err(got, two)       # (same arithmetic per unit; the slabs split the sources differently)


def test_shipped_library_picks_the_split_kernel_for_big_scenes():
    """48 sources x 140 000 samples = 864 (tile, source) units, more than one per CU: the shipped library runs the split-role
    kernel (the diagnostic build, which plans the same way, says so), three to four units per workgroup; vs the oracle."""
    n_src, n, k, s, l = 48, 140000, 512, 32, 128
    h, sigs, elev, azim, in_length = _scene(l, n_src, n, k, seed=50)
    with bas._hip.use_library(bas._hip.DIAG_LIB_PATH) as lib:
        assert _plan_code(lib, n_src, in_length, k, s, l) & 32
        assert not _plan_code(lib, 8, in_length, k, s, l) & 32          # 144 units: no workgroup would have a second one
    d = bas.irs_and_delaydiffs(h.upsampling, h.diffs_left, h.diffs_right, h.irs_left, h.irs_right)
    got = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none").cpu().numpy()
    want = _oracle(h, sigs, elev, azim, k, s)
    assert got.shape == want.shape and rel_err(got, want) <= REL, rel_err(got, want)
    again = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none").cpu().numpy()
    assert np.array_equal(got, again)                                    # deterministic: no atomics on the audio


def test_split_kernel_random_shapes():
    """tools/stress_fused.py ... split: random IR lengths, chunk / subchunk sizes, source counts and lengths on the
    adversarial table with the split-role kernel wherever its two LDS buffers fit, against the oracle."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_fused.py"), "20", "11", "split"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("worst") and "fs_kernel<128>" in last and "fs_kernel<0>" in last, last
