"""CPU tests added in round 3 (host logic only; no GPU call).

  * the float32 ("pyfloat") branch of the vectorised angle -> parameter step against the values the unmodified
    reference produced for Python-float azimuths (tests/golden/azim_params.npz) and against the scalar path
    that keeps numpy's own promotion rules;
  * which branch a trajectory function selects (apply_hrtf.trajectory_branch).
"""
import numpy as np
import pytest

from conftest import golden
import binaural_audio_synthesis_amd as bas


@pytest.mark.parametrize("kind", ["pyfloat", "f64"])
def test_batch_branch_against_reference_goldens(kind):
    """sphere.interpolation_params_batch(branch=...) on ring elevations = sphere.azim_to_interpolation_params of the
    reference for that scalar type (576 points: nodes +- 1e-9, 0, 2 pi, negative, > 2 pi, 60 / 75 degree rings, pole)."""
    g = golden("azim_params.npz")
    idx, w = bas.sphere.interpolation_params_batch(g["elev"], g["azim"], branch=kind)
    assert np.array_equal(idx[:, 0], g[f"before_{kind}"]) and np.array_equal(idx[:, 1], g[f"after_{kind}"])
    assert np.array_equal(idx[:, 2], g[f"before_{kind}"]) and np.array_equal(idx[:, 3], g[f"after_{kind}"])
    assert np.array_equal(w[:, 0], g[f"a_{kind}"]) and np.array_equal(w[:, 1], g[f"a_{kind}"])
    assert not w[:, 2].any()
    # the two branches really differ on this fixture (else the test would prove nothing)
    assert (g["before_pyfloat"] != g["before_f64"]).any() and (g["a_pyfloat"] != g["a_f64"]).any()


def test_batch_pyfloat_branch_equals_scalar_numpy_path():
    """Random angles incl. clamped elevations: the vectorised float32 branch is bit-identical to the scalar path fed
    Python floats (which evaluates the reference's own numpy expressions)."""
    rng = np.random.default_rng(31)
    e = rng.uniform(-1.2, 1.9, 3000)
    z = rng.uniform(-20.0, 40.0, 3000)
    nodes = np.deg2rad(np.arange(0, 361, 15, dtype=np.float64))
    e = np.concatenate([e, np.repeat(np.deg2rad([-45.0, 0.0, 37.0, 60.0, 75.0, 90.0]), nodes.size * 3)])
    z = np.concatenate([z, np.tile(np.concatenate([nodes, nodes + 1e-9, nodes - 1e-9]), 6)])
    idx, w = bas.sphere.interpolation_params_batch(e, z, branch="pyfloat")
    for i in range(e.size):
        want_idx, want_w = bas.sphere.interpolation_params(float(e[i]), float(z[i]))
        assert tuple(idx[i]) == tuple(want_idx), (i, e[i], z[i])
        assert tuple(w[i]) == tuple(want_w), (i, e[i], z[i])


def test_batch_branch_name_is_checked():
    with pytest.raises(ValueError):
        bas.sphere.interpolation_params_batch(np.zeros(3), np.zeros(3), branch="f32")


def test_trajectory_branch_of_the_reference_presets():
    """The CLI's presets (apply_hrtf.py:580-593): modulo-of-a-product lambdas return Python floats for the Python-int
    times the reference passes (float32 branch); the ones built on np.sin / np.arctan return np.float64."""
    from binaural_audio_synthesis_amd import cli
    from binaural_audio_synthesis_amd.apply_hrtf import trajectory_branch
    p = cli.presets(44100)
    assert trajectory_branch(p["circle_horizontal"]) == "pyfloat"
    assert trajectory_branch(p["circle_askew"]) == "pyfloat"
    assert trajectory_branch(p["spiral"]) == "pyfloat"
    assert trajectory_branch(p["circle_front"]) == "f64"
    assert trajectory_branch(p["passing"]) == "f64"
    assert trajectory_branch(p["halfcircle_vertical"]) == "f64"
    assert trajectory_branch(lambda t: (0, np.float32(1.0))) is None
    assert trajectory_branch(bas.synth.trajectory("spiral", length_s=1.0, turns=2.0)) == "f64"
