import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def tables():
    """The two synthetic tables, sha-guarded against the build container's bytes."""
    import binaural_audio_synthesis_amd as bas
    t = {"consistent": bas.synth.make_table("consistent", 0),
         "adversarial": bas.synth.make_table("adversarial", 1)}
    with open(os.path.join(GOLDEN, "table_sha.json")) as f:
        sha = json.load(f)
    for k, v in t.items():
        assert v.sha256() == sha[k], f"synthetic table '{k}' is not bit-reproducible on this machine"
    return t


def rel_err(got, want):
    """Norm-relative error used for every float32 parity bound:
    max|got-want| / max|want|."""
    want = np.asarray(want, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    denom = np.abs(want).max()
    if denom == 0:
        return float(np.abs(got).max())
    return float(np.abs(got - want).max() / denom)
