import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # the in-tree library is git-ignored: (re)build it when a source is newer or it
    # is absent (hipcc cross-compiles for gfx950 with or without a GPU)
    from setk_amd import build
    build.build_library(force=False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rms(a, b=None):
    a = np.asarray(a)
    d = a if b is None else a - np.asarray(b)
    return float(np.sqrt(np.mean(np.abs(d)**2)))


def pcm16_rel_rms(pcm, ref_float):
    """int16 samples (a wav the product wrote) against a float reference quantised
    the way libsndfile does (lrint(x * 32767)): both sides carry the same PCM16
    floor, so north_star's 1e-3 applies to what is left."""
    q = np.rint(np.asarray(ref_float, dtype=np.float64) * 32767.0)
    return rms(np.asarray(pcm, dtype=np.float64), q) / max(rms(q), 1e-30)


def rel_rms(a, ref):
    return rms(a, ref) / max(rms(ref), 1e-30)
