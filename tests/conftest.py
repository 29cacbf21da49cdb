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


def pytest_collection_modifyitems(config, items):
    """Order of the files under `pytest -x`: the parity tests at BASELINE sizes first, then the
    other parity files, and the subprocess / profiler / multi-rank contract tests last -- a
    failure there can never hide a parity test (round 5's record lost 233 of 248 tests that way)."""
    def rank(item):
        name = os.path.basename(str(item.fspath))
        if name == "test_gpu_baseline_sizes.py":
            return 0
        if name == "test_gpu_zz_contract.py":
            return 2
        return 1
    items.sort(key=rank)   # stable: the order inside each class stays pytest's


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


def rel_rms_outside_bins(wav, ref, bad_bins, guard=3, n_fft=512, hop=256):
    """Two waveforms compared bin by bin in the STFT domain, leaving out `bad_bins` (+- guard:
    the Hann window spreads a bin over its neighbours) and after fitting ONE real scale (the
    max-abs renorm of a wave depends on every bin, the left-out ones included).  For real
    recordings whose GEV pencil is singular in a few bins: there the reference's answer is
    scipy.linalg.eig's (libs/beamformer.py:54-59), i.e. rounding noise of arbitrary size, and the
    rest of the spectrum is still a well-posed comparison."""
    from oracle import np_oracle as o
    A = o.librosa_stft(np.asarray(wav, dtype=np.float64), n_fft, hop, center=True)
    B = o.librosa_stft(np.asarray(ref, dtype=np.float64), n_fft, hop, center=True)
    keep = np.ones(A.shape[0], dtype=bool)
    for b in np.asarray(bad_bins, dtype=int):
        keep[max(0, b - guard):b + guard + 1] = False
    if not keep.any():
        return None, 0
    A, B = A[keep], B[keep]
    scale = float(np.real(np.vdot(A, B)) / max(np.real(np.vdot(A, A)), 1e-300))
    return rms(A * scale, B) / max(rms(B), 1e-30), int(keep.sum())
