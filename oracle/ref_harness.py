"""
Load the *unmodified* reference python modules (funcwj/setk, scripts/sptk) from
/root/reference so they can be run as the ground-truth oracle in the build
container.

*** TEST INFRASTRUCTURE -- NOT PRODUCT CODE ***  (see oracle/np_oracle.py)

/root/reference does not exist on the GPU box: nothing under ``-m gpu``,
``smoke()`` or ``bench.py`` imports this file.  It is used by
``oracle/make_golden.py`` (fixture generation) and by the CPU-side tests that
``skipif`` the reference tree is absent.

The reference cannot be imported as-is in this image: librosa and
soundfile are not installed, numpy 2.x dropped ``np.complex``/``np.int`` and
changed ``np.linalg.solve`` broadcasting, scipy 1.15 dropped
``scipy.signal.hann``.  The five shims below are installed *before* import;
no reference source is copied or edited.
"""
import importlib
import io
import os
import sys
import types

import numpy as np
import scipy.io.wavfile
import scipy.signal

REF_ROOT = os.environ.get("SETK_REFERENCE", "/root/reference")
REF_SPTK = os.path.join(REF_ROOT, "scripts", "sptk")


def available():
    return os.path.isdir(REF_SPTK)


def _install_numpy_shims():
    # (1) aliases removed in numpy>=1.24
    for name, typ in (("complex", complex), ("int", int), ("float", float)):
        if not hasattr(np, name):
            setattr(np, name, typ)
    # (2) numpy<2 semantics: b.ndim == a.ndim-1 means "stack of vectors"
    if not getattr(np.linalg.solve, "_setk_shim", False):
        orig = np.linalg.solve

        def solve(a, b):
            a = np.asarray(a)
            b = np.asarray(b)
            if b.ndim == a.ndim - 1:
                return orig(a, b[..., None])[..., 0]
            return orig(a, b)

        solve._setk_shim = True
        solve._orig = orig
        np.linalg.solve = solve
    # (3) scipy.signal.hann moved to scipy.signal.windows
    if not hasattr(scipy.signal, "hann"):
        scipy.signal.hann = scipy.signal.windows.hann


def _soundfile_module():
    """(4) stand-in for SoundFile on top of scipy.io.wavfile: PCM16 -> float32
    scales by 1/32768, float -> PCM_16 scales by 32767 and rounds
    (libsndfile's documented default conversions)."""
    m = types.ModuleType("soundfile")

    def read(file, start=0, stop=None, dtype="float32"):
        sr, data = scipy.io.wavfile.read(file)
        start = 0 if start is None else start
        data = data[start:stop]
        if dtype == "float32":
            if data.dtype == np.int16:
                data = data.astype(np.float32) / 32768.0
            elif data.dtype == np.int32:
                data = data.astype(np.float32) / 2147483648.0
            else:
                data = data.astype(np.float32)
        else:
            data = data.astype(dtype)
        return data, sr

    def write(file, data, samplerate, subtype=None):
        data = np.asarray(data)
        if data.dtype.kind == "f":
            pcm = np.rint(data.astype(np.float64) * 32767.0)
            pcm = pcm.astype(np.int64).astype(np.int16)  # wraps like libsndfile
        else:
            pcm = data.astype(np.int16)
        scipy.io.wavfile.write(str(file), samplerate, pcm)

    m.read = read
    m.write = write
    return m


def _librosa_module():
    """(5) stand-in exposing librosa.stft / librosa.istft via the restatement."""
    from . import np_oracle as o
    m = types.ModuleType("librosa")

    def stft(y, n_fft=2048, hop_length=None, win_length=None, window="hann",
             center=True, dtype=None, pad_mode="reflect"):
        if win_length is None:
            win_length = n_fft
        if hop_length is None:
            hop_length = win_length // 4
        return o.librosa_stft(y, n_fft, hop_length, win_length=win_length,
                              window=window, center=center)

    def istft(stft_matrix, hop_length=None, win_length=None, window="hann",
              center=True, dtype=None, length=None):
        n_fft = 2 * (stft_matrix.shape[0] - 1)
        if win_length is None:
            win_length = n_fft
        if hop_length is None:
            hop_length = win_length // 4
        return o.librosa_istft(stft_matrix, hop_length, win_length=win_length,
                               window=window, center=center, length=length)

    m.stft = stft
    m.istft = istft
    return m


_loaded = {}


def load():
    """Returns a namespace with the reference's ``libs`` package
    (libs.beamformer, libs.utils, libs.data_handler, libs.kaldi_io,
    libs.cluster, libs.opts) imported unmodified."""
    if "libs" in _loaded:
        return _loaded["libs"]
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_SPTK}")
    _install_numpy_shims()
    sys.modules.setdefault("soundfile", _soundfile_module())
    sys.modules.setdefault("librosa", _librosa_module())
    if REF_SPTK not in sys.path:
        sys.path.insert(0, REF_SPTK)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        libs = importlib.import_module("libs")
        for sub in ("utils", "kaldi_io", "opts", "data_handler", "beamformer",
                    "cluster"):
            importlib.import_module(f"libs.{sub}")
    import logging
    for name in list(logging.root.manager.loggerDict):
        if name.startswith("libs") or name == "__main__":
            logging.getLogger(name).setLevel(logging.ERROR)
    _loaded["libs"] = libs
    return libs


def load_cli(name="apply_adaptive_beamformer"):
    """Import one of the reference CLI scripts as a module (its ``run(args)``
    is then callable with an argparse.Namespace)."""
    load()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return importlib.import_module(name)
