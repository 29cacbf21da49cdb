"""
Device-backed mirror of the part of funcwj/setk ``scripts/sptk/libs/spatial.py``
that consumes the beamformer path's primitives:

    directional_feats(spectrogram, steer_vector, df_pair=None)   (:184-208)

Same argument layout (M x F x T, M x F) and return (T x F) as the reference;
the kernel is setk_directional_feats in libsetk_hip.so.  The geometry-driven
features of that file (SRP, IPD grids, MSC) are outside the hot path.
"""
import numpy as np

from .. import _ffi


def directional_feats(spectrogram, steer_vector, df_pair=None):
    """mean over microphone pairs of cos((arg X_i - arg X_j) - (arg v_i - arg v_j));
    spectrogram M x F x T, steer_vector M x F  ->  T x F float32."""
    spectrogram = np.asarray(spectrogram)
    steer_vector = np.asarray(steer_vector)
    M, F, T = spectrogram.shape
    if steer_vector.shape != (M, F):
        raise ValueError(f"steer_vector {steer_vector.shape} does not match spectrogram "
                         f"{spectrogram.shape}")
    if df_pair is None:
        df_pair = [(i, j) for i in range(M) for j in range(i + 1, M)]
    if not len(df_pair):
        raise ValueError("no microphone pair given")
    spec = np.ascontiguousarray(np.transpose(spectrogram, (0, 2, 1)), dtype=np.complex64)
    sv = np.ascontiguousarray(np.transpose(steer_vector), dtype=np.complex64)  # F x M
    out = np.empty((T, F), dtype=np.float32)
    _ffi.default_context().directional_feats(spec, sv, [tuple(p) for p in df_pair], M, T, F, out)
    return out
