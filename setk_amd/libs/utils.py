"""
Mirror of scripts/sptk/libs/utils.py for the hot path: forward_stft /
inverse_stft run on the MI355X through libsetk_hip.so (setk_stft / setk_istft),
wave I/O through setk_amd.libs.wavio.  Same names, defaults and return layouts
as the reference (utils.py:25-173, 210-245).
"""
import logging
import math
import os
import warnings

import numpy as np

from . import wavio
from .. import _ffi

MAX_INT16 = np.iinfo(np.int16).max
EPSILON = np.finfo(np.float32).eps
default_format_str = "%(asctime)s [%(pathname)s:%(lineno)s - %(levelname)s ] %(message)s"

__all__ = [
    "forward_stft", "inverse_stft", "get_logger", "filekey", "write_wav", "read_wav",
    "cmat_abs", "nextpow2", "EPSILON", "stft_window"
]


def nextpow2(window_size):
    # reference utils.py:25-27
    return 2**math.ceil(math.log2(window_size))


def cmat_abs(cmat):
    # reference utils.py:30-42
    if not np.iscomplexobj(cmat):
        raise RuntimeError(f"function cmat_abs expect complex as input, but got {cmat.dtype}")
    return np.sqrt(cmat.real**2 + cmat.imag**2)


def write_wav(fname, samps, sr=16000, normalize=True):
    """Single/multi-channel wave writer, PCM_16 (reference utils.py:45-62)."""
    samps = samps.astype("float32" if normalize else "int16")
    if samps.ndim != 1 and samps.shape[0] < samps.shape[1]:
        samps = np.transpose(samps)
        samps = np.squeeze(samps)
    fdir = os.path.dirname(fname)
    if fdir and not os.path.exists(fdir):
        os.makedirs(fdir)
    wavio.write(fname, samps, sr)


def read_wav(fname, beg=0, end=None, normalize=True, sr=16000):
    """Returns C x N (or N) float32 (reference utils.py:65-92)."""
    samps, ret_sr = wavio.read(fname, start=beg, stop=end,
                               dtype="float32" if normalize else "int16")
    if sr != ret_sr:
        raise RuntimeError(f"Expect sr={sr} of {fname}, get {ret_sr} instead")
    if not normalize:
        samps = samps.astype("float32")
    if samps.ndim != 1:
        samps = np.transpose(samps)
    return samps


# scipy.signal.windows.general_cosine(M, a, sym=False), operation for operation (same bits):
# the three windows the STFT options are normally given, without the quarter second that
# importing scipy.signal costs a command-line run
_COSINE_WINDOWS = {"hann": (0.5, 0.5), "hamming": (0.54, 1.0 - 0.54),
                   "blackman": (0.42, 0.50, 0.08)}


def _cosine_window(name, frame_len):
    fac = np.linspace(-np.pi, np.pi, frame_len + 1)
    w = np.zeros(frame_len + 1)
    for k, a in enumerate(_COSINE_WINDOWS[name]):
        w += a * np.cos(k * fac)
    return w[:-1]


def stft_window(window, frame_len):
    """The analysis/synthesis window the reference hands to librosa:
    scipy.signal.get_window(name, frame_len, fftbins=True), the sqrt-hann
    special case (utils.py:116-117), or a caller supplied array."""
    if isinstance(window, str):
        if window == "sqrthann":
            w = _cosine_window("hann", frame_len)**0.5
        elif window in _COSINE_WINDOWS and frame_len > 1:
            w = _cosine_window(window, frame_len)
        else:
            import scipy.signal
            w = scipy.signal.get_window(window, frame_len, fftbins=True)
    else:
        w = np.asarray(window, dtype=np.float64)
        if w.shape != (frame_len,):
            raise ValueError("window size mismatch")
    return np.ascontiguousarray(w, dtype=np.float32)


def _plan(ctx, frame_len, frame_hop, n_fft, center, window):
    ctx.stft_plan(frame_len, frame_hop, n_fft, center, stft_window(window, frame_len))


def device_stft(samps, frame_len, frame_hop, round_power_of_two, center, window, ctx=None):
    """samps: C x N float32 -> C x T x F complex64 (time major, device layout)."""
    ctx = ctx or _ffi.default_context()
    n_fft = nextpow2(frame_len) if round_power_of_two else frame_len
    _plan(ctx, frame_len, frame_hop, n_fft, center, window)
    samps = np.ascontiguousarray(samps, dtype=np.float32)
    C, N = samps.shape
    T = ctx.num_frames(N)
    out = np.empty((C, T, n_fft // 2 + 1), dtype=np.complex64)
    ctx.stft(samps, out)
    return out


# return F x T or T x F (tranpose=True)
def forward_stft(samps,
                 frame_len=1024,
                 frame_hop=256,
                 round_power_of_two=True,
                 center=False,
                 window="hann",
                 apply_abs=False,
                 apply_log=False,
                 apply_pow=False,
                 transpose=True):
    """STFT of a mono signal (reference utils.py:96-138)."""
    if apply_log and not apply_abs:
        warnings.warn("Ignore apply_abs=False because apply_log=True")
        apply_abs = True
    if samps.ndim != 1:
        raise RuntimeError("Invalid shape, librosa.stft accepts mono input")
    spec = device_stft(samps[None], frame_len, frame_hop, round_power_of_two, center, window)[0]
    # spec: T x F (C order); F x T is its transposed view (Fortran order, like
    # the array librosa returns)
    stft_mat = spec if transpose else spec.T
    if apply_abs:
        stft_mat = cmat_abs(stft_mat)
    if apply_pow:
        stft_mat = np.power(stft_mat, 2)
    if apply_log:
        stft_mat = np.log(np.maximum(stft_mat, EPSILON))
    return stft_mat


# accept F x T or T x F (tranpose=True)
def inverse_stft(stft_mat,
                 frame_len=1024,
                 frame_hop=256,
                 center=False,
                 window="hann",
                 transpose=True,
                 norm=None,
                 power=None,
                 nsamps=None):
    """iSTFT (reference utils.py:142-173); n_fft = 2 (F - 1)."""
    ctx = _ffi.default_context()
    tf = stft_mat if transpose else np.transpose(stft_mat)  # T x F
    T, F = tf.shape
    n_fft = 2 * (F - 1)
    wide = tf.dtype == np.complex128
    _plan(ctx, frame_len, frame_hop, n_fft, center, window)
    spec = np.ascontiguousarray(tf, dtype=np.complex64)[None]
    L = ctx.istft_num_samples(T, nsamps)
    samps = np.empty((1, L), dtype=np.float32)
    nrm = np.array([norm], dtype=np.float32) if norm else None
    ctx.istft(spec, 1, T, nsamps, nrm, samps)
    samps = samps[0]
    if wide:
        samps = samps.astype(np.float64)
    if power:
        samps_pow = np.linalg.norm(samps, 2)**2 / samps.size
        samps = samps * np.sqrt(power / samps_pow)
    return samps


def check_doa(geometry, doa, online=False):
    """DoA in degrees: [0, 180] for a linear array, [0, 360) for a circular one
    (reference libs/utils.py:248-263); a list of them in the online mode."""
    for d in (doa if online else [doa]):
        if d < 0:
            return False
        if geometry == "linear" and d > 180:
            return False
        if geometry == "circular" and d >= 360:
            return False
    return True


def filekey(path):
    # reference utils.py:210-221
    fname = os.path.basename(path)
    if not fname:
        raise ValueError(f"{path}: is directory path?")
    token = fname.split(".")
    return token[0] if len(token) == 1 else ".".join(token[:-1])


def get_logger(name, format_str=default_format_str, date_format="%Y-%m-%d %H:%M:%S",
               file=False):
    # reference utils.py:224-245 (same format string; handlers added once)
    logger = logging.getLogger(name)
    logger.setLevel(logging.INFO)
    if not logger.handlers:
        formatter = logging.Formatter(fmt=format_str, datefmt=date_format)
        handlers = [logging.StreamHandler()]
        if file:
            handlers.insert(0, logging.FileHandler(name))
        for hd in handlers:
            hd.setLevel(logging.INFO)
            hd.setFormatter(formatter)
            logger.addHandler(hd)
    return logger
