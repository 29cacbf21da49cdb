"""
Mirror of scripts/sptk/libs/wpe.py (funcwj/setk): GWPE dereverberation and the
factorised WPD convolutional beamformer, computed on the MI355X.

    wpe(reverb, taps, delay, context, num_iters)          libs/wpe.py:84-110
    wpe_step(reverb, yt, lambda_)                         libs/wpe.py:58-81
    compute_tap_mat / compute_lambda                      libs/wpe.py:13-55
    facted_wpd(obs, cgmm_iters, wpd_iters, taps, ...)     libs/wpe.py:113-177

The tap-stacked correlation, its factorisation and the prediction filter run
in setk_wpe (csrc/wpe.hip, fp64 like the reference's complex128); facted_wpd
chains it with the device CGMM, the masked / power-weighted covariances, the
principal eigenvector + MVDR solve and the beamformer -- the same kernels as
the adaptive-beamformer path.  Layouts follow the reference: F x N x T for
wpe(), N x T x F for facted_wpd().
"""
import numpy as np

from .. import _ffi
from .utils import EPSILON, get_logger

logger = get_logger(__name__)

__all__ = ["wpe", "wpe_batch", "wpe_step", "compute_tap_mat", "compute_lambda", "facted_wpd"]


def compute_tap_mat(obs, taps, delay):
    """obs F x N x T -> F x NK x T (row k N + n: channel n delayed by k + delay).
    Host helper kept for API parity; the device kernels index the delays in place."""
    F, N, T = obs.shape
    y = np.zeros([F, N * taps, T], dtype=obs.dtype)
    for k in range(taps):
        d = k + delay
        if d >= T:
            break
        y[:, k * N:(k + 1) * N, d:] = obs[:, :, :T - d]
    return y


def compute_lambda(dereverb, ctx=0):
    """Time-varying variance F x T (host helper, same arithmetic as the device kernel)."""
    L = np.mean(dereverb.real**2 + dereverb.imag**2, axis=1)
    _, T = L.shape
    counts = np.zeros(T)
    lam = np.zeros_like(L)
    for c in range(-ctx, ctx + 1):
        s, e = max(c, 0), min(T, T + c)
        lam[:, s:e] += L[:, max(-c, 0):min(T, T - c)]
        counts[s:e] += 1
    return np.maximum(lam / counts, EPSILON)


def _note_rankdef(status, what="WPE"):
    """Rank-deficient tap correlations (fewer frames than channels x taps, a silent or duplicated
    channel): the device drops the columns at the noise level and still returns a filter where
    numpy.linalg.solve in the reference returns a different, noise-determined one -- say so."""
    n = int(np.count_nonzero(np.asarray(status) == _ffi.NUM_RANKDEF))
    if n:
        logger.warning(f"{what}: rank-deficient tap correlation in {n} bins; columns at the noise level "
                       "were dropped (the reference's numpy.linalg.solve gives a noise-determined filter there)")
    return n


def _to_ctf(fnt):
    """F x N x T -> the library's [C][T][F] complex64."""
    return np.ascontiguousarray(np.transpose(fnt, (1, 2, 0)), dtype=np.complex64)


def _run(spec_ctf, taps, delay, context, num_iters, lambda_enh=None, want_inv_lambda=False):
    C, T, F = spec_ctf.shape
    out = np.empty_like(spec_ctf)
    status = np.zeros(F, dtype=np.int32)
    inv = np.empty((T, F), dtype=np.float32) if want_inv_lambda else None
    _ffi.default_context().wpe(spec_ctf, C, T, F, taps, delay, context, num_iters, out,
                               lambda_enh=lambda_enh, inv_lambda_out=inv, status=status)
    _note_rankdef(status)
    if _ffi.wpe_failed(status).any():
        raise np.linalg.LinAlgError(
            f"Singular matrix (tap correlation, {int(np.count_nonzero(_ffi.wpe_failed(status)))} bins)")
    return out, inv


def wpe_step(reverb, yt, lambda_, taps=None, delay=None):
    """One WPE step with caller-supplied variances (libs/wpe.py:58-81).  reverb
    F x N x T, yt F x NK x T, lambda_ F x T -- used as given (float64, no floor).

    The device kernel indexes the delayed frames in place instead of reading a tap
    matrix, so yt must BE compute_tap_mat(reverb, taps, delay): pass taps / delay, or
    they are recovered from yt (taps = NK / N; delay = the shift under which yt's first
    block reproduces reverb, checked on the whole matrix).  A yt that is no tap matrix
    of reverb is refused -- there is no host fallback."""
    reverb = np.asarray(reverb)
    yt = np.asarray(yt)
    F, N, T = reverb.shape
    if yt.ndim != 3 or yt.shape[0] != F or yt.shape[2] != T or yt.shape[1] % N:
        raise ValueError(f"yt {yt.shape} does not stack taps of reverb {reverb.shape}")
    if taps is None:
        taps = yt.shape[1] // N
    if taps * N != yt.shape[1]:
        raise ValueError(f"taps = {taps} but yt has {yt.shape[1]} rows for {N} channels")
    if delay is None:
        # candidates: at most the number of leading all-zero frames of the first block
        # (a signal that itself starts with zero frames makes that an over-estimate)
        nz = np.flatnonzero(np.any(yt[:, :N, :] != 0, axis=(0, 1)))
        lead = int(nz[0]) if nz.size else T
        delay = next((d for d in range(min(lead, T - 1), -1, -1)
                      if np.array_equal(yt[:, :N, d:], reverb[:, :, :T - d].astype(yt.dtype))
                      and np.array_equal(yt, compute_tap_mat(reverb, taps, d).astype(yt.dtype))),
                     None)
        if delay is None:
            raise ValueError("yt is not compute_tap_mat(reverb, taps, delay) for any delay")
    elif not np.array_equal(yt, compute_tap_mat(reverb, taps, delay).astype(yt.dtype)):
        raise ValueError("yt differs from compute_tap_mat(reverb, taps, delay)")
    lam = np.ascontiguousarray(lambda_, dtype=np.float64)
    if lam.shape != (F, T):
        raise ValueError(f"lambda_ {lam.shape}, expected {(F, T)}")
    spec = _to_ctf(reverb)
    out = np.empty_like(spec)
    status = np.zeros(F, dtype=np.int32)
    _ffi.default_context().wpe_step(spec, N, T, F, taps, delay, lam, out, status=status)
    _note_rankdef(status)
    if _ffi.wpe_failed(status).any():
        raise np.linalg.LinAlgError(
            f"Singular matrix (tap correlation, {int(np.count_nonzero(_ffi.wpe_failed(status)))} bins)")
    return np.transpose(out, (2, 0, 1))


def _run_fnt(reverbs, taps, delay, context, num_iters):
    """F x N x T_u arrays through setk_wpe_batch_fnt: the reference's layout is the one the
    step kernel works in, so nothing is transposed on the host or on the device.  Returns
    (complex64 outputs, status [n][F])."""
    specs = [np.ascontiguousarray(r, dtype=np.complex64) for r in reverbs]
    F, N, _ = specs[0].shape
    if any(s.shape[0] != F or s.shape[1] != N for s in specs):
        raise ValueError("wpe_batch needs the same channel and bin count in every utterance")
    outs = [np.empty_like(s) for s in specs]
    status = np.zeros((len(specs), F), dtype=np.int32)
    _ffi.default_context().wpe_batch_fnt(specs, N, [s.shape[2] for s in specs], F, taps, delay,
                                         context, num_iters, outs, status=status)
    return outs, status


def wpe(reverb, taps=10, delay=3, context=1, num_iters=3):
    """GWPE.  reverb F x N x T complex -> dereverb F x N x T (complex128 like the
    reference's promoted result; computed in fp64, stored as complex64 between
    iterations)."""
    F, N, T = reverb.shape
    logger.info(f"WPE: F = {F}, N = {N}, T = {T}")
    outs, status = _run_fnt([reverb], taps, delay, context, num_iters)
    _note_rankdef(status)
    if _ffi.wpe_failed(status).any():
        raise np.linalg.LinAlgError(
            f"Singular matrix (tap correlation, {int(np.count_nonzero(_ffi.wpe_failed(status)))} bins)")
    return outs[0].astype(np.complex128)


def wpe_batch(reverbs, taps=10, delay=3, context=1, num_iters=3, dtype=np.complex64):
    """wpe() for a list of utterances with the same channel count (a setk_amd extension: the
    reference has no batched entry): one launch per iteration over every (bin, utterance).
    reverbs: F x N x T_u arrays -> list of dereverberated F x N x T_u arrays; an utterance
    whose tap correlation is singular comes back as None (the CLI skips it like the
    reference's LinAlgError).  The results are complex64 -- what the device computes and
    stores between iterations -- unless `dtype=np.complex128` asks for the reference's promoted
    type: that widening alone (20 MB per 4-ch 10 s utterance, on the host) used to cost more
    than the kernels and made the batched call slower per utterance than wpe() (round-3
    review).  Samples-in / samples-out users should take engine.BatchDereverb, which keeps the
    spectra on the device."""
    if not len(reverbs):
        return []
    outs, status = _run_fnt(reverbs, taps, delay, context, num_iters)
    _note_rankdef(status)
    bad = _ffi.wpe_failed(status)
    return [None if bad[u].any() else (outs[u] if np.dtype(dtype) == np.complex64 else outs[u].astype(dtype))
            for u in range(len(outs))]


def facted_wpd(obs, cgmm_iters=10, wpd_iters=3, taps=10, delay=3, context=1, update_alpha=False):
    """Joint dereverberation & denoising (factorised WPD).  obs N x T x F ->
    (tf_mask T x F x 2, wpd_enh T x F)."""
    ctx = _ffi.default_context()
    spec = np.ascontiguousarray(obs, dtype=np.complex64)  # [C][T][F]
    N, T, F = spec.shape
    logger.info(f"Facted WPD: F = {F}, N = {N}, T = {T}")
    enh = None
    gamma = np.empty((2, T, F), dtype=np.float32)
    mask = np.empty((T, F), dtype=np.float32)
    for i in range(wpd_iters):
        logger.info(f"Facted WPD: iter = {i + 1}/{wpd_iters}...")
        logger.info("Facted WPD: perform wpe...")
        der, inv_lam = _run(spec, taps, delay, context, 1, lambda_enh=enh, want_inv_lambda=True)
        logger.info("Facted WPD: mask estimation...")
        ctx.cgmm_masks(der, N, T, F, cgmm_iters, None, gamma, mask, update_alpha=bool(update_alpha))
        logger.info("Facted WPD: perform weighted mvdr...")
        Rd = np.empty((F, N, N), dtype=np.complex64)
        Rs = np.empty((F, N, N), dtype=np.complex64)
        # power-weighted covariance: the mask 1 / lambda gives Rd up to the per-bin
        # scale sum(1 / lambda) / T, which cancels in w = Rd^-1 d / (d^H Rd^-1 d)
        ctx.covar(der, inv_lam, N, T, F, Rd)
        ctx.covar(der, mask, N, T, F, Rs)
        w = np.empty((F, N), dtype=np.complex64)
        status = np.zeros(F, dtype=np.int32)
        ctx.weights(_ffi.BfOpts(kind=_ffi.BF_MVDR), Rs, Rd, None, F, N, w, status)
        if status.any():
            raise np.linalg.LinAlgError("Singular matrix (power-weighted covariance)")
        enh = np.empty((T, F), dtype=np.complex64)
        ctx.beamform(w, der, N, T, F, enh)
    tf_mask = np.transpose(gamma, (1, 2, 0)).astype(np.float64)  # T x F x K
    return tf_mask, enh.astype(np.complex128)
