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

__all__ = ["wpe", "wpe_step", "compute_tap_mat", "compute_lambda", "facted_wpd"]


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
    if status.any():
        raise np.linalg.LinAlgError(
            f"Singular matrix (tap correlation, {int(np.count_nonzero(status))} bins)")
    return out, inv


def wpe_step(reverb, yt, lambda_):
    """One WPE step with caller-supplied variances.  reverb F x N x T, yt F x NK x T
    (only its shape is used: taps = NK / N, delay from the first non-zero column is
    not recoverable, so yt must come from compute_tap_mat and the delay is inferred
    from its leading zero frames), lambda_ F x T."""
    F, N, T = reverb.shape
    taps = yt.shape[1] // N
    # delay = number of leading all-zero frames of the first tap block
    nz = np.flatnonzero(np.any(yt[:, :N, :] != 0, axis=(0, 1)))
    delay = int(nz[0]) if nz.size else 0
    enh = np.ascontiguousarray(np.sqrt(np.asarray(lambda_, dtype=np.float64)).T.astype(np.complex64))
    out, _ = _run(_to_ctf(reverb), taps, delay, 0, 1, lambda_enh=enh)
    return np.transpose(out, (2, 0, 1))


def wpe(reverb, taps=10, delay=3, context=1, num_iters=3):
    """GWPE.  reverb F x N x T complex -> dereverb F x N x T (complex128 like the
    reference's promoted result; computed in fp64, stored as complex64 between
    iterations)."""
    F, N, T = reverb.shape
    logger.info(f"WPE: F = {F}, N = {N}, T = {T}")
    out, _ = _run(_to_ctf(reverb), taps, delay, context, num_iters)
    return np.transpose(out, (2, 0, 1)).astype(np.complex128)


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
