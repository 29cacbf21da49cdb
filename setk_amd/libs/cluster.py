"""
Mirror of the CGMM part of scripts/sptk/libs/cluster.py (CgmmTrainer, :396-465): K = 2
classes with the deterministic start or an initial mask (what estimate_cgmm_masks.py uses by
default: the tuned kernels of csrc/cgmm_bin.hip / cgmm.hip behind setk_cgmm_masks), and
num_classes 3 or 4 with the reference's random start (csrc/cgmm_k.hip behind
setk_cgmm_masks_k, which also serves K = 2 on arrays of 9 - 16 channels); alpha fixed at 1 / K
or re-estimated (update_alpha).  The EM iterations run in libsetk_hip.so.

The K > 2 start is the reference's (cluster.py:429-434): gamma = np.random.uniform(size=[K, F,
T]) normalised over K, drawn from numpy's legacy GLOBAL generator -- which
estimate_cgmm_masks.py:28 seeds with --seed (777) once per run -- so the draw is made here,
on the host, from that same generator, in the same order (one draw per utterance): a run with
the same seed and the same table starts every utterance exactly as the reference does.

permu_aligner (:48-91, --solve-permu) is host post-processing of the K x T x F
posteriors, as in the reference: a per-bin assignment of classes to the running
centroids of overlapping bands (a few milliseconds of numpy per utterance).

Not mirrored (out of this path's scope): CACGMM, resuming from a pickled model.
"""
import numpy as np

from .. import _ffi
from .utils import EPSILON, get_logger

logger = get_logger(__name__)

__all__ = ["CgmmTrainer", "permu_aligner"]


class CgmmTrainer(object):
    """obs: M x F x T complex STFT, gamma: optional initial speech mask F x T.
    train(num_iters) -> posteriors K x F x T (float64 like the reference)."""

    def __init__(self, obs, num_classes, gamma=None, cgmm=None, update_alpha=False):
        if not 2 <= int(num_classes) <= 4:
            raise _ffi.SetkUnsupported(f"the device CGMM implements 2 <= num_classes <= 4, got {num_classes}")
        if cgmm is not None:
            raise _ffi.SetkUnsupported("resuming from a pickled cgmm model is not implemented")
        self.update_alpha = bool(update_alpha)
        self.num_classes = int(num_classes)
        M, F, T = obs.shape
        logger.info(f"CGMM instance: F = {F:d}, T = {T:}, M = {M}")
        self.shape = (M, F, T)
        self.spec = np.ascontiguousarray(np.transpose(obs, (0, 2, 1)), dtype=np.complex64)
        self.init = None    # K = 2: initial speech mask T x F
        self.gamma0 = None  # full start K x F x T (float64)
        if self.num_classes == 2:
            if gamma is not None:
                gamma = np.asarray(gamma)
                if gamma.shape != (F, T):
                    raise ValueError(f"initial mask must be F x T, got {gamma.shape}")
                self.init = np.ascontiguousarray(gamma.T, dtype=np.float32)
        elif gamma is not None:
            gamma = np.asarray(gamma, dtype=np.float64)
            if gamma.shape != (self.num_classes, F, T):
                raise ValueError(f"initial gamma must be K x F x T, got {gamma.shape}")
            self.gamma0 = np.ascontiguousarray(gamma)
        else:
            # cluster.py:429-434, from numpy's legacy global generator (seeded by the CLI)
            g = np.random.uniform(size=[self.num_classes, F, T])
            self.gamma0 = np.ascontiguousarray(g / np.sum(g, 0, keepdims=True))
            logger.info(f"Random initialized, num_classes = {self.num_classes}")
        self.gamma = None

    def train(self, num_iters):
        M, F, T = self.shape
        K = self.num_classes
        ctx = _ffi.default_context()
        if not np.isfinite(self.spec.view(np.float32)).all():
            # the reference's np.linalg.eigh raises on a covariance with NaN / inf
            # (cluster.py:104-113; uncaught by estimate_cgmm_masks.py: the run ends).  The device
            # EM's floors (max(q, eps), max(lambda, eps)) would swallow a NaN into finite
            # posteriors, so the samples are checked here
            raise np.linalg.LinAlgError("Eigenvalues did not converge (non-finite spectrogram)")
        gamma = np.empty((K, T, F), dtype=np.float32)
        if K == 2 and M <= 8:
            mask = np.empty((T, F), dtype=np.float32)
            ctx.cgmm_masks(self.spec, M, T, F, num_iters, self.init, gamma, mask,
                           update_alpha=self.update_alpha)
        else:
            status = np.zeros(F, dtype=np.int32)
            ctx.cgmm_masks_k(self.spec, M, T, F, K, num_iters, self.gamma0, self.init, gamma,
                             update_alpha=self.update_alpha, status=status)
            if status.any():
                # what np.linalg.eigh refuses (cluster.py:104-113): a covariance with NaN / inf, or
                # an eigendecomposition that did not converge
                raise np.linalg.LinAlgError(f"Eigenvalues did not converge ({int(np.count_nonzero(status))} "
                                            "frequency bins)")
        if not np.isfinite(gamma).all():
            # the reference's np.linalg.eigh raises on a covariance with NaN / inf
            # (cluster.py:104-113) -- a model that broke down; the tuned K = 2 EM has no status
            # word for it, its posteriors say it
            bad = int(np.count_nonzero(~np.isfinite(gamma).all(axis=(0, 1))))
            raise np.linalg.LinAlgError(f"Eigenvalues did not converge ({bad} frequency bins with "
                                        "non-finite posteriors)")
        self.gamma = np.transpose(gamma, (0, 2, 1)).astype(np.float64)
        return self.gamma


# bands of the aligner: (sweeps, first bin, last bin + 1), reference cluster.py:28-36
_ALIGN_PLAN = {
    257: ((20, 70, 170), (2, 90, 190), (2, 50, 150), (2, 110, 210), (2, 30, 130), (2, 130, 230),
          (2, 0, 110), (2, 150, 257)),
    513: ((20, 100, 200), (2, 120, 220), (2, 80, 180), (2, 140, 240), (2, 60, 160),
          (2, 160, 260), (2, 40, 140), (2, 180, 280), (2, 0, 120), (2, 200, 300), (2, 220, 320),
          (2, 240, 340), (2, 260, 360), (2, 280, 380), (2, 300, 400), (2, 320, 420),
          (2, 340, 440), (2, 360, 460), (2, 380, 480), (2, 400, 513)),
}


def _unit(mat, axis):
    return mat / np.maximum(np.linalg.norm(mat, axis=axis, keepdims=True), EPSILON)


def permu_aligner(masks, transpose=False):
    """Class permutation alignment over frequency (reference :48-91): masks K x T x F
    (K x F x T with transpose) -> aligned K x T x F.  Within each band of the plan the
    classes of a bin are re-assigned to the band's mean time profiles (cosine score,
    optimal assignment) until a sweep changes nothing."""
    from scipy.optimize import linear_sum_assignment
    masks = np.asarray(masks)
    if masks.ndim != 3:
        raise RuntimeError("Expect 3D TF-masks, K x T x F or K x F x T")
    if transpose:
        masks = np.transpose(masks, (0, 2, 1))
    K, _, F = masks.shape
    if F not in _ALIGN_PLAN:
        raise ValueError(f"Unsupported num_bins: {F}")
    profile = _unit(masks, 1)                       # unit time profiles, K x T x F
    owner = np.repeat(np.arange(K)[:, None], F, 1)  # class now sitting in slot k of bin f
    identity = np.arange(K)
    for sweeps, lo, hi in _ALIGN_PLAN[F]:
        for _ in range(sweeps):
            centroid = _unit(profile[..., lo:hi].mean(-1), -1)
            moved = False
            for f in range(lo, hi):
                score = centroid @ _unit(profile[..., f], -1).T
                _, pick = linear_sum_assignment(score, maximize=True)
                if np.any(pick != identity):
                    profile[..., f] = profile[pick, :, f]
                    owner[:, f] = owner[pick, f]
                    moved = True
            if not moved:
                break
    return np.take_along_axis(masks, owner[:, None, :], axis=0)
