"""
Mirror of the CGMM part of scripts/sptk/libs/cluster.py (CgmmTrainer,
:396-465) for the configuration estimate_cgmm_masks.py uses by default:
K = 2 classes, alpha fixed at 1/2 or re-estimated (update_alpha), deterministic
start or an initial mask.
The EM iterations run in libsetk_hip.so (setk_cgmm_masks, csrc/cgmm.hip).

Not mirrored (out of this path's scope): K > 2 (random initialisation
from numpy's unseeded global generator: not reproducible in the reference
either), the permutation aligner that only matters for K > 2, CACGMM.
"""
import numpy as np

from .. import _ffi
from .utils import EPSILON, get_logger

logger = get_logger(__name__)

__all__ = ["CgmmTrainer"]


class CgmmTrainer(object):
    """obs: M x F x T complex STFT, gamma: optional initial speech mask F x T.
    train(num_iters) -> posteriors K x F x T (float64 like the reference)."""

    def __init__(self, obs, num_classes, gamma=None, cgmm=None, update_alpha=False):
        if num_classes != 2:
            raise _ffi.SetkUnsupported("the device CGMM implements num_classes = 2")
        if cgmm is not None:
            raise _ffi.SetkUnsupported("resuming from a pickled cgmm model is not implemented")
        self.update_alpha = bool(update_alpha)
        M, F, T = obs.shape
        logger.info(f"CGMM instance: F = {F:d}, T = {T:}, M = {M}")
        self.shape = (M, F, T)
        self.spec = np.ascontiguousarray(np.transpose(obs, (0, 2, 1)), dtype=np.complex64)
        self.init = None
        if gamma is not None:
            gamma = np.asarray(gamma)
            if gamma.shape != (F, T):
                raise ValueError(f"initial mask must be F x T, got {gamma.shape}")
            self.init = np.ascontiguousarray(gamma.T, dtype=np.float32)
        self.gamma = None

    def train(self, num_iters):
        M, F, T = self.shape
        gamma = np.empty((2, T, F), dtype=np.float32)
        mask = np.empty((T, F), dtype=np.float32)
        _ffi.default_context().cgmm_masks(self.spec, M, T, F, num_iters, self.init, gamma, mask,
                                          update_alpha=self.update_alpha)
        self.gamma = np.transpose(gamma, (0, 2, 1)).astype(np.float64)
        return self.gamma
