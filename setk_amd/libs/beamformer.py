"""
Mirror of the mask-based (supervised) part of scripts/sptk/libs/beamformer.py.
Same functions, classes, argument meaning, array layouts and error behaviour:

    obs      N x F x T complex   (N: microphones, F: bins, T: frames)
    tf_mask  T x F real
    covar    F x N x N complex
    weight   F x N complex
    output   F x T complex

All arithmetic runs in libsetk_hip.so (setk_covar / setk_pevd / setk_weights /
setk_ban / setk_rank1 / setk_beamform); numerical failures reported by the
device (noise covariance not positive definite, ...) are raised as
numpy.linalg.LinAlgError exactly where the reference's LAPACK calls would.

Eigenvector gauge: the reference inherits a LAPACK dependent per-bin sign from
eigh / eigh(A, B).  Here every principal eigenvector has component 0 real and
non-negative (for the pencil the rule applies to L^H v with Rn = L L^H), see
DESIGN.md "Gauge".

Geometry based beamformers (reference :133-212, 343-512): the delay-and-sum and
superdirective WEIGHTS are F x N closed forms of the array geometry and the
direction of arrival (a steer vector, for SD one N x N solve per bin against the
diffuse-field coherence) -- a few kilobytes of float64 host arithmetic per
direction, kept on the host in the reference's own formulas; applying them to an
utterance is the device beamformer (setk_beamform / setk_apply_weights_batch).
"""
import os

import numpy as np

from .. import _ffi
from .utils import EPSILON

__all__ = [
    "compute_covar", "solve_pevd", "do_ban", "rank1_constraint", "Beamformer", "FixedBeamformer",
    "SupervisedBeamformer", "MvdrBeamformer", "MpdrBeamformer", "PmwfBeamformer",
    "GevdBeamformer", "OnlineSupervisedBeamformer", "OnlineMvdrBeamformer",
    "OnlineGevdBeamformer", "diffuse_covar", "plane_steer_vector", "linear_steer_vector",
    "circular_steer_vector", "DSBeamformer", "LinearDSBeamformer", "CircularDSBeamformer",
    "LinearSDBeamformer", "CircularSDBeamformer"
]

_STATUS_TEXT = {
    _ffi.NUM_SINGULAR: "Singular matrix",
    _ffi.NUM_NOCONV: "Eigenvalues did not converge",
    _ffi.NUM_NONFINITE: "Array must not contain infs or NaNs",
}


def _ctx():
    return _ffi.default_context()


def _c64(x):
    return np.ascontiguousarray(x, dtype=np.complex64)


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def _time_major(obs):
    """N x F x T (reference layout) -> [N][T][F] contiguous complex64."""
    return np.ascontiguousarray(np.transpose(obs, (0, 2, 1)), dtype=np.complex64)


def _raise_on_status(status):
    bad = status[status != 0]
    if bad.size:
        code = int(bad.max())
        raise np.linalg.LinAlgError(
            f"{_STATUS_TEXT.get(code, 'numerical failure')} "
            f"(frequency bins {np.flatnonzero(status)[:8].tolist()} ...)")


def do_ban(weight, Rn):
    """Blind Analytical Normalization (reference beamformer.py:14-28).
    weight F x N, Rn F x N x N -> F x N."""
    F, N = weight.shape
    out = np.empty((F, N), dtype=np.complex64)
    _ctx().ban(_c64(weight), _c64(Rn), F, N, out)
    return out.astype(np.result_type(weight.dtype, np.complex64))


def solve_pevd(Rs, Rn=None):
    """Principal eigenvector of the covariance matrix (pair), F x N
    (reference beamformer.py:31-63; complex128 for the generalised problem)."""
    F, N, _ = Rs.shape
    out = np.empty((F, N), dtype=np.complex64)
    status = np.zeros(F, dtype=np.int32)
    _ctx().pevd(_c64(Rs), None if Rn is None else _c64(Rn), F, N, 0, out, status)
    _raise_on_status(status)
    return out if Rn is None else out.astype(np.complex128)


def rank1_constraint(Rs, Rn=None):
    """(Generalised) rank-1 approximation of Rs, F x N x N
    (reference beamformer.py:66-84)."""
    F, N, _ = Rs.shape
    out = np.empty((F, N, N), dtype=np.complex64)
    status = np.zeros(F, dtype=np.int32)
    _ctx().rank1(_c64(Rs), None if Rn is None else _c64(Rn), F, N, out, status)
    _raise_on_status(status)
    return out if Rn is None else out.astype(np.complex128)


def compute_covar(obs, tf_mask):
    """covar[f] = sum_t m[t, f] x x^H / max(sum_t m[t, f], 1e-6)
    (reference beamformer.py:87-103).  obs N x F x T, tf_mask T x F."""
    N, F, T = obs.shape
    out = np.empty((F, N, N), dtype=np.complex64)
    _ctx().covar(_time_major(obs), _f32(tf_mask), N, T, F, out)
    wide = obs.dtype == np.complex128 or np.asarray(tf_mask).dtype == np.float64
    return out.astype(np.complex128) if wide else out


class Beamformer(object):
    def __init__(self):
        pass

    def beamform(self, weight, obs):
        """out[f, t] = sum_n conj(weight[f, n]) obs[n, f, t]
        (reference beamformer.py:220-234)."""
        if weight.shape[0] != obs.shape[1] or weight.shape[1] != obs.shape[0]:
            raise ValueError("Input obs do not match with weight, " +
                             f"{weight.shape} vs {obs.shape}")
        N, F, T = obs.shape
        out = np.empty((T, F), dtype=np.complex64)
        _ctx().beamform(_c64(weight), _time_major(obs), N, T, F, out)
        enh = out.T  # F x T view
        wide = weight.dtype == np.complex128 or obs.dtype == np.complex128
        return enh.astype(np.complex128) if wide else enh


class FixedBeamformer(Beamformer):
    """Beamformer with predefined weights F x N (reference :323-340)."""

    def __init__(self, weight):
        super(FixedBeamformer, self).__init__()
        self.weight = weight

    def run(self, obs):
        return self.beamform(self.weight, obs)


# ---- array geometry: steer vectors and the diffuse-field coherence (host, float64) ----
def diffuse_covar(num_bins, dist_mat, sr=16000, c=340, diag_eps=0.1):
    """Coherence of the spherically isotropic noise field, F x N x N:
    sinc(2 f d_ij / c) + diag_eps I (reference :133-151; numpy's normalised sinc)."""
    dist_mat = np.asarray(dist_mat, dtype=np.float64)
    N = dist_mat.shape[0]
    omega = np.pi * np.arange(num_bins) * sr / (num_bins - 1)
    return np.sinc(dist_mat[None] * omega[:, None, None] / c) + np.eye(N) * diag_eps


def plane_steer_vector(distance, num_bins, c=340, sr=16000):
    """exp(-j omega d_n / c), F x N, for distances projected on the DoA (:154-165)."""
    omega = np.pi * np.arange(num_bins) * sr / (num_bins - 1)
    return np.exp(-1j * np.outer(omega, np.asarray(distance, dtype=np.float64) / c))


def linear_steer_vector(topo, doa, num_bins, c=340, sr=16000):
    """Linear array at positions topo (m), DoA in degrees, 0 = end-fire (:168-184)."""
    dist = np.cos(doa * np.pi / 180) * np.asarray(topo, dtype=np.float64)
    return plane_steer_vector(dist, num_bins, c=c, sr=sr)


def circular_steer_vector(redius, num_arounded, doa, num_bins, c=349, sr=16000, center=False):
    """Uniform circular array (optionally with a centre microphone first) (:187-212)."""
    dirc = np.arange(num_arounded) * 2 * np.pi / num_arounded
    dist = np.cos(dirc - doa * np.pi / 180) * redius
    if center:
        dist = np.concatenate([np.array([0]), dist])
    return plane_steer_vector(-dist, num_bins, c=c, sr=sr)


def _superdirective(steer_vector, Rn):
    """w = Rn^-1 d / (d^H Rn^-1 d) per bin (:454-460, 505-511)."""
    numerator = np.linalg.solve(Rn, steer_vector[..., None])[..., 0]
    denominator = np.einsum("...d,...d->...", steer_vector.conj(), numerator)
    return numerator / denominator[..., None]


class DSBeamformer(Beamformer):
    """Base of the geometry beamformers: weight(doa, num_bins) on the host, the
    beamforming on the device (reference :343-374)."""

    def __init__(self, num_mics):
        super(DSBeamformer, self).__init__()
        self.num_mics = num_mics

    def weight(self, doa, num_bins, c=340, sr=16000):
        raise NotImplementedError

    def run(self, doa, obs, c=340, sr=16000):
        if obs.shape[0] != self.num_mics:
            raise ValueError("Shape of obs do not match with number" +
                             f"of microphones, {self.num_mics} vs {obs.shape[0]}")
        weight = self.weight(doa, obs.shape[1], c=c, sr=sr)
        return self.beamform(weight, obs)


class LinearDSBeamformer(DSBeamformer):
    """Delay and sum, linear array (:377-396)."""

    def __init__(self, linear_topo):
        super(LinearDSBeamformer, self).__init__(len(linear_topo))
        self.linear_topo = np.array(linear_topo)

    def weight(self, doa, num_bins, c=340, sr=16000):
        return linear_steer_vector(self.linear_topo, doa, num_bins, c=c, sr=sr) / self.num_mics


class CircularDSBeamformer(DSBeamformer):
    """Delay and sum, circular array (:399-427)."""

    def __init__(self, radius, num_arounded, center=False):
        super(CircularDSBeamformer, self).__init__(num_arounded + 1 if center else num_arounded)
        self.radius = radius
        self.center = center
        self.num_arounded = num_arounded

    def weight(self, doa, num_bins, c=340, sr=16000):
        sv = circular_steer_vector(self.radius, self.num_arounded, doa, num_bins, c=c, sr=sr,
                                   center=self.center)
        return sv / self.num_mics


class LinearSDBeamformer(LinearDSBeamformer):
    """Superdirective beamformer in a diffuse noise field, linear array (:430-460)."""

    def __init__(self, linear_topo):
        super(LinearSDBeamformer, self).__init__(linear_topo)
        mat = np.tile(self.linear_topo, (self.num_mics, 1))
        self.distance_mat = np.abs(mat - np.transpose(mat))

    def weight(self, doa, num_bins, c=340, sr=16000, diag_eps=0.1):
        sv = super(LinearSDBeamformer, self).weight(doa, num_bins, c=c, sr=sr)
        Rn = diffuse_covar(num_bins, self.distance_mat, sr=sr, c=c, diag_eps=diag_eps)
        return _superdirective(sv, Rn)


class CircularSDBeamformer(CircularDSBeamformer):
    """Superdirective beamformer, circular array (:463-511)."""

    def __init__(self, radius, num_arounded, center=False):
        super(CircularSDBeamformer, self).__init__(radius, num_arounded, center=center)
        self.distance_mat = self._compute_distance_mat()

    def _compute_distance_mat(self):
        distance_mat = np.zeros((self.num_mics, self.num_mics))
        if self.center:
            distance_mat[0, 1:] = self.radius
            raw = 1
        else:
            raw = 0
        ang = np.pi / self.num_arounded
        for r in range(raw, self.num_mics):
            for c in range(r + 1, self.num_mics):
                distance_mat[r, c] = np.abs(np.sin((c - r) * ang) * 2 * self.radius)
        distance_mat += distance_mat.T
        return distance_mat

    def weight(self, doa, num_bins, c=340, sr=16000, diag_eps=1e-5):
        sv = super(CircularSDBeamformer, self).weight(doa, num_bins, c=c, sr=sr)
        Rn = diffuse_covar(num_bins, self.distance_mat, sr=sr, c=c, diag_eps=diag_eps)
        return _superdirective(sv, Rn)


class SupervisedBeamformer(Beamformer):
    """Base class of the TF-mask based beamformers (reference :237-283).

    strict_reference (class or instance attribute, default from SETK_STRICT_REFERENCE=1, else
    False): raise LinAlgError("Singular matrix") exactly where the reference's
    numpy.linalg.solve does (SETK_FLAG_STRICT_REFERENCE, include/setk_hip.h); by default a
    covariance that is singular but not all-zero is regularised and solved."""
    _kind = None
    _wide = False  # reference dtype of the weights (complex128 for GEV)
    strict_reference = os.environ.get("SETK_STRICT_REFERENCE", "0") not in ("", "0", "false")

    def _mode_flags(self, ban):
        return (_ffi.FLAG_BAN if ban else 0) | \
            (_ffi.FLAG_STRICT_REFERENCE if self.strict_reference else 0)

    def __init__(self, num_bins):
        super(SupervisedBeamformer, self).__init__()
        self.num_bins = num_bins

    def compute_covar_mat(self, target_mask, obs):
        if target_mask.shape[1] != self.num_bins or target_mask.ndim != 2:
            raise ValueError("Input mask matrix should be shape as " +
                             f"[num_frames x num_bins], now is {target_mask.shape}")
        if obs.shape[1] != target_mask.shape[1] or obs.shape[2] != target_mask.shape[0]:
            raise ValueError("Shape of input obs do not match with " +
                             f"mask matrix, {obs.shape} vs {target_mask.shape}")
        return compute_covar(obs, target_mask)

    def _opts(self, ban=False):
        return _ffi.BfOpts(kind=self._kind, flags=self._mode_flags(ban), pmwf_beta=0.0,
                           pmwf_ref=-1, rank1=0)

    def _device_weight(self, Rs, Rn, Ry=None, ban=False):
        F, N, _ = Rs.shape
        out = np.empty((F, N), dtype=np.complex64)
        status = np.zeros(F, dtype=np.int32)
        _ctx().weights(self._opts(ban), _c64(Rs), None if Rn is None else _c64(Rn),
                       None if Ry is None else _c64(Ry), F, N, out, status)
        _raise_on_status(status)
        return out.astype(np.complex128) if self._wide else out

    def weight(self, Rs, Rn):
        raise NotImplementedError

    def run(self, mask_s, obs, mask_n=None, ban=False):
        """mask_s T x F, obs N x F x T -> enhanced F x T (reference :270-283)."""
        Rn = self.compute_covar_mat(1 - mask_s if mask_n is None else mask_n, obs)
        Rs = self.compute_covar_mat(mask_s, obs)
        weight = self._device_weight(Rs, Rn, ban=ban) if self._kind is not None else (
            do_ban(self.weight(Rs, Rn), Rn) if ban else self.weight(Rs, Rn))
        return self.beamform(weight, obs)


class MvdrBeamformer(SupervisedBeamformer):
    """h = Rn^-1 d / (d^H Rn^-1 d), d = P(Rs)  (reference :515-539)."""
    _kind = _ffi.BF_MVDR

    def weight(self, Rs, Rn):
        return self._device_weight(Rs, Rn)


class GevdBeamformer(SupervisedBeamformer):
    """h = P(Rs, Rn), the max-SNR beamformer (reference :662-682)."""
    _kind = _ffi.BF_GEVD
    _wide = True

    def weight(self, Rs, Rn):
        return self._device_weight(Rs, Rn)


class PmwfBeamformer(SupervisedBeamformer):
    """Parameterized multichannel Wiener filter (reference :593-659):
    W = Rn^-1 Rs / (beta + tr(Rn^-1 Rs)), column ref_channel (or the channel of
    maximum estimated SNR when ref_channel < 0)."""
    _kind = _ffi.BF_PMWF

    def __init__(self, num_bins, beta=0, ref_channel=-1, rank1_appro=""):
        super(PmwfBeamformer, self).__init__(num_bins)
        self.ref_channel = ref_channel
        self.rank1_appro = rank1_appro
        self.beta = beta

    def _opts(self, ban=False):
        rank1 = {"eig": _ffi.RANK1_EIG, "gev": _ffi.RANK1_GEV}.get(self.rank1_appro,
                                                                   _ffi.RANK1_NONE)
        return _ffi.BfOpts(kind=self._kind, flags=self._mode_flags(ban),
                           pmwf_beta=float(self.beta), pmwf_ref=int(self.ref_channel),
                           rank1=rank1)

    def weight(self, Rs, Rn):
        _, N, _ = Rs.shape
        if self.ref_channel >= N:
            raise RuntimeError("Reference channel ID exceeds total " +
                               f"channels: {self.ref_channel} vs {N}")
        self._wide = self.rank1_appro == "gev"
        return self._device_weight(Rs, Rn)

    def run(self, mask_s, obs, mask_n=None, ban=False):
        if self.ref_channel >= obs.shape[0]:
            raise RuntimeError("Reference channel ID exceeds total " +
                               f"channels: {self.ref_channel} vs {obs.shape[0]}")
        self._wide = self.rank1_appro == "gev"
        return super(PmwfBeamformer, self).run(mask_s, obs, mask_n=mask_n, ban=ban)


class MpdrBeamformer(SupervisedBeamformer):
    """h = Ry^-1 d / (d^H Ry^-1 d) (reference :542-590); d = P(Rs) or, with
    whiten, Rn P(Rs, Rn)."""

    def __init__(self, num_bins, whiten=False):
        super(MpdrBeamformer, self).__init__(num_bins)
        self.whiten = whiten
        self._kind = _ffi.BF_MPDR_WHITEN if whiten else _ffi.BF_MPDR
        self._wide = bool(whiten)

    def weight(self, Rs, Ry, Rn=None):
        kind = _ffi.BF_MPDR if Rn is None else _ffi.BF_MPDR_WHITEN
        saved = self._kind, self._wide
        self._kind, self._wide = kind, Rn is not None
        try:
            return self._device_weight(Rs, Rn, Ry=Ry)
        finally:
            self._kind, self._wide = saved

    def run(self, mask_s, obs, mask_n=None, ban=False):
        Rs = self.compute_covar_mat(mask_s, obs)
        Ry = self.compute_covar_mat(np.ones_like(mask_s), obs)
        Rn = None
        if self.whiten:
            Rn = self.compute_covar_mat(1 - mask_s if mask_n is None else mask_n, obs)
        elif ban:
            # the reference reaches `do_ban(weight, Rn)` with Rn undefined here
            # (beamformer.py:590, NameError); raise a diagnosable error instead
            raise ValueError("BAN needs a noise covariance: use MpdrBeamformer(whiten=True)")
        weight = self.weight(Rs, Ry, Rn=Rn)
        return self.beamform(do_ban(weight, Rn) if ban else weight, obs)


class OnlineSupervisedBeamformer(SupervisedBeamformer):
    """Block-online variant with recursive covariance smoothing (reference
    :286-320).  NB: the reference CLI calls run(..., normalize=) and fails with
    a TypeError (apply_adaptive_beamformer.py:42-45); the keyword here is ban=."""

    def __init__(self, num_bins, num_channels, alpha=0.8):
        super(OnlineSupervisedBeamformer, self).__init__(num_bins)
        self.covar_mat_shape = (num_bins, num_channels, num_channels)
        self.reset_stats(alpha=alpha)

    def reset_stats(self, alpha=0.8):
        self.Rs = np.zeros(self.covar_mat_shape, dtype=np.complex128)
        self.Rn = np.zeros(self.covar_mat_shape, dtype=np.complex128)
        self.alpha = alpha
        self.reset = True

    def run(self, mask_s, obs, mask_n=None, ban=False):
        Rn = self.compute_covar_mat(1 - mask_s if mask_n is None else mask_n, obs)
        Rs = self.compute_covar_mat(mask_s, obs)
        phi = 1 if self.reset else (1 - self.alpha)
        self.Rs = self.Rs * self.alpha + phi * Rs
        self.Rn = self.Rn * self.alpha + phi * Rn
        self.reset = False
        weight = self._device_weight(self.Rs, self.Rn)
        return self.beamform(do_ban(weight, Rn) if ban else weight, obs)


class OnlineMvdrBeamformer(OnlineSupervisedBeamformer):
    _kind = _ffi.BF_MVDR


class OnlineGevdBeamformer(OnlineSupervisedBeamformer):
    _kind = _ffi.BF_GEVD
    _wide = True
