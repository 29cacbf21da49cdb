"""
CPU oracle for the mask-based adaptive-beamformer hot path.

*** TEST INFRASTRUCTURE -- NOT PRODUCT CODE ***
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product path (``setk_amd``) never
imports it and fails loudly when the HIP library is missing.

This is a plain numpy/scipy *restatement* of the algorithm the reference
executes on its hot path.  Every function cites the reference lines it follows
(paths relative to the funcwj/setk tree):

    STFT / iSTFT wrappers     scripts/sptk/libs/utils.py:96-173
    (librosa 0.8.1 stft/istft  -- third-party, pinned in requirements.txt:2,
     NOT vendored in the reference; semantics restated from the published
     library behaviour, see ``librosa_stft`` / ``librosa_istft``)
    covariance                scripts/sptk/libs/beamformer.py:87-103
    principal eigenvectors    scripts/sptk/libs/beamformer.py:31-63
    BAN                       scripts/sptk/libs/beamformer.py:14-28
    rank-1 constraint         scripts/sptk/libs/beamformer.py:66-84
    MVDR / MPDR / PMWF / GEV  scripts/sptk/libs/beamformer.py:515-682
    beamform                  scripts/sptk/libs/beamformer.py:220-234
    CLI compute loop          scripts/sptk/apply_adaptive_beamformer.py:130-178
    CGMM mask estimation      scripts/sptk/libs/cluster.py:94-287,396-465

Parity pinning: ``tests/test_oracle_golden.py`` checks this restatement
against (i) the reference's stored doc outputs
(doc/adaptive_beamformer/asset/*.wav, committed as fixtures under
tests/golden/) and (ii) vectors produced by running the *unmodified* reference
modules in the build container (oracle/make_golden.py).

Eigenvector gauge: LAPACK decides a per-bin +-1 sign of
``solve_pevd`` outputs.  ``fix_gauge_evd`` / ``fix_gauge_gev`` implement the
declared policy (component 0 real and non-negative; for the generalised
problem the rule is applied to y = L^H v with Rn = L L^H).
"""
import math

import numpy as np
import scipy.linalg
import scipy.signal

EPSILON = np.finfo(np.float32).eps  # libs/utils.py:16


# ----------------------------------------------------------------------------
# librosa 0.8.1 restatement (third-party arithmetic reached from utils.py:123,159)
# ----------------------------------------------------------------------------
def nextpow2(n):
    # libs/utils.py:25-27
    return 2**math.ceil(math.log2(n))


def _pad_center(w, size):
    n = w.shape[0]
    lpad = (size - n) // 2
    out = np.zeros(size, dtype=w.dtype)
    out[lpad:lpad + n] = w
    return out


def make_window(window, win_length):
    """scipy.signal.get_window(window, win_length, fftbins=True), or a
    caller-supplied array (the reference passes an ndarray for "sqrthann",
    libs/utils.py:116-117)."""
    if isinstance(window, str):
        if window == "sqrthann":
            return scipy.signal.windows.hann(win_length, sym=False)**0.5
        return scipy.signal.get_window(window, win_length, fftbins=True)
    w = np.asarray(window)
    if w.shape[0] != win_length:
        raise ValueError("window size mismatch")
    return w


def librosa_stft(y, n_fft, hop_length, win_length=None, window="hann",
                 center=True):
    """librosa.stft (0.8.1): periodic window zero-padded & centred to n_fft,
    reflect padding of n_fft//2 when center, frames 1+(len-n_fft)//hop,
    float64 rfft, stored complex64 for float32 input.  Returns F x T."""
    if win_length is None:
        win_length = n_fft
    w = _pad_center(make_window(window, win_length).astype(np.float64), n_fft)
    y = np.asarray(y)
    if center:
        y = np.pad(y, n_fft // 2, mode="reflect")
    elif n_fft > y.shape[-1]:
        raise ValueError("n_fft larger than input")
    n_frames = 1 + (y.shape[-1] - n_fft) // hop_length
    idx = (np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :])
    frames = y[idx]  # n_fft x T
    spec = np.fft.rfft(w[:, None] * frames, axis=0)
    out_dtype = np.complex64 if y.dtype == np.float32 else np.complex128
    return np.asfortranarray(spec.astype(out_dtype))


def librosa_istft(S, hop_length, win_length=None, window="hann", center=True,
                  length=None):
    """librosa.istft (0.8.1): irfft, * padded window, overlap-add,
    / sum(window^2) where > tiny, trim n_fft//2 when center."""
    S = np.asarray(S)
    n_fft = 2 * (S.shape[0] - 1)
    if win_length is None:
        win_length = n_fft
    w = _pad_center(make_window(window, win_length).astype(np.float64), n_fft)
    if length is None:
        n_frames = S.shape[1]
    else:
        padded = length + (n_fft if center else 0)
        n_frames = min(S.shape[1], int(np.ceil(padded / hop_length)))
    dtype = np.float32 if S.dtype == np.complex64 else np.float64
    n_out = n_fft + hop_length * (n_frames - 1)
    y = np.zeros(n_out, dtype=dtype)
    ytmp = w[:, None] * np.fft.irfft(S[:, :n_frames], axis=0)
    for t in range(n_frames):
        s = t * hop_length
        y[s:s + n_fft] += ytmp[:, t]
    win_sq = _pad_center(make_window(window, win_length).astype(np.float64)**2,
                         n_fft)
    wss = np.zeros(n_out, dtype=dtype)
    for t in range(n_frames):
        s = t * hop_length
        wss[s:s + n_fft] += win_sq[:max(0, min(n_fft, n_out - s))]
    nz = wss > np.finfo(dtype).tiny
    y[nz] /= wss[nz]
    if length is None:
        if center:
            y = y[n_fft // 2:-(n_fft // 2)]
    else:
        start = n_fft // 2 if center else 0
        y = y[start:]
        if y.shape[0] > length:
            y = y[:length]
        elif y.shape[0] < length:
            y = np.pad(y, (0, length - y.shape[0]))
    return y


# ----------------------------------------------------------------------------
# libs/utils.py wrappers
# ----------------------------------------------------------------------------
def forward_stft(samps, frame_len=1024, frame_hop=256, round_power_of_two=True,
                 center=False, window="hann", apply_abs=False, apply_log=False,
                 apply_pow=False, transpose=True):
    # libs/utils.py:96-138
    if apply_log and not apply_abs:
        apply_abs = True
    if samps.ndim != 1:
        raise RuntimeError("Invalid shape, librosa.stft accepts mono input")
    n_fft = nextpow2(frame_len) if round_power_of_two else frame_len
    m = librosa_stft(samps, n_fft, frame_hop, win_length=frame_len,
                     window=window, center=center)
    if apply_abs:
        m = np.sqrt(m.real**2 + m.imag**2)
    if apply_pow:
        m = np.power(m, 2)
    if apply_log:
        m = np.log(np.maximum(m, EPSILON))
    if transpose:
        m = np.transpose(m)
    return m


def inverse_stft(stft_mat, frame_len=1024, frame_hop=256, center=False,
                 window="hann", transpose=True, norm=None, power=None,
                 nsamps=None):
    # libs/utils.py:142-173
    if transpose:
        stft_mat = np.transpose(stft_mat)
    samps = librosa_istft(stft_mat, frame_hop, win_length=frame_len,
                          window=window, center=center, length=nsamps)
    if norm:
        samps_norm = np.linalg.norm(samps, np.inf)
        samps = samps * norm / (samps_norm + EPSILON)
    if power:
        samps_pow = np.linalg.norm(samps, 2)**2 / samps.size
        samps = samps * np.sqrt(power / samps_pow)
    return samps


def multichannel_stft(samps, **stft_kwargs):
    """SpectrogramReader._load, libs/data_handler.py:492-503 -> N x F x T."""
    if samps.ndim == 1:
        return forward_stft(samps, **stft_kwargs)
    samps = np.ascontiguousarray(samps)
    return np.stack([forward_stft(s, **stft_kwargs) for s in samps])


# ----------------------------------------------------------------------------
# libs/beamformer.py
# ----------------------------------------------------------------------------
def compute_covar(obs, tf_mask):
    # libs/beamformer.py:87-103 ; obs N x F x T, mask T x F -> F x N x N
    obs = np.transpose(obs, (1, 0, 2))
    mask = np.expand_dims(np.transpose(tf_mask), axis=1)
    den = np.maximum(np.sum(mask, axis=-1, keepdims=True), 1e-6)
    return np.einsum("...dt,...et->...de", mask * obs, obs.conj()) / den


def fix_gauge_evd(vec):
    """Declared gauge: component 0 real and non-negative (vec: F x N)."""
    v0 = vec[:, 0]
    mag = np.abs(v0)
    ph = np.where(mag > 0, np.conj(v0) / np.where(mag > 0, mag, 1), 1)
    return vec * ph[:, None]


def fix_gauge_gev(vec, Rn):
    """Declared gauge for the pencil (Rs, Rn): with Rn = L L^H (L lower),
    y = L^H v has a real non-negative component 0.  y0 = conj(L00)*v0 + ...
    only column 0 of L^H row 0 -> y0 = sum_k conj(L[k,0]) v[k].

    A bin whose Rn is not positive definite to float64 has no such L.  The reference does
    NOT raise there (hegvd's refusal is caught and scipy.linalg.eig answers,
    libs/beamformer.py:54-59; verified on its own recordings with the unmodified modules:
    tests/golden/ref_skipset.json) and neither may the gauge: such a bin gets the plain
    rule (component 0 of v real, non-negative) -- its vector is QZ's answer to a singular
    pencil, i.e. noise, and no comparison is made there (gev_fallback_bins)."""
    out = np.empty_like(vec)
    for f in range(vec.shape[0]):
        try:
            L = np.linalg.cholesky(Rn[f])
            y0 = np.vdot(L[:, 0], vec[f])  # (L^H v)_0
        except np.linalg.LinAlgError:
            y0 = vec[f, 0]
        mag = abs(y0)
        out[f] = vec[f] * (np.conj(y0) / mag if mag > 0 else 1)
    return out


def gev_fallback_bins(Rs, Rn):
    """Bins where scipy.linalg.eigh(Rs[f], Rn[f]) refuses (Rn not positive definite for
    LAPACK's potrf) and the reference's scipy.linalg.eig fallback answers
    (libs/beamformer.py:52-59): their vectors are rounding artefacts."""
    bad = []
    for f in range(Rs.shape[0]):
        try:
            scipy.linalg.eigh(Rs[f], Rn[f])
        except np.linalg.LinAlgError:
            bad.append(f)
    return np.asarray(bad, dtype=int)


def solve_pevd(Rs, Rn=None, gauge=False):
    # libs/beamformer.py:31-63
    if Rn is None:
        _, vecs = np.linalg.eigh(Rs)
        pvec = vecs[:, :, -1]
        return fix_gauge_evd(pvec) if gauge else pvec
    F, N, _ = Rs.shape
    pvec = np.zeros((F, N), dtype=complex)
    for f in range(F):
        try:
            _, vecs = scipy.linalg.eigh(Rs[f], Rn[f])
            pvec[f] = vecs[:, -1]
        except np.linalg.LinAlgError:
            vals, vecs = scipy.linalg.eig(Rs[f], Rn[f])
            pvec[f] = vecs[:, np.argmax(vals)]
    return fix_gauge_gev(pvec, Rn) if gauge else pvec


def do_ban(weight, Rn):
    # libs/beamformer.py:14-28
    nom = np.einsum("...a,...ab,...bc,...c->...", np.conj(weight), Rn, Rn,
                    weight)
    den = np.einsum("...a,...ab,...b->...", np.conj(weight), Rn, weight)
    filt = np.sqrt(np.sqrt(nom.real**2 + nom.imag**2)) / np.maximum(
        np.real(den), EPSILON)
    return filt[:, None] * weight


def rank1_constraint(Rs, Rn=None, gauge=False):
    # libs/beamformer.py:66-84 (gauge-free: pvec enters as v v^H)
    pvecs = solve_pevd(Rs, Rn=Rn, gauge=gauge)
    if Rn is not None:
        pvecs = np.einsum("...ab,...b->...a", Rn, pvecs)
    r1 = np.einsum("...a,...b->...ab", pvecs, pvecs.conj())
    scale = np.trace(Rs, axis1=-1, axis2=-2) / np.maximum(
        np.trace(r1, axis1=-1, axis2=-2), EPSILON)
    return scale[..., None, None] * r1


def _solve_vec(A, b):
    """np.linalg.solve with numpy<2 'stack of vectors' semantics
    (the reference relies on it at libs/beamformer.py:536)."""
    return np.linalg.solve(A, b[..., None])[..., 0]


def mvdr_weight(Rs, Rn, gauge=False):
    # libs/beamformer.py:527-539
    sv = solve_pevd(Rs, gauge=gauge)
    num = _solve_vec(Rn, sv)
    den = np.einsum("...d,...d->...", sv.conj(), num)
    return num / np.expand_dims(den, axis=-1)


def mpdr_weight(Rs, Ry, Rn=None, gauge=False):
    # libs/beamformer.py:555-571
    if Rn is None:
        sv = solve_pevd(Rs, gauge=gauge)
    else:
        gev = solve_pevd(Rs, Rn, gauge=gauge)
        sv = np.einsum("...ab,...b->...a", Rn, gev)
    num = _solve_vec(Ry, sv)
    den = np.einsum("...d,...d->...", sv.conj(), num)
    return num / np.expand_dims(den, axis=-1)


def gevd_weight(Rs, Rn, gauge=False):
    # libs/beamformer.py:674-682
    return solve_pevd(Rs, Rn, gauge=gauge)


def pmwf_weight(Rs, Rn, beta=0, ref_channel=-1, rank1_appro="", gauge=False):
    # libs/beamformer.py:632-659
    _, N, _ = Rs.shape
    if rank1_appro == "eig":
        Rs = rank1_constraint(Rs, gauge=gauge)
    if rank1_appro == "gev":
        Rs = rank1_constraint(Rs, Rn=Rn, gauge=gauge)
    num = np.linalg.solve(Rn, Rs)
    den = beta + np.trace(num, axis1=1, axis2=2)
    wmat = num / den[..., None, None]
    if ref_channel < 0:
        snr = []
        for c in range(N):
            w = wmat[..., c]
            ps = np.einsum("...fa,...fab,...fb->...", np.conj(w), Rs, w)
            pn = np.einsum("...fa,...fab,...fb->...", np.conj(w), Rn, w)
            snr.append(np.real(ps) / np.maximum(EPSILON, np.real(pn)))
        ref_channel = int(np.argmax(snr))
    if ref_channel >= N:
        raise RuntimeError("Reference channel ID exceeds total channels")
    return wmat[..., ref_channel]


def directional_feats(spectrogram, steer_vector, df_pair=None):
    """libs/spatial.py:184-208: mean over microphone pairs of
    cos((arg X_i - arg X_j) - (arg v_i - arg v_j)).  spectrogram M x F x T,
    steer_vector M x F  ->  T x F.  (A common phase of v cancels: gauge free.)"""
    M = spectrogram.shape[0]
    arg_s, arg_t = np.angle(spectrogram), np.angle(steer_vector)
    if df_pair is None:
        df_pair = [(i, j) for i in range(M) for j in range(i + 1, M)]
    df = []
    for i, j in df_pair:
        delta_s = arg_s[i] - arg_s[j]
        delta_t = (arg_t[i] - arg_t[j])[:, None]
        df.append(np.cos(delta_s - delta_t))
    return np.transpose(np.average(np.stack(df), axis=0))


def beamform(weight, obs):
    # libs/beamformer.py:220-234 ; weight F x N, obs N x F x T -> F x T
    if weight.shape[0] != obs.shape[1] or weight.shape[1] != obs.shape[0]:
        raise ValueError("Input obs do not match with weight")
    obs = np.transpose(obs, (1, 0, 2))
    return np.einsum("...n,...nt->...t", weight.conj(), obs)


# ---- geometry beamformers (libs/beamformer.py:133-212, 343-512) ----
def classic_weight(kind, geometry, doa, num_bins, c=340, sr=16000, linear_topo=(),
                   radius=0.05, num_arounded=6, circular_center=False, diag_eps=None):
    """Delay-and-sum ("ds") or superdirective ("sd") weights F x N of a linear /
    circular array for one direction of arrival (degrees).
    linear: dist_n = cos(doa) topo_n (:182); circular: dist_n = -r cos(2 pi n / N - doa),
    a centre microphone first at distance 0 (:207-212); steer vector
    exp(-j omega dist / c), omega_f = pi f sr / (F - 1) (:163-164); DS = sv / N (:396, 427);
    SD = Rn^-1 d / (d^H Rn^-1 d) with d the DS weight and Rn the diffuse coherence
    sinc(omega D / c) + diag_eps I, diag_eps 0.1 linear / 1e-5 circular (:133-151,
    438-460, 488-511)."""
    omega = np.pi * np.arange(num_bins) * sr / (num_bins - 1)
    if geometry == "linear":
        topo = np.asarray(linear_topo, dtype=np.float64)
        dist = np.cos(doa * np.pi / 180) * topo
        n = len(topo)
        mat = np.tile(topo, (n, 1))
        dmat = np.abs(mat - mat.T)
        eps = 0.1 if diag_eps is None else diag_eps
    else:
        dirc = np.arange(num_arounded) * 2 * np.pi / num_arounded
        dist = -np.cos(dirc - doa * np.pi / 180) * radius
        if circular_center:
            dist = np.concatenate([[0.0], dist])
        n = len(dist)
        dmat = np.zeros((n, n))
        raw = 0
        if circular_center:
            dmat[0, 1:] = radius
            raw = 1
        ang = np.pi / num_arounded
        for r in range(raw, n):
            for q in range(r + 1, n):
                dmat[r, q] = np.abs(np.sin((q - r) * ang) * 2 * radius)
        dmat += dmat.T
        eps = 1e-5 if diag_eps is None else diag_eps
    ds = np.exp(-1j * np.outer(omega, dist / c)) / n
    if kind == "ds":
        return ds
    Rn = np.sinc(dmat[None] * omega[:, None, None] / c) + np.eye(n) * eps
    num = np.linalg.solve(Rn, ds[..., None])[..., 0]
    den = np.einsum("...d,...d->...", ds.conj(), num)
    return num / den[..., None]


def classic_enhance(samps, kind, geometry, doa, normalize=False, chunk_len=-1, c=343, sr=16000,
                    frame_len=512, frame_hop=256, window="hann", center=True, **geo):
    """apply_classic_beamformer.py:88-112 for one utterance: STFT -> DS / SD beamformer
    (one DoA, or one per chunk of chunk_len frames) -> inverse STFT, rescaled to
    max |samps| only with normalize."""
    kw = dict(frame_len=frame_len, frame_hop=frame_hop, window=window, center=center)
    obs = multichannel_stft(samps, transpose=False, **kw)
    F = obs.shape[1]
    if chunk_len > 0:
        enh = np.hstack([beamform(classic_weight(kind, geometry, d, F, c=c, sr=sr, **geo),
                                  obs[:, :, k * chunk_len:(k + 1) * chunk_len])
                         for k, d in enumerate(doa)])
    else:
        enh = beamform(classic_weight(kind, geometry, doa, F, c=c, sr=sr, **geo), obs)
    norm = float(np.max(np.abs(samps))) if normalize else None
    return inverse_stft(enh, norm=norm, transpose=False, **kw)


BEAMFORMERS = ("mvdr", "mpdr", "mpdr-whiten", "gevd", "pmwf-0", "pmwf-1")


def supervised_run(kind, mask_s, obs, mask_n=None, ban=False, pmwf_ref=-1,
                   rank1_appro="", gauge=False, return_parts=False):
    """SupervisedBeamformer.run / MpdrBeamformer.run,
    libs/beamformer.py:270-283, 573-590."""
    if kind in ("mpdr", "mpdr-whiten"):
        Rs = compute_covar(obs, mask_s)
        Ry = compute_covar(obs, np.ones_like(mask_s))
        Rn = None
        if kind == "mpdr-whiten":
            Rn = compute_covar(obs, 1 - mask_s if mask_n is None else mask_n)
        w = mpdr_weight(Rs, Ry, Rn=Rn, gauge=gauge)
        if ban:
            if Rn is None:
                # reference raises NameError here (libs/beamformer.py:590)
                raise NameError("Rn")
            w = do_ban(w, Rn)
    else:
        Rn = compute_covar(obs, 1 - mask_s if mask_n is None else mask_n)
        Rs = compute_covar(obs, mask_s)
        if kind == "mvdr":
            w = mvdr_weight(Rs, Rn, gauge=gauge)
        elif kind == "gevd":
            w = gevd_weight(Rs, Rn, gauge=gauge)
        elif kind in ("pmwf-0", "pmwf-1"):
            w = pmwf_weight(Rs, Rn, beta=int(kind[-1]), ref_channel=pmwf_ref,
                            rank1_appro=rank1_appro, gauge=gauge)
        else:
            raise ValueError(kind)
        if ban:
            w = do_ban(w, Rn)
    enh = beamform(w, obs)
    if return_parts:
        return enh, dict(Rs=Rs, Rn=Rn, weight=w)
    return enh


def compute_vad_masks(spectrogram, proportion):
    # apply_adaptive_beamformer.py:50-71 (vectorised, same threshold/index)
    e = np.sqrt(spectrogram.real**2 + spectrogram.imag**2)
    vec = np.sort(e.flatten())
    filt = np.sum(vec) * (1 - proportion)
    threshold, cumsum, index = 0, 0, 0
    while index < vec.shape[0]:
        threshold = vec[index]
        cumsum += threshold
        if cumsum > filt:
            break
        index += 1
    return (e < threshold).transpose(), index


def enhance_utterance(samps, mask, kind="mvdr", itf_mask=None, frame_len=512,
                      frame_hop=256, center=True, round_power_of_two=True,
                      window="hann", ban=False, pmwf_ref=-1, rank1_appro="",
                      post_mask=False, vad_proportion=1, gauge=False,
                      return_parts=False):
    """The per-utterance body of apply_adaptive_beamformer.py:130-178.
    samps: C x N float32; mask: T x F (or F x T); returns float waveform."""
    kw = dict(frame_len=frame_len, frame_hop=frame_hop, window=window,
              center=center, transpose=False)
    stft_mat = multichannel_stft(samps,
                                 round_power_of_two=round_power_of_two, **kw)
    norm = np.max(np.abs(samps))
    speech_mask = mask
    if itf_mask is None:
        speech_mask = np.minimum(speech_mask, 1)
    interf_mask = itf_mask
    _, F, _ = stft_mat.shape
    if speech_mask.shape[0] == F and speech_mask.shape[1] != F:
        speech_mask = np.transpose(speech_mask)
        if interf_mask is not None:
            interf_mask = np.transpose(interf_mask)
    if 0.5 < vad_proportion < 1:
        vad_mask, _ = compute_vad_masks(stft_mat[0], vad_proportion)
        speech_mask = np.where(vad_mask, 1.0e-4, speech_mask)
        if interf_mask is not None:
            interf_mask = np.where(vad_mask, 1.0e-4, interf_mask)
    enh, parts = supervised_run(kind, speech_mask, stft_mat, mask_n=interf_mask,
                                ban=ban, pmwf_ref=pmwf_ref,
                                rank1_appro=rank1_appro, gauge=gauge,
                                return_parts=True)
    if post_mask:
        enh = enh * np.transpose(speech_mask)
    wav = inverse_stft(enh, norm=norm, **kw)
    if return_parts:
        parts.update(stft=stft_mat, enh=enh, norm=norm)
        return wav, parts
    return wav


# ----------------------------------------------------------------------------
# WPE / facted WPD (libs/wpe.py)
# ----------------------------------------------------------------------------
def compute_tap_mat(obs, taps, delay):
    """libs/wpe.py:13-29.  obs F x N x T -> F x NK x T, row k N + n = channel n
    delayed by k + delay frames."""
    F, N, T = obs.shape
    y = np.zeros([F, N * taps, T], dtype=obs.dtype)
    for k in range(taps):
        d = k + delay
        if d >= T:
            break
        y[:, k * N:(k + 1) * N, d:] = obs[:, :, :T - d]
    return y


def compute_lambda(dereverb, ctx=0):
    """libs/wpe.py:32-55: channel-mean power, averaged over the +-ctx frames that
    exist, floored at eps_f32.  F x N x T -> F x T (float64 through the count)."""
    L = np.mean(dereverb.real**2 + dereverb.imag**2, axis=1)
    _, T = L.shape
    counts = np.zeros(T)
    lam = np.zeros_like(L)
    for c in range(-ctx, ctx + 1):
        s, e = max(c, 0), min(T, T + c)
        lam[:, s:e] += L[:, max(-c, 0):min(T, T - c)]
        counts[s:e] += 1
    return np.maximum(lam / counts, EPSILON)


def wpe_step(reverb, yt, lam):
    """libs/wpe.py:58-81.  reverb F x N x T, yt F x NK x T, lam F x T."""
    yn = yt / lam[:, None, :]
    R = np.einsum("...mt,...nt->...mn", yn, yt.conj())
    r = np.einsum("...mt,...nt->...mn", yn, reverb.conj())
    G = np.linalg.solve(R, r)
    return reverb - np.einsum("...na,...nb->...ab", G.conj(), yt)


def wpe(reverb, taps=10, delay=3, context=1, num_iters=3):
    """libs/wpe.py:84-110.  F x N x T -> F x N x T."""
    yt = compute_tap_mat(reverb, taps, delay)
    dereverb = reverb
    for _ in range(num_iters):
        dereverb = wpe_step(reverb, yt, compute_lambda(dereverb, ctx=context))
    return dereverb


def facted_wpd(obs, cgmm_iters=10, wpd_iters=3, taps=10, delay=3, context=1, gauge=False,
               update_alpha=False):
    """libs/wpe.py:113-177.  obs N x T x F ->
    (tf_mask T x F x 2, wpd_enh T x F).  gauge=True fixes the sign of the
    steering vector (solve_pevd) as everywhere else in this oracle."""
    obs = np.einsum("ntf->fnt", obs)
    yt = compute_tap_mat(obs, taps, delay)
    wpd_enh = None
    for i in range(wpd_iters):
        lam = compute_lambda(obs, ctx=context) if i == 0 else np.abs(wpd_enh)**2
        lam = np.maximum(lam, EPSILON)
        der = wpe_step(obs, yt, lam)
        der_r = np.einsum("fnt->nft", der)
        gamma = cgmm_gamma(der_r, cgmm_iters, update_alpha=update_alpha)  # K x F x T
        Rd = np.einsum("...nt,...mt->...nm", der / lam[:, None], der.conj()) / der.shape[-1]
        Rs = compute_covar(der_r, gamma[0].T)
        sv = solve_pevd(Rs, gauge=gauge)
        num = _solve_vec(Rd, sv)
        den = np.einsum("...d,...d->...", sv.conj(), num)
        weight = num / den[:, None]
        wpd_enh = np.einsum("...n,...nt->...t", weight.conj(), der)
    return np.transpose(gamma, (2, 1, 0)), wpd_enh.T


# ----------------------------------------------------------------------------
# CGMM mask estimation (libs/cluster.py) -- K=2, deterministic init
# ----------------------------------------------------------------------------
class _Covariance:
    # libs/cluster.py:94-135
    def __init__(self, covar):
        covar = (covar + np.einsum("...xy->...yx", covar.conj())) / 2
        w, v = np.linalg.eigh(covar)
        w = w / np.maximum(np.amax(w, axis=-1, keepdims=True), EPSILON)
        self.w = np.maximum(w, EPSILON)
        self.v = v

    def inv(self):
        return np.einsum("...xy,...y,...zy->...xz", self.v, 1 / self.w,
                         self.v.conj())

    def logdet(self):
        return np.sum(np.log(self.w), axis=-1, keepdims=True)


def cgmm_masks(stft_mat, num_iters=20, init_mask=None):
    """CgmmTrainer(K=2).train + estimate_cgmm_masks.py:44-64.
    stft_mat N x F x T -> speech mask T x F float32."""
    gamma = cgmm_gamma(stft_mat, num_iters, init_mask)
    return np.transpose(gamma, (0, 2, 1))[0].astype(np.float32)


def cgmm_gamma(stft_mat, num_iters=20, init_mask=None, update_alpha=False, num_classes=2, seed=None,
               gamma0=None):
    """CgmmTrainer(stft_mat, num_classes, update_alpha=...).train(num_iters): posteriors
    K x F x T (float64).  update_alpha: Cgmm.update, libs/cluster.py:246-257.
    num_classes != 2: the random start of libs/cluster.py:427-434 -- np.random.uniform from
    the legacy GLOBAL generator, which estimate_cgmm_masks.py:28 seeds once per run; `seed`
    re-seeds it here (None: whatever state the generator is in, as for the second utterance
    of a run), `gamma0` K x F x T passes a start in directly."""
    obs = np.einsum("mft->fmt", stft_mat)
    F, M, T = obs.shape
    K = int(num_classes)
    if K != 2 or gamma0 is not None:
        if gamma0 is None:
            if seed is not None:
                np.random.seed(seed)
            gamma0 = np.random.uniform(size=[K, F, T])
            gamma0 = gamma0 / np.sum(gamma0, 0, keepdims=True)
        gamma = np.asarray(gamma0, dtype=np.float64)
        den = np.maximum(np.sum(gamma, axis=-1, keepdims=True), EPSILON)
        R = np.einsum("...t,...xt,...yt->...xy", gamma, obs, obs.conj()) / den[..., None]
    elif init_mask is None:  # libs/cluster.py:419-425
        Rs = np.einsum("...dt,...et->...de", obs, obs.conj()) / T
        Rn = np.stack([np.eye(M, M, dtype=complex) for _ in range(F)])
        R = np.stack([Rs, Rn])
    else:  # :427, 436-440 ; init_mask F x T
        gamma = np.stack([init_mask, 1 - init_mask])
        den = np.maximum(np.sum(gamma, axis=-1, keepdims=True), EPSILON)
        R = np.einsum("...t,...xt,...yt->...xy", gamma, obs,
                      obs.conj()) / den[..., None]
    cov = _Covariance(R)
    phi = np.einsum("...xt,...xy,...yt->...t", obs.conj(), cov.inv(), obs)
    phi = np.maximum(np.abs(phi), EPSILON) / M
    alpha = np.ones([K, F]) / K

    def predict(cov, phi):  # libs/cluster.py:261-287, 214-235
        log_pdf = -M * np.log(phi) - cov.logdet()
        log_pdf = log_pdf - np.amax(log_pdf, 0, keepdims=True)
        nom = np.exp(log_pdf) * alpha[..., None]  # alpha: enclosing scope, see the loop
        return nom / np.maximum(np.sum(nom, 0, keepdims=True), EPSILON)

    gamma = predict(cov, phi)
    for _ in range(num_iters):  # libs/cluster.py:455-465, 193-212
        if update_alpha:
            alpha = np.mean(gamma, -1)
        den = np.sum(gamma, -1, keepdims=True)
        R = np.einsum("...t,...xt,...yt->...xy", gamma * M / phi, obs,
                      obs.conj())
        R = R / np.maximum(den[..., None], EPSILON)
        cov = _Covariance(R)
        phi = np.einsum("...xt,...xy,...yt->...t", obs.conj(), cov.inv(), obs)
        phi = np.maximum(np.abs(phi), EPSILON) / M
        gamma = predict(cov, phi)
    return gamma


# ----------------------------------------------------------------------------
# synthetic workload -- shared by tests and bench.py
# ----------------------------------------------------------------------------
def synth_utterance(index, num_channels, num_samples, return_parts=False):
    """default_rng(1234+index): point source delayed by d_c in [0,8) samples per
    channel (gain 0.3) + spatially white N(0, 0.05^2) noise, whole mix x 0.2."""
    rng = np.random.default_rng(1234 + index)
    src = rng.standard_normal(num_samples + 8).astype(np.float32)
    delays = rng.integers(0, 8, size=num_channels)
    speech = np.stack([src[8 - d:8 - d + num_samples] for d in delays]) * 0.3
    noise = rng.standard_normal((num_channels, num_samples)).astype(
        np.float32) * 0.05
    mix = ((speech + noise) * 0.2).astype(np.float32)
    if return_parts:
        return mix, (speech * 0.2).astype(np.float32), (noise * 0.2).astype(
            np.float32)
    return mix



SKIPSET_KINDS = {
    # name -> enhance_utterance keywords (the beamformers of apply_adaptive_beamformer.py)
    "mvdr": dict(kind="mvdr"), "gevd": dict(kind="gevd"), "mpdr": dict(kind="mpdr"),
    "mpdr-whiten": dict(kind="mpdr-whiten"), "pmwf-0": dict(kind="pmwf-0"),
    "pmwf-0-eig": dict(kind="pmwf-0", rank1_appro="eig"),
    "pmwf-0-gev": dict(kind="pmwf-0", rank1_appro="gev"),
}


def skipset_cases():
    """Inputs on which the reference's numpy.linalg.solve does / does not raise
    (tests/golden/ref_skipset.json holds what the UNMODIFIED reference did with each:
    oracle/make_golden.py gen_skipset).  name -> (samps C x N, mask T x F).  Structurally
    singular: the elimination cancels exactly (duplicated, silent, power-of-two scaled
    channel; all-zero noise covariance; silence).  Singular to rounding only: a channel that
    is 0.3 x another or the sum of two others, a noise mask with fewer frames than channels."""
    mix, sp, nz = synth_utterance(3, 4, 32000, return_parts=True)
    mask = irm_mask(sp, nz)
    out = {}
    out["plain"] = (mix, mask)
    out["ones-mask"] = (mix, np.ones_like(mask))
    out["mask-above-one"] = (mix, np.ones_like(mask) * 1.5)
    d = mix.copy(); d[1] = d[0]
    out["dup-channel"] = (d, mask)
    d = mix.copy(); d[3] = 0
    out["zero-channel"] = (d, mask)
    d = mix.copy(); d[2] = 2 * d[0]
    out["channel-x2"] = (d, mask)
    d = mix.copy(); d[2] = np.float32(0.3) * d[0]
    out["channel-x0.3"] = (d, mask)
    d = mix.copy(); d[2] = d[0] + d[1]
    out["sum-channel"] = (d, mask)
    m = np.ones_like(mask); m[:3] = 0
    out["noise-in-3-frames"] = (mix, m)
    out["silence"] = (np.zeros_like(mix), mask)
    return out


def synth_scene(index, num_channels, num_samples, return_parts=False):
    """A scene with time-frequency structure for the blind (CGMM) path: a gated,
    spectrally coloured point source through a short random FIR per channel, in
    spatially correlated (full-rank) diffuse noise.  default_rng(4321 + index)."""
    rng = np.random.default_rng(4321 + index)
    C, N = num_channels, num_samples
    # source: coloured noise (two resonances), gated on/off in 0.1 - 0.6 s segments
    src = rng.standard_normal(N + 64)
    for fc, r in ((rng.uniform(300, 900), 0.97), (rng.uniform(1500, 3000), 0.9)):
        a1, a2 = -2 * r * np.cos(2 * np.pi * fc / 16000.0), r * r
        src = src + 0.7 * scipy.signal.lfilter([1.0], [1.0, a1, a2], src) * (1 - r)
    env = np.zeros(N + 64)
    pos, on = 0, bool(rng.integers(0, 2))
    while pos < N + 64:
        seg = int(rng.uniform(0.1, 0.6) * 16000)
        if on:
            env[pos:pos + seg] = rng.uniform(0.5, 1.0)
        on = not on
        pos += seg
    src = src * env
    src = src / max(np.sqrt(np.mean(src**2)), 1e-9)
    taps = rng.standard_normal((C, 16)) * np.exp(-np.arange(16) / 3.0)
    taps[:, 0] += 1.0
    speech = np.stack([np.convolve(src, taps[c])[32:32 + N] for c in range(C)]) * 0.25
    mixm = np.eye(C) + 0.4 * rng.standard_normal((C, C))
    noise = (mixm @ rng.standard_normal((C, N))) * 0.08
    mix = ((speech + noise) * 0.2).astype(np.float32)
    if return_parts:
        return mix, (speech * 0.2).astype(np.float32), (noise * 0.2).astype(np.float32)
    return mix


def irm_mask(speech, noise, frame_len=512, frame_hop=256, center=True):
    """compute_mask.py:77-107 "irm" on channel 0: |S| / sqrt(|S|^2+|V|^2+eps),
    T x F float32."""
    kw = dict(frame_len=frame_len, frame_hop=frame_hop, center=center,
              window="hann", transpose=True)
    S = np.abs(forward_stft(speech[0], **kw))
    V = np.abs(forward_stft(noise[0], **kw))
    return (S / np.sqrt(S**2 + V**2 + EPSILON)).astype(np.float32)
