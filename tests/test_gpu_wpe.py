"""
GPU parity of WPE / factorised WPD (libs/wpe.py) against the oracle
restatement (itself equal to the unmodified reference, tests/test_oracle_golden.py)
and the reference's stored doc outputs (doc/wpe/asset).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden, pcm16_rel_rms, rel_rms, rms
from oracle import np_oracle as o
from oracle import make_golden as mg

pytestmark = pytest.mark.gpu
STFT_KW = dict(frame_len=512, frame_hop=256, window="hann", center=True)


def test_wpe_small_known_answer():
    from setk_amd.libs import wpe as W
    g = load_golden("ref_wpe.npz")
    rev, _ = mg.wpe_small_case()
    out = W.wpe(rev, taps=4, delay=2, context=1, num_iters=2)
    assert out.shape == g["small.wpe"].shape and out.dtype == np.complex128
    assert rel_rms(out, g["small.wpe"]) < 1e-5, rel_rms(out, g["small.wpe"])
    assert np.array_equal(W.compute_tap_mat(rev, 3, 1), g["small.tap"])
    assert np.allclose(W.compute_lambda(rev, ctx=2), g["small.lambda"], rtol=1e-6)
    # one step with caller supplied variances
    lam = o.compute_lambda(rev, ctx=1)
    yt = o.compute_tap_mat(rev, 4, 2)
    assert rel_rms(W.wpe_step(rev, yt, lam), o.wpe_step(rev, yt, lam)) < 1e-5
    # an observation that itself STARTS with exactly-zero frames (digital silence): the delay
    # must not be read off the zero prefix of yt; variances far below eps are used as given
    rev0 = rev.copy()
    rev0[:, :, :5] = 0
    yt0 = o.compute_tap_mat(rev0, 4, 2)
    lam0 = np.maximum(o.compute_lambda(rev0, ctx=1) * 1e-9, 1e-16)
    ref0 = o.wpe_step(rev0, yt0, lam0)
    assert rel_rms(W.wpe_step(rev0, yt0, lam0), ref0) < 1e-5
    assert rel_rms(W.wpe_step(rev0, yt0, lam0, taps=4, delay=2), ref0) < 1e-5
    with pytest.raises(ValueError):
        W.wpe_step(rev0, yt0 * 2, lam0)
    with pytest.raises(ValueError):
        W.wpe_step(rev0, yt0, lam0, taps=4, delay=3)


@pytest.mark.parametrize("C,taps,N", [(8, 10, 40000), (2, 12, 20000), (6, 5, 30000), (1, 10, 16000)])
def test_wpe_matches_oracle(C, taps, N):
    """Up to NK = 80 tap-stacked channels (8 ch x 10 taps): cond(R) reaches 1e10, the
    reference computes in complex128 -- the device result must agree to 1e-4."""
    from setk_amd.libs import wpe as W
    mix = o.synth_utterance(90 + C, C, N)
    # reverberate: a few decaying reflections per channel
    rng = np.random.default_rng(C)
    rev = mix.copy()
    for d in (700, 1500, 2600, 4100):
        rev[:, d:] += 0.4 * rng.uniform(0.5, 1.0) * mix[:, :-d][::-1]
    kw = dict(frame_len=512, frame_hop=128, window="hann", center=True)
    obs = o.multichannel_stft(rev, transpose=True, **kw)
    if obs.ndim == 2:
        obs = obs[None]
    fnt = np.transpose(obs, (2, 0, 1))
    ref = o.wpe(fnt, taps=taps, delay=3, context=1, num_iters=3)
    out = W.wpe(fnt, taps=taps, delay=3, context=1, num_iters=3)
    assert rel_rms(out, ref) < 1e-4, (C, taps, rel_rms(out, ref))


def test_wpe_batch_equals_single_calls():
    """setk_wpe_batch: ragged lengths in one launch per iteration == one utterance at a
    time, bit for bit; a singular utterance inside the batch comes back as None."""
    from setk_amd.libs import wpe as W
    rng = np.random.default_rng(3)
    revs = [(rng.standard_normal((257, 3, T)) + 1j * rng.standard_normal((257, 3, T))).astype(np.complex64)
            for T in (90, 61, 140)]
    revs.insert(2, np.zeros((257, 3, 50), np.complex64))
    outs = W.wpe_batch(revs, taps=5, delay=2, context=1, num_iters=2)
    wide = W.wpe_batch(revs, taps=5, delay=2, context=1, num_iters=2, dtype=np.complex128)
    assert outs[2] is None and wide[2] is None
    for k in (0, 1, 3):
        single = W.wpe(revs[k], taps=5, delay=2, context=1, num_iters=2)
        # complex64 (what the device stores) unless the reference's promoted type is asked for
        assert outs[k].dtype == np.complex64 and np.array_equal(outs[k].astype(np.complex128), single), k
        assert wide[k].dtype == np.complex128 and np.array_equal(wide[k], single), k


def test_wpe_batch_var_equals_single_calls():
    """setk_wpe_batch_var (the WPE step of facted_wpd for a batch: per-utterance variances of a
    previous enhanced signal in, 1 / lambda out, libs/wpe.py:146-162) == setk_wpe one utterance at
    a time, bit for bit; without variances it is setk_wpe_batch."""
    from setk_amd import _ffi
    ctx = _ffi.default_context()
    rng = np.random.default_rng(11)
    C, F, frames = 3, 257, (90, 61, 140)
    specs = [(rng.standard_normal((C, T, F)) + 1j * rng.standard_normal((C, T, F))).astype(np.complex64)
             for T in frames]
    enh = [(rng.standard_normal((T, F)) + 1j * rng.standard_normal((T, F))).astype(np.complex64) for T in frames]
    for use_enh in (True, False):
        outs = [np.empty_like(x) for x in specs]
        invs = [np.empty((T, F), np.float32) for T in frames]
        status = np.full((len(frames), F), -1, np.int32)
        ctx.wpe_batch_var(specs, C, frames, F, 5, 2, 1, 1, outs, lambda_enh=enh if use_enh else None,
                          inv_lambda_outs=invs, status=status)
        assert not status.any()
        for k, T in enumerate(frames):
            one, inv1, st1 = np.empty_like(specs[k]), np.empty((T, F), np.float32), np.full(F, -1, np.int32)
            ctx.wpe(specs[k], C, T, F, 5, 2, 1, 1, one, lambda_enh=enh[k] if use_enh else None,
                    inv_lambda_out=inv1, status=st1)
            assert not st1.any() and np.array_equal(outs[k], one) and np.array_equal(invs[k], inv1), (use_enh, k)
    plain = [np.empty_like(x) for x in specs]
    ctx.wpe_batch(specs, C, frames, F, 5, 2, 1, 1, plain)
    assert all(np.array_equal(a, b) for a, b in zip(plain, outs))


def test_wpe_beyond_256_tap_rows_is_refused():
    from setk_amd import _ffi
    from setk_amd.libs import wpe as W
    fnt = np.zeros((9, 16, 50), np.complex64)
    with pytest.raises(_ffi.SetkUnsupported):
        W.wpe(fnt, taps=17)


@pytest.mark.parametrize("N,taps", [(8, 12), (16, 6), (16, 10), (8, 11), (16, 16)])
def test_wpe_wide_shapes_match_the_oracle(N, taps):
    """Channels x taps whose R does not fit the 160 KB of a CU (the reference has no bound,
    libs/wpe.py:58-81): R is factored in global memory, everything else is the LDS form's
    code.  Against the complex128 oracle on reverberant (AR-filtered) data, two iterations,
    plus the batch entry with a second, shorter utterance."""
    from oracle import np_oracle as o
    from setk_amd.libs import wpe as W
    rng = np.random.default_rng(100 * N + taps)
    F, T = 6, 60 * taps + 300
    def scene(T):
        x = rng.standard_normal((F, N, T)) + 1j * rng.standard_normal((F, N, T))
        for t in range(4, T):                      # a few reflections per channel
            x[:, :, t] += 0.5 * x[:, :, t - 3] - 0.3j * np.roll(x[:, :, t - 4], 1, axis=1)
        return x.astype(np.complex64)
    rev = scene(T)
    ref = o.wpe(rev.astype(np.complex128), taps=taps, delay=3, context=1, num_iters=2)
    got = W.wpe(rev, taps=taps, delay=3, context=1, num_iters=2)
    err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    assert err < 1e-4, (N, taps, err)
    # the filter did something: the result differs from the input by far more than that
    assert np.linalg.norm(got - rev) / np.linalg.norm(rev) > 1e-2
    short = scene(T // 2 + 7)
    outs = W.wpe_batch([rev, short], taps=taps, delay=3, context=1, num_iters=2)
    assert np.array_equal(outs[0].astype(np.complex128), got)
    ref2 = o.wpe(short.astype(np.complex128), taps=taps, delay=3, context=1, num_iters=2)
    assert np.linalg.norm(outs[1] - ref2) / np.linalg.norm(ref2) < 1e-4


def test_wpe_singular_is_linalg_error():
    from setk_amd.libs import wpe as W
    fnt = np.zeros((5, 2, 40), np.complex64)  # all-zero observation: R = 0
    with pytest.raises(np.linalg.LinAlgError):
        W.wpe(fnt, taps=3, delay=1)


def test_wpe_rank_deficient_goes_through_like_lu():
    """Fewer frames than channels x taps: R is semi-definite.  numpy's pivoted LU (the
    reference) returns a finite result without raising; the device Cholesky drops the
    columns whose pivot is at the noise level (dividing by a floored pivot lets the noise
    grow until it overflows: 1 bin in 5 of the first case with the MFMA summation order)
    and must neither drop the utterance nor return anything unbounded."""
    from setk_amd.libs import wpe as W
    rng = np.random.default_rng(12)
    fnt = (rng.standard_normal((5, 4, 10)) + 1j * rng.standard_normal((5, 4, 10))).astype(np.complex64)
    out = W.wpe(fnt, taps=5, delay=1, context=0, num_iters=1)
    assert out.shape == fnt.shape and np.all(np.isfinite(out))
    # ... and says so: SETK_NUM_RANKDEF in the status words (a note, not an error), a warning
    # in the log (the divergence from numpy.linalg.solve used to be silent)
    import logging
    from setk_amd import _ffi
    spec = np.ascontiguousarray(np.transpose(fnt, (1, 2, 0)))
    st = np.zeros(5, dtype=np.int32)
    _ffi.default_context().wpe(spec, 4, 10, 5, 5, 1, 0, 1, np.empty_like(spec), status=st)
    assert (st == _ffi.NUM_RANKDEF).all() and not _ffi.wpe_failed(st).any()
    full = (rng.standard_normal((5, 2, 200)) + 1j * rng.standard_normal((5, 2, 200))).astype(np.complex64)
    spec = np.ascontiguousarray(np.transpose(full, (1, 2, 0)))
    _ffi.default_context().wpe(spec, 2, 200, 5, 3, 1, 0, 1, np.empty_like(spec), status=st)
    assert (st == 0).all()
    records = []
    h = logging.Handler()
    h.emit = records.append
    logging.getLogger("setk_amd.libs.wpe").addHandler(h)
    try:
        W.wpe(fnt, taps=5, delay=1, context=0, num_iters=1)
    finally:
        logging.getLogger("setk_amd.libs.wpe").removeHandler(h)
    assert any("rank-deficient" in r.getMessage() for r in records)
    for seed, (C, T, taps) in enumerate([(4, 10, 5), (4, 16, 6), (2, 9, 12), (8, 30, 10), (6, 7, 3)]):
        rng = np.random.default_rng(200 + seed)
        fnt = (rng.standard_normal((64, C, T)) + 1j * rng.standard_normal((64, C, T))).astype(np.complex64)
        out = W.wpe(fnt, taps=taps, delay=1, context=0, num_iters=2)
        assert np.all(np.isfinite(out)), (C, T, taps)
        assert np.abs(out).max() < 100 * np.abs(fnt).max(), (C, T, taps, np.abs(out).max())


def test_facted_wpd_matches_oracle():
    from setk_amd.libs import wpe as W
    _, mix = mg.wpe_small_case()
    obs = o.multichannel_stft(mix, transpose=True, **STFT_KW)
    mask_ref, enh_ref = o.facted_wpd(obs, cgmm_iters=3, wpd_iters=2, taps=4, delay=2, context=1,
                                     gauge=True)
    mask, enh = W.facted_wpd(obs, cgmm_iters=3, wpd_iters=2, taps=4, delay=2, context=1)
    assert mask.shape == mask_ref.shape and enh.shape == enh_ref.shape
    assert np.mean(np.abs(mask - mask_ref)) < 2e-4
    assert rel_rms(enh, enh_ref) < 1e-3, rel_rms(enh, enh_ref)


def test_doc_wpe_and_wpd_clis(tmp_path):
    """doc/wpe/README.md: the two command lines, against the stored wavs."""
    import scipy.io.wavfile
    from test_oracle_golden import resolve_gauge
    g = load_golden("ref_wpe.npz")
    td = str(tmp_path)
    scipy.io.wavfile.write(f"{td}/egs.wav", 16000, g["egs"])
    open(f"{td}/wav.scp", "w").write(f"egs {td}/egs.wav\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts/sptk/apply_wpe.py"),
                        "--frame-len", "512", "--frame-hop", "128", f"{td}/wav.scp", f"{td}/wpe"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Processed 1 utterances over 1" in r.stderr
    sr, y = scipy.io.wavfile.read(f"{td}/wpe/egs.wav")
    assert y.shape == g["wpe_egs"].shape and y.dtype == np.int16
    err = rms(y.astype(np.float64), g["wpe_egs"].astype(np.float64)) / rms(g["wpe_egs"].astype(np.float64))
    assert err < 1e-3, err
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts/sptk/apply_wpd.py"),
                        "--frame-len", "512", "--taps", "10", "--delay", "3", "--context", "1",
                        "--wpd-iters", "2", "--cgmm-iters", "10", "--update-alpha", "false",
                        "--dump-mask", "true", f"{td}/wav.scp", f"{td}/wpd"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    sr, y = scipy.io.wavfile.read(f"{td}/wpd/egs.wav")
    assert y.shape == g["wpd_egs"].shape
    assert np.load(f"{td}/wpd/egs.npy").shape == (209, 257)
    # the stored wav carries LAPACK's per-bin signs, ours the declared gauge:
    # compare with the gauge-fixed oracle, and the oracle with the stored wav
    samps = (g["egs"].astype(np.float32) / 32768.0).T.copy()
    obs = o.multichannel_stft(samps, transpose=True, **STFT_KW)
    _, enh = o.facted_wpd(obs, cgmm_iters=10, wpd_iters=2, taps=10, delay=3, context=1, gauge=True)
    ref = o.inverse_stft(enh, norm=np.max(np.abs(samps)), transpose=True, **STFT_KW)
    assert pcm16_rel_rms(y, ref) < 1e-3, pcm16_rel_rms(y, ref)


@pytest.mark.parametrize("hop,window,pcm", [(128, "blackman", False), (256, "hann", True)])
def test_batch_dereverb_resident_matches_oracle(hop, window, pcm):
    """engine.BatchDereverb (what apply_wpe.py runs): samples -> STFT -> WPE -> inverse STFT
    without leaving the device, ragged batch, against the oracle's forward_stft -> wpe ->
    inverse_stft of every channel (apply_wpe.py:30-66)."""
    from setk_amd.engine import BatchDereverb, Pcm16Frames
    kw = dict(frame_len=512, frame_hop=hop, window=window, center=True)
    utts = []
    for u, N in enumerate((30000, 17333, 24001)):
        mix = o.synth_utterance(300 + u, 3, N)
        rev = mix.copy()
        for d in (900, 2100):
            rev[:, d:] += 0.4 * mix[:, :-d]
        if pcm:
            q = np.round(rev * 32768.0 * 0.5).astype(np.int16)
            utts.append((Pcm16Frames(np.ascontiguousarray(q.T)), q.astype(np.float32) / 32768.0))
        else:
            utts.append((rev.astype(np.float32), rev.astype(np.float32)))
    eng = BatchDereverb(taps=6, delay=3, context=1, num_iters=3, **kw)
    outs = eng.run([a for a, _ in utts])
    for (_, samps), got in zip(utts, outs):
        obs = o.multichannel_stft(samps, transpose=True, **kw)
        der = o.wpe(np.transpose(obs, (2, 0, 1)), taps=6, delay=3, context=1, num_iters=3)
        ref = np.stack([o.inverse_stft(s, transpose=True, **kw) for s in np.transpose(der, (1, 2, 0))])
        assert got.shape == ref.shape and got.dtype == np.float32
        assert rel_rms(got, ref) < 1e-4, rel_rms(got, ref)
    # the CLI's form: frames quantised on the device by the wav writer's rule
    from setk_amd.libs.wavio import float_to_pcm16
    q = BatchDereverb(taps=6, delay=3, context=1, num_iters=3, pcm16=True, **kw).run([a for a, _ in utts])
    for f32, i16 in zip(outs, q):
        assert i16.dtype == np.int16 and np.array_equal(i16, float_to_pcm16(f32).T)
    # a silent utterance inside the batch is the reference's LinAlgError, the others go through
    outs2 = eng.run([utts[0][0], np.zeros((3, 20000), np.float32)] if not pcm else
                    [utts[0][0], Pcm16Frames(np.zeros((20000, 3), np.int16))])
    assert outs2[1] is None and np.array_equal(outs2[0], outs[0])


def test_batch_wpd_resident_engine_matches_oracle():
    """engine.BatchWpd (what apply_wpd.py runs since round 5): samples -> STFT -> 2 x (WPE step,
    CGMM, power- and mask-weighted covariances, MVDR, beamformer) -> inverse STFT + renorm, on
    device pointers of one scratch block; a ragged batch, float32 and 16-bit PCM inputs, against
    the oracle's facted_wpd (libs/wpe.py:113-177) + inverse_stft (apply_wpd.py:47-49)."""
    from setk_amd.engine import BatchWpd, Pcm16Frames
    kw = dict(frame_len=512, frame_hop=256, window="hann", center=True)
    utts, floats = [], []
    for u, N in enumerate((24000, 17001)):
        mix, sp, nz = o.synth_utterance(340 + u, 4, N, return_parts=True)
        rev = mix.copy()
        rev[:, 700:] += 0.35 * mix[:, :-700]
        q = np.round(rev * 32768.0 * 0.8).astype(np.int16)
        floats.append(q.astype(np.float32) / 32768.0)
        utts.append(Pcm16Frames(np.ascontiguousarray(q.T)) if u == 0 else floats[-1])
    eng = BatchWpd(taps=6, delay=2, context=1, wpd_iters=2, cgmm_iters=6, **kw)
    res = eng.run(utts)
    assert all(r is not None for r in res)
    for (wave, mask), samps in zip(res, floats):
        obs = o.multichannel_stft(samps, transpose=True, round_power_of_two=True, **kw)
        tf_mask, enh = o.facted_wpd(obs, cgmm_iters=6, wpd_iters=2, taps=6, delay=2, context=1, gauge=True)
        ref = o.inverse_stft(enh, norm=np.max(np.abs(samps)), transpose=True, **kw)
        assert wave.dtype == np.float32 and wave.shape == ref.shape
        assert rms(wave, ref) / rms(ref) < 1e-3, rms(wave, ref) / rms(ref)
        assert mask.shape == tf_mask[..., 0].shape and np.mean(np.abs(mask - tf_mask[..., 0])) < 1e-3
    # PCM16 out: the device quantises by libsndfile's rule
    eng16 = BatchWpd(taps=6, delay=2, context=1, wpd_iters=2, cgmm_iters=6, pcm16=True, **kw)
    (w16, _), = eng16.run(utts[:1])
    assert w16.dtype == np.int16 and pcm16_rel_rms(w16, res[0][0]) < 1e-3
    eng.close()
    eng16.close()
    # another transform size: the numpy mirror behind the same interface
    eng400 = BatchWpd(taps=4, delay=2, context=1, wpd_iters=1, cgmm_iters=4, frame_len=400, frame_hop=160,
                      round_power_of_two=False)
    (w400, m400), = eng400.run([floats[1][:, :12000]])
    assert np.isfinite(w400).all() and m400.shape[1] == 201
