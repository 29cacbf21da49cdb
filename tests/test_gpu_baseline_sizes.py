"""
GPU-vs-oracle parity AT THE SIZES BASELINE.json names:

  configs[1]  4-ch  x 160 000 samples (10 s)  oracle-mask MVDR
  configs[2]  8-ch  x 480 000 samples (30 s)  oracle-mask MVDR, the 125-utterance
              shard bench.py times (same work split: several partial slabs and
              pass-2 ranges per utterance)
  configs[3]  8-ch  x 480 000 GEV
  configs[4]  6-ch  x 160 000 CGMM (20 EM iterations) -> MVDR

Waveforms are compared before PCM16 quantisation, gauge fixed on both sides, at
north_star's tolerance: <= 1e-3 relative RMS.  The fused covariance (the output
of stft_covar_kernel<C,false> + covar_finalize_kernel, tapped through
setk_enhance_batch_taps) is compared as a covariance, <= 1e-5, including the
Nyquist bin that takes a side path inside the kernel, and the weights
<= 1e-4.  The numpy oracle needs ~0.15 s per 8-ch / 30-s utterance.
"""
import numpy as np
import pytest
import torch

from conftest import rel_rms, rms
from oracle import np_oracle as o

pytestmark = pytest.mark.gpu
WAVE_TOL = 1e-3   # BASELINE.json north_star
COVAR_TOL = 1e-5
WEIGHT_TOL = 1e-4
F = 257


@pytest.fixture(scope="module")
def ctx():
    from setk_amd import _ffi
    c = _ffi.Context(0)
    c.stft_plan(512, 256, 512, True)
    yield c
    c.close()


def synth(index, C, N):
    mix, sp, nz = o.synth_utterance(index, C, N, return_parts=True)
    return mix, o.irm_mask(sp, nz)


def run_fused(ctx, kind, utts, masks, copies=1, taps=False):
    """utts/masks: the distinct utterances; the batch is `copies` rounds of them
    (device clones).  Returns (waves of the whole batch, status, taps dict)."""
    from setk_amd import _ffi
    dev = torch.device("cuda:0")
    C, nd = utts[0].shape[0], len(utts)
    a0 = [torch.from_numpy(np.ascontiguousarray(u)).to(dev) for u in utts]
    m0 = [torch.from_numpy(np.ascontiguousarray(m, dtype=np.float32)).to(dev) for m in masks]
    a, m, ns = [], [], []
    for r in range(copies):
        for i in range(nd):
            a.append(a0[i] if r == 0 else a0[i].clone())
            m.append(m0[i] if r == 0 else m0[i].clone())
            ns.append(utts[i].shape[1])
    n = len(a)
    outs = [torch.empty(ctx.istft_num_samples(ctx.num_frames(k)), dtype=torch.float32,
                        device=dev) for k in ns]
    tp = None
    if taps:
        tp = dict(Rs=torch.empty((n, F, C, C), dtype=torch.complex64, device=dev),
                  Rn=torch.empty((n, F, C, C), dtype=torch.complex64, device=dev),
                  weight=torch.empty((n, F, C), dtype=torch.complex64, device=dev),
                  maxabs=torch.empty(n, dtype=torch.float32, device=dev))
    kid = {"mvdr": _ffi.BF_MVDR, "gevd": _ffi.BF_GEVD}[kind]
    opts = _ffi.BfOpts(kind=kid, flags=_ffi.FLAG_CLAMP_MASK)
    st = ctx.enhance_batch(opts, C, [t.data_ptr() for t in a], ns, [t.data_ptr() for t in m],
                           None, [t.data_ptr() for t in outs], taps=tp)
    torch.cuda.synchronize()
    if tp:
        tp = {k: v.cpu().numpy() for k, v in tp.items()}
    return [t.cpu().numpy() for t in outs], st, tp


def check_against_oracle(kind, utts, masks, waves, taps=None, label=""):
    nd = len(utts)
    worst = 0.0
    for i in range(nd):
        ref, parts = o.enhance_utterance(utts[i], masks[i], kind=kind, gauge=True,
                                         return_parts=True)
        assert waves[i].shape == ref.shape
        err = rel_rms(waves[i], ref)
        worst = max(worst, err)
        assert err < WAVE_TOL, (label, kind, i, err)
        if taps is not None:
            Rs, Rn, w = taps["Rs"][i], taps["Rn"][i], taps["weight"][i]
            assert rel_rms(Rs, parts["Rs"]) < COVAR_TOL, (label, "Rs", rel_rms(Rs, parts["Rs"]))
            assert rel_rms(Rn, parts["Rn"]) < COVAR_TOL, (label, "Rn", rel_rms(Rn, parts["Rn"]))
            # the Nyquist bin is accumulated by a side path of the kernel
            assert rel_rms(Rs[256], parts["Rs"][256]) < COVAR_TOL
            assert rel_rms(Rn[256], parts["Rn"][256]) < COVAR_TOL
            assert np.max(np.abs(Rs - np.conj(np.transpose(Rs, (0, 2, 1))))) == 0
            assert rel_rms(w, parts["weight"]) < WEIGHT_TOL, (label, "w",
                                                             rel_rms(w, parts["weight"]))
            assert abs(taps["maxabs"][i] - np.max(np.abs(utts[i]))) == 0
    print(f"[{label}] {kind}: worst waveform rel rms vs oracle {worst:.2e}")
    return worst


def test_cfg1_4ch_10s_mvdr(ctx):
    """configs[1]: 4-ch x 10 s MVDR; three distinct utterances x 2 in one batch."""
    utts, masks = zip(*[synth(200 + i, 4, 160000) for i in range(3)])
    waves, st, taps = run_fused(ctx, "mvdr", utts, masks, copies=2, taps=True)
    assert st == [0] * 6
    check_against_oracle("mvdr", utts, masks, waves, taps, "cfg1 4ch x 10s")
    for i in range(3):
        assert np.array_equal(waves[i], waves[3 + i])


@pytest.mark.parametrize("kind", ["mvdr", "gevd"])
def test_cfg2_cfg3_8ch_30s_bench_shard(ctx, kind):
    """configs[2] / configs[3]: the 125 x (8-ch x 30 s) shard bench.py times, i.e.
    the same partial-slab / pass-2 range split; 3 distinct utterances, all 125
    outputs must equal their source's bit for bit and the oracle's to 1e-3."""
    utts, masks = zip(*[synth(300 + i, 8, 480000) for i in range(3)])
    copies = 42  # 126 utterances >= the bench shard of 125
    waves, st, taps = run_fused(ctx, kind, utts, masks, copies=copies, taps=True)
    assert st == [0] * (3 * copies)
    check_against_oracle(kind, utts, masks, waves, taps, "cfg2/3 8ch x 30s x 126")
    for r in range(1, copies):
        for i in range(3):
            assert np.array_equal(waves[i], waves[3 * r + i]), (r, i)


@pytest.mark.parametrize("kind", ["mvdr", "gevd"])
def test_8ch_30s_small_batch_many_slabs(ctx, kind):
    """The same utterances in a 3-utterance batch: the work list then cuts every
    utterance into many more partial slabs (up to 32) than in the shard above."""
    utts, masks = zip(*[synth(300 + i, 8, 480000) for i in range(3)])
    waves, st, taps = run_fused(ctx, kind, utts, masks, copies=1, taps=True)
    assert st == [0, 0, 0]
    check_against_oracle(kind, utts, masks, waves, taps, "8ch x 30s x 3")


def test_cfg4_6ch_cgmm_then_mvdr(ctx):
    """configs[4]: 6-ch, CGMM (20 EM iterations) on the device -> MVDR on the device,
    against the oracle's CGMM -> MVDR."""
    from setk_amd.engine import CgmmEstimator
    N = 160000
    mix = o.synth_utterance(400, 6, N)
    est = CgmmEstimator(num_iters=20, ctx=ctx)
    (mask_dev,) = est.estimate([mix])
    obs = o.multichannel_stft(mix, transpose=False, frame_len=512, frame_hop=256, window="hann",
                              center=True)
    mask_ref = o.cgmm_masks(obs, 20)
    assert mask_dev.shape == mask_ref.shape == (626, F)
    dm = np.abs(mask_dev - mask_ref)
    print(f"[cfg4] CGMM mask: mean |d| {dm.mean():.2e}, max |d| {dm.max():.2e}, "
          f"cells > 1e-3: {(dm > 1e-3).mean():.2e}")
    assert dm.mean() < 1e-4
    # the synthetic scene is so clean that the estimated mask is almost binary: the
    # noise covariance is then rank deficient and the reference beamforms on LU
    # rounding noise.  As the CGMM CLI test does, soften the mask for the waveform.
    soft_dev = (0.05 + 0.9 * mask_dev).astype(np.float32)
    soft_ref = (0.05 + 0.9 * mask_ref).astype(np.float32)
    (wave,), st, _ = run_fused(ctx, "mvdr", [mix], [soft_dev])
    assert st == [0]
    # (a) the beamformer stage alone, same mask on both sides: north_star tolerance
    ref_same = o.enhance_utterance(mix, soft_dev, kind="mvdr", gauge=True)
    err_same = rel_rms(wave, ref_same)
    # (b) end to end: device CGMM + MVDR vs oracle CGMM + MVDR
    ref_e2e = o.enhance_utterance(mix, soft_ref, kind="mvdr", gauge=True)
    err_e2e = rel_rms(wave, ref_e2e)
    print(f"[cfg4] waveform rel rms: same mask {err_same:.2e}, end to end {err_e2e:.2e}")
    assert err_same < WAVE_TOL
    assert err_e2e < WAVE_TOL


def test_full_batch_1000_utterances_oracle_spot_checks(ctx):
    """bench.py's `full_batch` leg: the whole configs[2] batch (1000 x 8-ch x 30 s,
    19.2 GB of audio) resident on ONE GPU in one setk_enhance_batch call.  Utterances
    0, 500 and 999 are distinct scenes checked against the oracle; the 997 others are
    copies of a fourth (also checked) and must equal each other bit for bit."""
    from setk_amd import _ffi
    dev = torch.device("cuda:0")
    C, N, n = 8, 480000, 1000
    special = {0: 600, 500: 601, 999: 602}
    src = {i: synth(idx, C, N) for i, idx in special.items()}
    filler = synth(603, C, N)
    f_a = torch.from_numpy(filler[0]).to(dev)
    f_m = torch.from_numpy(np.ascontiguousarray(filler[1], dtype=np.float32)).to(dev)
    audio, masks = [], []
    for i in range(n):
        if i in src:
            audio.append(torch.from_numpy(src[i][0]).to(dev))
            masks.append(torch.from_numpy(np.ascontiguousarray(src[i][1], dtype=np.float32)).to(dev))
        else:
            audio.append(f_a.clone())
            masks.append(f_m.clone())
    L = ctx.istft_num_samples(ctx.num_frames(N))
    outs = [torch.empty(L, dtype=torch.float32, device=dev) for _ in range(n)]
    opts = _ffi.BfOpts(kind=_ffi.BF_MVDR, flags=_ffi.FLAG_CLAMP_MASK)
    st = ctx.enhance_batch(opts, C, [t.data_ptr() for t in audio], [N] * n,
                           [t.data_ptr() for t in masks], None, [t.data_ptr() for t in outs])
    torch.cuda.synchronize()
    assert st == [0] * n
    worst = 0.0
    for i, (mix, mask) in list(src.items()) + [(1, filler)]:
        ref = o.enhance_utterance(mix, mask, kind="mvdr", gauge=True)
        err = rel_rms(outs[i].cpu().numpy(), ref)
        worst = max(worst, err)
        assert err < WAVE_TOL, (i, err)
    w1 = outs[1]
    for i in (2, 250, 499, 501, 750, 998):
        assert torch.equal(outs[i], w1), i
    print(f"[full batch 1000 x 8ch x 30s] worst waveform rel rms vs oracle {worst:.2e}")


def test_pcm16_streaming_pipeline_at_bench_size(tmp_path):
    """The end-to-end path bench.py times (PCM16 wav + float32 numpy mask files ->
    scripts/sptk/apply_adaptive_beamformer.py -> PCM16 wav) at 8-ch x 30 s: the files the
    streaming pipeline wrote against the oracle run on the SAME 16-bit samples."""
    import subprocess
    import sys
    import scipy.io.wavfile
    from conftest import ROOT, pcm16_rel_rms
    from setk_amd.libs import wavio
    td = str(tmp_path)
    C, N, nd, n = 8, 480000, 2, 6
    scenes = []
    for i in range(nd):
        mix, mask = synth(610 + i, C, N)
        pcm = wavio.float_to_pcm16(mix.T)
        wavio.write_pcm16(f"{td}/s{i}.wav", pcm, 16000)
        np.save(f"{td}/s{i}.npy", mask.astype(np.float32))
        scenes.append((pcm.astype(np.float32).T / np.float32(32768.0), mask))
    with open(f"{td}/wav.scp", "w") as ws, open(f"{td}/mask.scp", "w") as ms:
        for k in range(n):
            ws.write(f"u{k} {td}/s{k % nd}.wav\n")
            ms.write(f"u{k} {td}/s{k % nd}.npy\n")
    r = subprocess.run([sys.executable, f"{ROOT}/scripts/sptk/apply_adaptive_beamformer.py",
                        "--mask-format", "numpy", "--batch-utts", "4", f"{td}/wav.scp",
                        f"{td}/mask.scp", f"{td}/enh"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"Processed {n} utterances out of {n}" in r.stderr
    refs = [o.enhance_utterance(np.ascontiguousarray(s), m, kind="mvdr", gauge=True)
            for s, m in scenes]
    for k in range(n):
        sr, y = scipy.io.wavfile.read(f"{td}/enh/u{k}.wav")
        assert sr == 16000 and y.dtype == np.int16 and y.shape == refs[k % nd].shape
        err = pcm16_rel_rms(y, refs[k % nd])
        assert err < WAVE_TOL, (k, err)
    print("[pcm16 pipeline 8ch x 30s] ok")
