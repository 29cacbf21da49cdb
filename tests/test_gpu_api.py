"""
GPU tests of the drop-in surface: the python mirror of libs.utils /
libs.beamformer and the apply_adaptive_beamformer CLI, against the vectors the
unmodified reference produced (tests/golden) and against the oracle.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.io.wavfile

from conftest import ROOT, load_golden, rms, rel_rms, pcm16_rel_rms
from oracle import np_oracle as o
from oracle import make_golden as mg

pytestmark = pytest.mark.gpu

STFT_KW = dict(frame_len=512, frame_hop=256, window="hann", center=True)


def test_forward_inverse_stft_mirror():
    from setk_amd.libs import utils
    from setk_amd import _ffi
    g = load_golden("ref_stft.npz")
    for name, N, fl, hop, center, rp2, window in mg.STFT_CASES:
        x = g[f"{name}.x"]
        n_fft = o.nextpow2(fl) if rp2 else fl
        kw = dict(frame_len=fl, frame_hop=hop, center=center, window=window)
        S = utils.forward_stft(x, round_power_of_two=rp2, transpose=False, **kw)
        ref = g[f"{name}.S"]
        assert S.shape == ref.shape and S.dtype == np.complex64
        assert rel_rms(S, ref) < 1e-4, name
        St = utils.forward_stft(x, round_power_of_two=rp2, transpose=True, **kw)
        assert np.array_equal(St, S.T)
        mag = utils.forward_stft(x, round_power_of_two=rp2, transpose=False, apply_log=True, **kw)
        assert np.allclose(mag, np.log(np.maximum(np.abs(ref), utils.EPSILON)), atol=2e-3)
        tol = 1e-5 if center else 1e-4
        y = utils.inverse_stft(ref, transpose=False, **kw)
        assert y.dtype == np.float32 and rms(y, g[f"{name}.y"]) < tol, name
        yn = utils.inverse_stft(ref, transpose=False, norm=0.5, **kw)
        assert rms(yn, g[f"{name}.y_norm"]) < tol, name
    with pytest.raises(RuntimeError):
        utils.forward_stft(np.zeros((2, 4000), np.float32), frame_len=512)
    # an odd transform size has no inverse in the reference either (librosa.istft takes
    # n_fft = 2 (F - 1)); even sizes that are not powers of two run (tests/test_gpu_wide.py)
    with pytest.raises(ValueError):
        utils.forward_stft(np.zeros(4000, np.float32), frame_len=401, round_power_of_two=False)


@pytest.mark.parametrize("frame_len,hop", [(1024, 256), (256, 64), (2048, 512), (400, 160)])
def test_enhance_other_fft_sizes(frame_len, hop):
    """n_fft != 512 runs through the stand-alone operators (generic radix-2 FFT)."""
    from setk_amd.engine import BatchEnhancer
    mix, sp, nz = o.synth_utterance(90, 4, 24000, return_parts=True)
    kw = dict(frame_len=frame_len, frame_hop=hop, center=True, window="hann")
    mask = o.irm_mask(sp, nz, frame_len=frame_len, frame_hop=hop)
    for kind in ("mvdr", "pmwf-0", "gevd"):
        eng = BatchEnhancer(beamformer=kind, **kw)
        (wav, st), = eng.enhance([(mix, mask, None)])
        assert st == 0
        ref = o.enhance_utterance(mix, mask, kind=kind, gauge=True, **kw)
        assert wav.shape == ref.shape
        assert rms(wav, ref) / rms(ref) < 1e-3, (frame_len, kind)


def test_config0_roundtrip_on_device():
    """BASELINE configs[0] (1-ch 10 s STFT -> iSTFT) through the mirror."""
    from setk_amd.libs import utils
    x = o.synth_utterance(0, 1, 160000)[0]
    S = utils.forward_stft(x, transpose=False, **STFT_KW)
    assert S.shape == (257, 626)
    assert rel_rms(S, o.forward_stft(x, transpose=False, **STFT_KW)) < 1e-4
    y = utils.inverse_stft(S, transpose=False, **STFT_KW)
    assert y.shape[0] == 160000 and rms(y, x) / rms(x) < 1e-5


@pytest.mark.parametrize("case", mg.BF_CASES, ids=[c[0] for c in mg.BF_CASES])
def test_beamformer_mirror_against_reference_vectors(case):
    from setk_amd.libs import beamformer as B
    from test_oracle_golden import ORACLE_KINDS
    g = load_golden("ref_beamformer.npz")
    name = case[0]
    mix, mask = mg.bf_inputs(case)
    obs = o.multichannel_stft(mix, transpose=False, **STFT_KW)
    N, F, T = obs.shape
    Rs = B.compute_covar(obs, mask)
    Rn = B.compute_covar(obs, 1 - mask)
    assert Rs.shape == (F, N, N) and Rs.dtype == np.complex64
    assert rel_rms(Rs, g[f"{name}.Rs"]) < 1e-5 and rel_rms(Rn, g[f"{name}.Rn"]) < 1e-5
    pe = B.solve_pevd(g[f"{name}.Rs"])
    assert rel_rms(pe, o.fix_gauge_evd(g[f"{name}.pevd"])) < 1e-4
    pg = B.solve_pevd(g[f"{name}.Rs"], g[f"{name}.Rn"])
    assert pg.dtype == np.complex128
    assert rel_rms(pg, o.fix_gauge_gev(g[f"{name}.pgevd"], g[f"{name}.Rn"].astype(complex))) < 1e-4
    # rank-1 approximations and BAN on the reference's covariances
    for Rn_opt in (None, g[f"{name}.Rn"]):
        r1 = B.rank1_constraint(g[f"{name}.Rs"], Rn_opt)
        ref = o.rank1_constraint(g[f"{name}.Rs"], Rn_opt)
        assert rel_rms(r1, ref) < 1e-4
    w = o.mvdr_weight(g[f"{name}.Rs"], g[f"{name}.Rn"], gauge=True)
    assert rel_rms(B.do_ban(w, g[f"{name}.Rn"]), o.do_ban(w, g[f"{name}.Rn"])) < 1e-5
    classes = {
        "mvdr": (lambda: B.MvdrBeamformer(F), {}),
        "mvdr_ban": (lambda: B.MvdrBeamformer(F), dict(ban=True)),
        "gevd": (lambda: B.GevdBeamformer(F), {}),
        "gevd_ban": (lambda: B.GevdBeamformer(F), dict(ban=True)),
        "pmwf0": (lambda: B.PmwfBeamformer(F, beta=0), {}),
        "pmwf1": (lambda: B.PmwfBeamformer(F, beta=1), {}),
        "pmwf0_ref1": (lambda: B.PmwfBeamformer(F, beta=0, ref_channel=1), {}),
        "pmwf0_eig": (lambda: B.PmwfBeamformer(F, beta=0, rank1_appro="eig"), {}),
        "pmwf0_gev": (lambda: B.PmwfBeamformer(F, beta=0, rank1_appro="gev"), {}),
        "mpdr": (lambda: B.MpdrBeamformer(F), {}),
        "mpdr_whiten": (lambda: B.MpdrBeamformer(F, whiten=True), {}),
        "mpdr_whiten_ban": (lambda: B.MpdrBeamformer(F, whiten=True), dict(ban=True)),
    }
    norm = float(np.max(np.abs(mix)))
    for kind, (mk, kw) in classes.items():
        enh = mk().run(mask, obs, **kw)
        assert enh.shape == (F, T)
        wav = o.inverse_stft(enh, norm=norm, transpose=False, **STFT_KW)
        if kind.startswith("pmwf"):
            # gauge free: the vector stored by the unmodified reference
            ref = g[f"{name}.{kind}.wav"]
        else:
            # the stored vector carries LAPACK's per-bin sign (pinned against the
            # oracle in test_oracle_golden); compare under the declared gauge
            okind, okw = ORACLE_KINDS[kind]
            ref = o.inverse_stft(o.supervised_run(okind, mask, obs, gauge=True, **okw),
                                 norm=norm, transpose=False, **STFT_KW)
        err = rms(wav, ref) / rms(ref)
        assert err < 1e-3, (name, kind, err)
    # error behaviour of the reference
    with pytest.raises(ValueError):
        B.MvdrBeamformer(F).run(mask[:, :100], obs)
    with pytest.raises(ValueError):
        B.MvdrBeamformer(F).run(mask[:-1], obs)
    with pytest.raises(ValueError):
        B.Beamformer().beamform(np.zeros((F, N + 1), np.complex64), obs)
    with pytest.raises(np.linalg.LinAlgError):
        B.MvdrBeamformer(F).run(np.ones_like(mask), obs)  # noise mask == 0 -> singular
    with pytest.raises(RuntimeError):
        B.PmwfBeamformer(F, ref_channel=N).run(mask, obs)


def _write_inputs(td, g, fmt):
    import scipy.io.wavfile
    from setk_amd.libs.data_handler import ArchiveWriter
    keys = []
    with open(os.path.join(td, "wav.scp"), "w") as ws:
        for i in range(2):
            scipy.io.wavfile.write(os.path.join(td, f"u{i}.wav"), 16000, g[f"u{i}.pcm"])
            ws.write(f"u{i} {td}/u{i}.wav\n")
            keys.append(f"u{i}")
    if fmt == "numpy":
        with open(os.path.join(td, "mask.scp"), "w") as ms:
            for k in keys:
                np.save(os.path.join(td, f"{k}.npy"), g[f"{k}.mask"])
                ms.write(f"{k} {td}/{k}.npy\n")
    else:
        with ArchiveWriter(os.path.join(td, "mask.ark"), os.path.join(td, "mask.scp")) as w:
            for k in keys:
                w.write(k, g[f"{k}.mask"])
    return keys


@pytest.mark.parametrize("fmt", ["numpy", "kaldi"])
def test_cli_end_to_end_against_reference_cli(tmp_path, fmt):
    """Run the drop-in CLI as a subprocess on the inputs the reference CLI was
    run on (oracle/make_golden.py gen_cli) and compare the PCM16 files."""
    import scipy.io.wavfile
    from test_oracle_golden import per_bin_gain_fit
    g = load_golden("ref_cli.npz")
    td = str(tmp_path)
    keys = _write_inputs(td, g, fmt)
    for bf in ("mvdr", "gevd", "pmwf-0"):
        dst = os.path.join(td, bf)
        cmd = [sys.executable, os.path.join(ROOT, "scripts", "sptk", "apply_adaptive_beamformer.py"),
               "--frame-len", "512", "--frame-hop", "256", "--mask-format", fmt,
               "--beamformer", bf, os.path.join(td, "wav.scp"), os.path.join(td, "mask.scp"), dst]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        assert "Processed 2 utterances out of 2" in r.stderr
        assert f"Using offline {bf} beamformer" in r.stderr
        for k in keys:
            sr, y = scipy.io.wavfile.read(os.path.join(dst, k + ".wav"))
            ref = g[f"{k}.{bf}"]
            assert sr == 16000 and y.dtype == np.int16 and y.shape == ref.shape
            a = y.astype(np.float64) / 32768
            if bf == "pmwf-0":
                # gauge free: the file the reference CLI itself wrote
                b = ref.astype(np.float64) / 32768
            else:
                # the reference's file carries LAPACK's per-bin signs (pinned to
                # the oracle in test_oracle_golden::test_cli_goldens)
                samps = (g[f"{k}.pcm"].astype(np.float32) / 32768.0).T.copy()
                b = o.enhance_utterance(samps, g[f"{k}.mask"], kind=bf, gauge=True)
                b = np.rint(b.astype(np.float64) * 32767) / 32768
            assert rms(a, b) / rms(b) < 1e-3, (bf, k)
    # --skip-existing: with one output removed, a re-run writes that one and leaves the other
    dst = os.path.join(td, "mvdr")
    os.remove(os.path.join(dst, keys[0] + ".wav"))
    kept = os.path.join(dst, keys[1] + ".wav")
    before = (os.stat(kept).st_mtime_ns, open(kept, "rb").read())
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "sptk", "apply_adaptive_beamformer.py"),
           "--mask-format", fmt, "--skip-existing", "true", os.path.join(td, "wav.scp"),
           os.path.join(td, "mask.scp"), dst]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "--skip-existing: 1 of 2 utterances already" in r.stderr
    assert os.path.getsize(os.path.join(dst, keys[0] + ".wav")) > 44
    assert (os.stat(kept).st_mtime_ns, open(kept, "rb").read()) == before


def test_cli_options_vad_postmask_itf_online(tmp_path):
    import scipy.io.wavfile
    g = load_golden("ref_cli.npz")
    td = str(tmp_path)
    _write_inputs(td, g, "numpy")
    samps = (g["u0.pcm"].astype(np.float32) / 32768.0).T.copy()
    mask = g["u0.mask"]
    itf = np.random.default_rng(4).uniform(0.1, 0.9, size=mask.shape).astype(np.float32)
    with open(os.path.join(td, "itf.scp"), "w") as f:
        for k in ("u0", "u1"):
            m = itf if k == "u0" else np.random.default_rng(5).uniform(
                0.1, 0.9, size=g["u1.mask"].shape).astype(np.float32)
            np.save(os.path.join(td, f"itf_{k}.npy"), m)
            f.write(f"{k} {td}/itf_{k}.npy\n")
    script = os.path.join(ROOT, "scripts", "sptk", "apply_adaptive_beamformer.py")
    base = [sys.executable, script, "--mask-format", "numpy"]
    cases = {
        "vad": (["--vad-proportion", "0.9", "--post-masking", "true", "--ban", "true"],
                dict(vad_proportion=0.9, post_mask=True, ban=True)),
        "itf": (["--itf-mask", os.path.join(td, "itf.scp")], dict(itf_mask=itf)),
        # MPDR kinds with an interferer mask at the default transform size: mpdr never
        # reads the mask (fused path), mpdr-whiten takes Rn from it (stand-alone operators)
        "mpdr_itf": (["--beamformer", "mpdr", "--itf-mask", os.path.join(td, "itf.scp")],
                     dict(kind="mpdr", itf_mask=itf)),
        "mpdrw_itf": (["--beamformer", "mpdr-whiten", "--itf-mask", os.path.join(td, "itf.scp")],
                      dict(kind="mpdr-whiten", itf_mask=itf)),
    }
    for name, (extra, okw) in cases.items():
        okw = dict(okw)
        kind = okw.pop("kind", "mvdr")
        dst = os.path.join(td, name)
        r = subprocess.run(base + extra + [os.path.join(td, "wav.scp"),
                                           os.path.join(td, "mask.scp"), dst],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        sr, y = scipy.io.wavfile.read(os.path.join(dst, "u0.wav"))
        ref = o.enhance_utterance(samps, mask, kind=kind, gauge=True, **okw)
        assert pcm16_rel_rms(y, ref) < 1e-3, (name, pcm16_rel_rms(y, ref))
    # block-online mode runs (the reference's raises TypeError) and is sane
    dst = os.path.join(td, "online")
    r = subprocess.run(base + ["--online.chunk-size", "16", os.path.join(td, "wav.scp"),
                               os.path.join(td, "mask.scp"), dst],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "too small" in r.stderr
    r = subprocess.run(base + ["--online.chunk-size", "32", "--online.channels", "4",
                               os.path.join(td, "u0only.scp"), os.path.join(td, "mask.scp"), dst],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0  # missing scp
    with open(os.path.join(td, "u0only.scp"), "w") as f:
        f.write(f"u0 {td}/u0.wav\n")
    r = subprocess.run(base + ["--online.chunk-size", "32", "--online.channels", "4",
                               os.path.join(td, "u0only.scp"), os.path.join(td, "mask.scp"), dst],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    sr, y = scipy.io.wavfile.read(os.path.join(dst, "u0.wav"))
    assert y.shape[0] == 8192 and np.max(np.abs(y)) > 1000


def test_pcm16_device_ingest_is_bit_identical(tmp_path):
    """setk_pcm16_to_float == read_wav's host decode (int16 / 32768, C x N), and the
    batch engine fed with the stored frames returns the samples it returns for
    the host-decoded float audio."""
    import torch
    from setk_amd import _ffi
    from setk_amd.engine import BatchEnhancer, Pcm16Frames
    from setk_amd.libs import wavio
    from setk_amd.libs.data_handler import WaveReader
    rng = np.random.default_rng(5)
    ctx = _ffi.default_context()
    for C, N in [(1, 777), (3, 4099), (8, 16000)]:
        pcm = rng.integers(-32768, 32768, size=(N, C), dtype=np.int64).astype(np.int16)
        out = np.empty((C, N), dtype=np.float32)
        ctx.pcm16_to_float(pcm, C, N, out)                       # host pointers (staged)
        assert np.array_equal(out, pcm.T.astype(np.float32) / np.float32(32768))
        dout = torch.empty((C, N), dtype=torch.float32, device="cuda")
        ctx.pcm16_to_float(torch.from_numpy(pcm).cuda(), C, N, dout)  # device pointers
        assert np.array_equal(dout.cpu().numpy(), out)

    mix, sp, nz = o.synth_utterance(3, 6, 24000, return_parts=True)
    mask = o.irm_mask(sp, nz)
    frames = wavio.float_to_pcm16(mix.T)
    wavio.write_pcm16(str(tmp_path / "u.wav"), frames, 16000)
    (tmp_path / "wav.scp").write_text(f"u {tmp_path}/u.wav\n")
    reader = WaveReader(str(tmp_path / "wav.scp"), sr=16000)
    raw = reader.read_pcm16("u")
    assert raw.dtype == np.int16 and raw.shape == (24000, 6)
    host = reader.read("u")
    eng = BatchEnhancer(beamformer="mvdr", pcm16=True)
    (w_dev, st_dev), (w_host, st_host) = eng.enhance([(Pcm16Frames(raw), mask, None),
                                                      (host, mask, None)])
    assert st_dev == 0 and st_host == 0
    assert np.array_equal(w_dev, w_host)


def _consumer_inputs(td, g):
    from setk_amd.libs import wavio
    with open(os.path.join(td, "wav.scp"), "w") as ws, open(os.path.join(td, "mask.scp"), "w") as ms, \
            open(os.path.join(td, "beam.scp"), "w") as bs:
        for k in ("u0", "u1"):
            wavio.write_pcm16(os.path.join(td, f"{k}.wav"), g[f"{k}.pcm"], 16000)
            np.save(os.path.join(td, f"{k}.npy"), g[f"{k}.mask"])
            ws.write(f"{k} {td}/{k}.wav\n")
            ms.write(f"{k} {td}/{k}.npy\n")
            bs.write(f"{k} {int(g[f'{k}.beam'])}\n")
    np.save(os.path.join(td, "w.npy"), g["weights"])


def test_consumers_fixed_beamformer_and_directional_feats(tmp_path):
    """the drop-in apply_fixed_beamformer.py / compute_df_on_mask.py
    (and the library calls under them) against what the unmodified reference
    CLIs produced for the same inputs (tests/golden/ref_consumers.npz)."""
    from setk_amd.engine import FixedBatchBeamformer, Pcm16Frames
    from setk_amd.libs import beamformer as B, spatial as S, utils as U, wavio
    from setk_amd.libs.data_handler import ScriptReader
    g = load_golden("ref_consumers.npz")
    td = str(tmp_path)
    _consumer_inputs(td, g)
    kw = dict(frame_len=512, frame_hop=256, center=True, window="hann")
    pairs = [(0, 1), (1, 3), (0, 2)]

    # ---- library level ----
    eng = FixedBatchBeamformer(g["weights"], pcm16=True)
    outs = eng.run([(Pcm16Frames(g[f"{k}.pcm"]), int(g[f"{k}.beam"])) for k in ("u0", "u1")])
    for k, pcm in zip(("u0", "u1"), outs):
        ref = g[f"{k}.fixed"]
        assert pcm.dtype == np.int16 and pcm.shape == ref.shape
        assert rms(pcm.astype(np.float64), ref.astype(np.float64)) / rms(ref.astype(np.float64)) < 1e-3
        samps = g[f"{k}.pcm"].T.astype(np.float32) / np.float32(32768)
        obs = np.stack([U.forward_stft(c, round_power_of_two=True, transpose=False, **kw)
                        for c in samps])                                    # M x F x T
        sv = B.solve_pevd(B.compute_covar(obs, np.minimum(g[f"{k}.mask"], 1)))
        df = S.directional_feats(obs, sv.T, df_pair=pairs)
        assert df.shape == g[f"{k}.df"].shape and df.dtype == np.float32
        assert np.max(np.abs(df - g[f"{k}.df"])) < 2e-3
        assert np.max(np.abs(df - o.directional_feats(obs, sv.T, df_pair=pairs))) < 1e-4
    with pytest.raises(ValueError):
        S.directional_feats(obs, sv.T[:2], df_pair=pairs)
    # the batched, device-resident engine behind the CLI: samples + masks in, features out
    from setk_amd.engine import BatchDirectionalFeatures
    dfe = BatchDirectionalFeatures(pairs, **kw)
    res = dfe.run([(Pcm16Frames(g["u0.pcm"]), g["u0.mask"]),
                   (g["u1.pcm"].T.astype(np.float32) / np.float32(32768), np.ascontiguousarray(g["u1.mask"].T))])
    for k, (feats_k, code) in zip(("u0", "u1"), res):
        assert code == 0 and feats_k.dtype == np.float32 and feats_k.shape == g[f"{k}.df"].shape
        assert np.max(np.abs(feats_k - g[f"{k}.df"])) < 2e-3
    # several chunks: two lanes (handles, streams), each driven by its own thread -- the same
    # features whatever the chunking
    many = BatchDirectionalFeatures(pairs, chunk_utts=4, **kw)
    two = [(Pcm16Frames(g["u0.pcm"]), g["u0.mask"]), (Pcm16Frames(g["u1.pcm"]), g["u1.mask"])]
    ref2 = many.run(two)
    for rep in range(2):
        got = many.run(two * 9)
        assert len(got) == 18
        for i, (f_i, code) in enumerate(got):
            assert code == 0 and np.array_equal(f_i, ref2[i % 2][0])
    many.close()
    # a non-finite sample: numpy's eigh raises LinAlgError in the reference; here a status
    bad = g["u0.pcm"].T.astype(np.float32) / np.float32(32768)
    bad[1, 1000] = np.nan
    (none, code), (again, c2) = dfe.run([(bad, g["u0.mask"]), (Pcm16Frames(g["u0.pcm"]), g["u0.mask"])])
    assert none is None and code != 0 and c2 == 0 and np.array_equal(again, res[0][0])
    with pytest.raises(ValueError):
        BatchDirectionalFeatures([(0, 7)], **kw).run([(Pcm16Frames(g["u0.pcm"]), g["u0.mask"])])
    dfe.close()
    # another transform size: the stand-alone operators behind the same interface
    kw1k = dict(frame_len=1024, frame_hop=256, center=True, window="hann")
    samps1 = g["u1.pcm"].T.astype(np.float32) / np.float32(32768)
    obs1k = np.stack([o.forward_stft(c, round_power_of_two=True, transpose=False, **kw1k) for c in samps1])
    m1k = np.random.default_rng(3).uniform(0.1, 0.9, size=(obs1k.shape[2], 513)).astype(np.float32)
    (f1k, code1k), = BatchDirectionalFeatures(pairs, **kw1k).run([(samps1, m1k)])
    sv1k = o.solve_pevd(o.compute_covar(obs1k, m1k), gauge=True)
    assert code1k == 0 and np.max(np.abs(f1k - o.directional_feats(obs1k, sv1k.T, df_pair=pairs))) < 2e-3

    # ---- command line level ----
    py = [sys.executable]
    r = subprocess.run(py + [os.path.join(ROOT, "scripts", "sptk", "apply_fixed_beamformer.py"),
                             "--beam", f"{td}/beam.scp", f"{td}/wav.scp", f"{td}/w.npy", f"{td}/fixed"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Processed 2 utterances" in r.stderr
    r2 = subprocess.run(py + [os.path.join(ROOT, "scripts", "sptk", "compute_df_on_mask.py"),
                              "--mask-format", "numpy", "--df-pair", "0,1;1,3;0,2", "--scp",
                              f"{td}/df.scp", f"{td}/wav.scp", f"{td}/mask.scp", f"{td}/df.ark"],
                        capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    feats = ScriptReader(f"{td}/df.scp")
    for k in ("u0", "u1"):
        y, sr = wavio.read(f"{td}/fixed/{k}.wav", dtype="int16")
        ref = g[f"{k}.fixed"].astype(np.float64)
        assert sr == 16000 and rms(y.astype(np.float64), ref) / rms(ref) < 1e-3
        assert np.max(np.abs(feats[k] - g[f"{k}.df"])) < 2e-3
    # a single F x M weight needs no beam table (the reference stumbles here)
    np.save(f"{td}/w1.npy", g["weights"][1])
    r3 = subprocess.run(py + [os.path.join(ROOT, "scripts", "sptk", "apply_fixed_beamformer.py"),
                              f"{td}/wav.scp", f"{td}/w1.npy", f"{td}/fixed1"],
                        capture_output=True, text=True, timeout=600)
    assert r3.returncode == 0, r3.stderr[-2000:]
    # other transform size: the stand-alone operator path of the same engine
    rng = np.random.default_rng(9)
    w1k = ((rng.standard_normal((2, 513, 4)) + 1j * rng.standard_normal((2, 513, 4))) / 4).astype(
        np.complex64)
    eng2 = FixedBatchBeamformer(w1k, frame_len=1024, frame_hop=256)
    samps = g["u1.pcm"].T.astype(np.float32) / np.float32(32768)
    (wav,) = eng2.run([(samps, 1)])
    kw2 = dict(frame_len=1024, frame_hop=256, center=True, window="hann")
    obs = np.stack([o.forward_stft(c, round_power_of_two=True, transpose=False, **kw2)
                    for c in samps])
    ref = o.inverse_stft(o.beamform(w1k[1], obs), norm=float(np.max(np.abs(samps))),
                         transpose=False, **kw2)
    assert wav.shape == ref.shape and rms(wav, ref) / rms(ref) < 1e-3


def test_streaming_pipeline_equals_batch_path(tmp_path):
    """The streaming host pipeline (pinned slabs, payloads read as stored, batched
    device ingest, several batches in flight) writes the same wavs as the
    one-batch-at-a-time path, for every kind of scp entry: PCM16 wav + float32
    npy (read straight into the slab), Kaldi FM archive masks, a transposed (F x T)
    mask, a float64 mask, a 32-bit float wav, a per-channel glob, ragged lengths,
    different channel counts, a missing mask (skipped)."""
    import scipy.io.wavfile
    from setk_amd.libs import wavio
    from setk_amd.libs.data_handler import ArchiveWriter
    td = str(tmp_path)
    rng = np.random.default_rng(5)
    lens = [16000, 9000, 23456, 16000, 5000, 12000, 8000, 30000, 7000]
    chans = [4, 4, 4, 4, 4, 4, 2, 4, 4]
    refs = {}
    os.makedirs(f"{td}/m")
    with open(f"{td}/wav.scp", "w") as ws, open(f"{td}/mask.scp", "w") as ms, \
            ArchiveWriter(f"{td}/masks.ark", f"{td}/ark.scp") as aw:
        for i, (n, c) in enumerate(zip(lens, chans)):
            mix, sp, nz = o.synth_utterance(500 + i, c, n, return_parts=True)
            mask = (0.1 + 0.8 * o.irm_mask(sp, nz)).astype(np.float32)
            pcm = wavio.float_to_pcm16(mix.T)
            key = f"utt{i}"
            if i == 3:    # IEEE float wav: decoded on the host
                scipy.io.wavfile.write(f"{td}/{key}.wav", 16000, mix.T.copy())
                samps = mix
            elif i == 5:  # per-channel files behind a glob
                for ch in range(c):
                    wavio.write_pcm16(f"{td}/{key}.CH{ch}.wav", pcm[:, ch], 16000)
                samps = pcm.T.astype(np.float32) / 32768
            else:
                wavio.write_pcm16(f"{td}/{key}.wav", pcm, 16000)
                samps = pcm.T.astype(np.float32) / 32768
            ws.write(f"{key} {td}/{key}.CH*.wav\n" if i == 5 else f"{key} {td}/{key}.wav\n")
            stored = mask
            if i == 1:
                stored = np.ascontiguousarray(mask.T)       # F x T
            if i == 2:
                stored = mask.astype(np.float64)
            if i != 8:                                       # utt8 has no mask: skipped
                np.save(f"{td}/m/{key}.npy", stored)
                ms.write(f"{key} {td}/m/{key}.npy\n")
                aw.write(key, stored.astype(np.float32))
                refs[key] = o.enhance_utterance(samps.astype(np.float32), mask, kind="mvdr",
                                                gauge=True)
    script = os.path.join(ROOT, "scripts", "sptk", "apply_adaptive_beamformer.py")

    def run(dst, fmt, mask_scp, *extra, env=None):
        r = subprocess.run([sys.executable, script, "--mask-format", fmt, "--batch-utts", "3",
                            *extra, f"{td}/wav.scp", mask_scp, dst],
                           capture_output=True, text=True, timeout=600,
                           env=None if env is None else dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-3000:]
        assert "Processed 8 utterances out of 9" in r.stderr
        assert r.stderr.count("Processing utterance") == 8
        return {k: scipy.io.wavfile.read(os.path.join(dst, k + ".wav"))[1] for k in refs}

    a = run(f"{td}/pipe", "numpy", f"{td}/mask.scp", "--profile", f"{td}/prof.json")
    b = run(f"{td}/batch", "numpy", f"{td}/mask.scp", "--pipeline", "false")
    c = run(f"{td}/kaldi", "kaldi", f"{td}/ark.scp")
    d = run(f"{td}/zc", "numpy", f"{td}/mask.scp", "--zero-copy", "true", "--profile",
            f"{td}/prof_zc.json")
    # the read stage's other forms (the default is the library's native reader pool, one call per
    # batch): the interpreter's reader threads with a mapping (rounds 3 - 4) or preadv, and the
    # native pool with every payload above its mapping threshold / per-payload copies up
    e = run(f"{td}/mm", "numpy", f"{td}/mask.scp", env={"SETK_READ_MODE": "mmap"})
    f = run(f"{td}/pr", "numpy", f"{td}/mask.scp", env={"SETK_READ_MODE": "preadv"})
    g = run(f"{td}/nm", "numpy", f"{td}/mask.scp", "--h2d", "payload", env={"SETK_MMAP_MIN_KB": "4"})
    # batches closed by bytes (--batch-mb, round 6) instead of by count: other batch boundaries
    h = run(f"{td}/mb", "numpy", f"{td}/mask.scp", "--batch-utts", "32", "--batch-mb", "1", "--profile",
            f"{td}/prof_mb.json")
    assert not os.path.exists(f"{td}/pipe/utt8.wav")
    for k, ref in refs.items():
        assert a[k].dtype == np.int16 and np.array_equal(a[k], b[k]), k
        assert np.array_equal(a[k], c[k]) and np.array_equal(a[k], d[k]), k
        assert np.array_equal(a[k], e[k]) and np.array_equal(a[k], f[k]) and np.array_equal(a[k], g[k]), k
        assert pcm16_rel_rms(a[k], ref) < 1e-3, (k, pcm16_rel_rms(a[k], ref))
        assert pcm16_rel_rms(h[k], ref) < 1e-3 and np.max(np.abs(h[k].astype(np.int32) - a[k].astype(np.int32))) <= 1, k
    import json
    prof = json.load(open(f"{td}/prof.json"))
    mb = json.load(open(f"{td}/prof_mb.json"))["stages"]
    assert mb["batch_mb"] == 1 and mb["batches"] >= 3 and "marks" in prof   # (1 MB closes the first batch after six utterances; the 2-channel one splits the rest)
    assert prof["mode"] == "pipeline" and prof["utts"] == 8 and prof["stages"]["batches"] >= 3
    assert prof["stages"]["read_mode"] == "native"
    # --zero-copy: PCM16 wavs and float32 C-ordered masks leave the page cache without a host copy
    assert json.load(open(f"{td}/prof_zc.json"))["stages"]["zero_copy_payloads"] >= 8
    assert prof["stages"]["zero_copy_payloads"] == 0


def _cm_body(kind, mat, vmin, vrange, rng):
    """A Kaldi CompressedMatrix body holding (an approximation of) `mat`: what follows the archive's
    16-byte global header (formats of the reference's libs/kaldi_io.py:248-292)."""
    rows, cols = mat.shape
    u = np.clip((mat - vmin) / vrange, 0.0, 1.0)
    if kind == "CM2":
        return np.rint(u * 65535).astype("<u2").tobytes()
    if kind == "CM3":
        return np.rint(u * 255).astype(np.uint8).tobytes()
    # CM: per column four sorted uint16 percentiles, then one byte per element, column-major
    pch = np.sort(rng.integers(0, 65536, size=(cols, 4)).astype("<u2"), axis=1)
    body = rng.integers(0, 256, size=(cols, rows)).astype(np.uint8)
    body[:, :8] = np.array([0, 63, 64, 65, 192, 193, 194, 255], dtype=np.uint8)[None, :min(8, rows)]
    return pch.tobytes() + body.tobytes()


def test_kaldi_compressed_masks_decoded_on_the_device(tmp_path):
    """SURVEY 8f-3 "Kaldi compressed-matrix (CM) mask decode": setk_kaldi_cm_decode_batch against the
    host decode (libs/kaldi_io.uncompress, itself pinned on the reference's own `uncompress`,
    kaldi_io.py:248-292) -- bit for bit, all three formats, both orientations, one launch; the
    reference-decoded golden bodies; and the streaming command line on a compressed mask archive:
    the same wave files as with the host decode, with a quarter of the mask bytes shipped."""
    import json
    import struct
    import torch
    from setk_amd import _ffi
    from setk_amd.libs import kaldi_io, wavio
    rng = np.random.default_rng(11)
    ctx = _ffi.default_context()
    g = load_golden("ref_kaldi.npz")
    head = (float(g["cm.head"][0]), float(g["cm.head"][1]), 9, 6)
    cases = [("CM", g["cm.pch"].tobytes() + g["cm.body"].tobytes(), head, g["cm.decoded"]),
             ("CM2", g["cm2.body"].tobytes(), head, g["cm2.decoded"]),
             ("CM3", g["cm3.body"].tobytes(), head, g["cm3.decoded"])]
    for kind in ("CM", "CM2", "CM3"):
        for rows, cols in ((63, 257), (257, 63), (1, 5), (300, 257)):
            vmin, vrange = float(np.float32(rng.uniform(-2, 0))), float(np.float32(rng.uniform(0.5, 3)))
            mat = rng.uniform(vmin, vmin + vrange, size=(rows, cols))
            cases.append((kind, _cm_body(kind, mat, vmin, vrange, rng), (vmin, vrange, rows, cols), None))
    items, keep, want = [], [], []
    for kind, body, hd, golden in cases:
        for tr in (False, True):
            src = torch.from_numpy(np.frombuffer(body, dtype=np.uint8).copy()).cuda()
            rows, cols = hd[2], hd[3]
            dst = torch.full((cols, rows) if tr else (rows, cols), np.nan, dtype=torch.float32, device="cuda")
            host = kaldi_io.uncompress(body, kind, hd).astype(np.float32)
            if golden is not None:
                assert np.allclose(host, golden, atol=1e-6)
            items.append((kind, hd[0], hd[1], rows, cols, tr, src.data_ptr(), dst.data_ptr()))
            keep.append((src, dst))
            want.append(np.ascontiguousarray(host.T) if tr else host)
    ctx.kaldi_cm_decode_batch(items)
    torch.cuda.synchronize()
    for (src, dst), w, it in zip(keep, want, items):
        assert np.array_equal(dst.cpu().numpy(), w), it[:6]
    with pytest.raises(ValueError):     # SETK_ERR_INVALID: API misuse <-> ValueError
        ctx.kaldi_cm_decode_batch([("CM2", 0.0, 1.0, 0, 4, False, keep[0][0].data_ptr(), keep[0][1].data_ptr())])

    # ---- the command line on an archive of compressed masks ----
    td = str(tmp_path)
    lens, fmts = [16000, 9000, 23456, 12000], ["CM2", "CM3", "CM2", "CM"]
    refs = {}
    with open(f"{td}/wav.scp", "w") as ws, open(f"{td}/masks.ark", "wb") as ark, open(f"{td}/mask.scp", "w") as ms:
        for i, (n, kind) in enumerate(zip(lens, fmts)):
            mix, sp, nz = o.synth_utterance(700 + i, 4, n, return_parts=True)
            mask = (0.1 + 0.8 * o.irm_mask(sp, nz)).astype(np.float32)
            stored = np.ascontiguousarray(mask.T) if i == 2 else mask       # utt2: F x T
            body = _cm_body(kind, stored.astype(np.float64), 0.0, 1.0, rng)
            hd = (0.0, 1.0, stored.shape[0], stored.shape[1])
            key = f"utt{i}"
            pcm = wavio.float_to_pcm16(mix.T)
            wavio.write_pcm16(f"{td}/{key}.wav", pcm, 16000)
            ws.write(f"{key} {td}/{key}.wav\n")
            ark.write(key.encode() + b" ")
            ms.write(f"{key} {td}/masks.ark:{ark.tell()}\n")
            ark.write(b"\0B" + kind.encode() + b" " + struct.pack("<ffii", *hd) + body)
            dec = kaldi_io.uncompress(body, kind, hd).astype(np.float32)
            refs[key] = o.enhance_utterance(pcm.T.astype(np.float32) / 32768, dec.T if i == 2 else dec,
                                            kind="mvdr", gauge=True)
    script = os.path.join(ROOT, "scripts", "sptk", "apply_adaptive_beamformer.py")
    outs = {}
    for mode in ("1", "0"):
        r = subprocess.run([sys.executable, script, "--mask-format", "kaldi", "--profile", f"{td}/prof{mode}.json",
                            f"{td}/wav.scp", f"{td}/mask.scp", f"{td}/enh{mode}"], capture_output=True, text=True,
                           timeout=600, env=dict(os.environ, SETK_CM_DEVICE=mode))
        assert r.returncode == 0, r.stderr[-3000:]
        assert "Processed 4 utterances out of 4" in r.stderr
        outs[mode] = {k: scipy.io.wavfile.read(f"{td}/enh{mode}/{k}.wav")[1] for k in refs}
    for k, ref in refs.items():
        assert np.array_equal(outs["1"][k], outs["0"][k]), k            # device decode == host decode
        assert pcm16_rel_rms(outs["1"][k], ref) < 1e-3, (k, pcm16_rel_rms(outs["1"][k], ref))
    b1 = json.load(open(f"{td}/prof1.json"))["stages"]["bytes_in"]
    b0 = json.load(open(f"{td}/prof0.json"))["stages"]["bytes_in"]
    assert b1 < b0      # the archive's bytes went up, not their float32 expansion


def test_rccl_comm_through_the_c_abi_single_rank():
    """setk_comm_* (csrc/comm.hip): RCCL reached from the library itself, no torch.distributed.
    A one-GPU box can only form a communicator of one rank (RCCL refuses two ranks on one
    device); that still exercises dlopen(librccl), ncclGetUniqueId, ncclCommInitRank, an
    all-reduce on the device and the teardown.  The multi-rank control flow around it is the CPU
    test test_shard_without_torch (TCP star) and the 2 / 8-rank command-line tests."""
    import ctypes
    from setk_amd import _ffi
    lib = _ffi.load_library()
    lib.setk_comm_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_char_p,
                                     ctypes.c_int, ctypes.c_int]
    lib.setk_comm_allreduce_f64.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int,
                                            ctypes.c_int]
    lib.setk_comm_barrier.argtypes = [ctypes.c_void_p]
    lib.setk_comm_destroy.argtypes = [ctypes.c_void_p]
    lib.setk_comm_last_error.restype = ctypes.c_char_p
    uid = ctypes.create_string_buffer(128)
    assert lib.setk_comm_unique_id(uid) == 0, lib.setk_comm_last_error()
    assert any(uid.raw)
    comm = ctypes.c_void_p()
    assert lib.setk_comm_create(ctypes.byref(comm), 0, uid.raw, 0, 1) == 0, lib.setk_comm_last_error()
    vals = (ctypes.c_double * 3)(1.5, -2.0, 7.0)
    assert lib.setk_comm_allreduce_f64(comm, vals, 3, 0) == 0, lib.setk_comm_last_error()
    assert list(vals) == [1.5, -2.0, 7.0]
    assert lib.setk_comm_allreduce_f64(comm, vals, 3, 1) == 0 and list(vals) == [1.5, -2.0, 7.0]
    assert lib.setk_comm_barrier(comm) == 0
    assert lib.setk_comm_allreduce_f64(comm, vals, 65, 0) != 0   # more values than the call takes
    assert lib.setk_comm_destroy(comm) == 0
    # and through the Shard of the command lines, as a launcher with one rank would set it up
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from setk_amd import _ffi; _ffi.set_torch_free()\n"
            "from setk_amd.dist import Shard\n"
            "s = Shard(); s.barrier(); print(s.sum_counts([3, 4]), 'torch' in sys.modules)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "[3, 4] False" in r.stdout, r.stderr[-1500:]


def test_shard_falls_back_to_tcp_when_librccl_cannot_be_loaded(tmp_path):
    """On a GPU host (/dev/kfd present) with no loadable librccl (SETK_RCCL_LIB points nowhere) the
    ranks agree on the TCP star instead of crashing or waiting (round-5 advice, comm.hip:55)."""
    import json
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = ("import sys, json; sys.path.insert(0, %r)\n"
            "from setk_amd import _ffi; _ffi.set_torch_free()\n"
            "from setk_amd.dist import Shard\n"
            "sh = Shard(); sh.barrier(); tot = sh.sum_counts([sh.rank + 1, 5])\n"
            "print(json.dumps(dict(backend=sh.backend, tot=tot))); sh.close()\n" % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SETK_RCCL_LIB="/nonexistent/librccl.so.1")
        env.pop("SETK_DIST_BACKEND", None)
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-2000:]
        rec = json.loads(out.strip().splitlines()[-1])
        assert rec["backend"] == "tcp" and rec["tot"] == [3.0, 10.0]


def test_cli_strict_reference_skips_what_the_reference_skips(tmp_path):
    """--strict-reference true: the command line logs 'Raise linalg error' and writes no file for
    exactly the utterances on which the reference's numpy.linalg.solve raises (a duplicated
    channel: an exactly singular noise covariance, tests/golden/ref_skipset.json) -- and writes
    every file by default (the regularised solve), for the same table."""
    from setk_amd.libs import wavio
    td = str(tmp_path)
    cases = o.skipset_cases()
    os.makedirs(f"{td}/m")
    with open(f"{td}/wav.scp", "w") as ws, open(f"{td}/mask.scp", "w") as ms:
        for key in ("plain", "dup-channel", "channel-x0.3", "zero-channel"):
            samps, mask = cases[key]
            pcm = wavio.float_to_pcm16(samps.T)
            if key == "dup-channel":
                pcm[:, 1] = pcm[:, 0]          # (exact after quantisation too)
            k = key.replace(".", "_")
            wavio.write_pcm16(f"{td}/{k}.wav", pcm, 16000)
            np.save(f"{td}/m/{k}.npy", mask.astype(np.float32))
            ws.write(f"{k} {td}/{k}.wav\n")
            ms.write(f"{k} {td}/m/{k}.npy\n")
    script = os.path.join(ROOT, "scripts", "sptk", "apply_adaptive_beamformer.py")

    def run(dst, *extra):
        r = subprocess.run([sys.executable, script, "--mask-format", "numpy", *extra,
                            f"{td}/wav.scp", f"{td}/mask.scp", dst], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        return sorted(os.listdir(dst)), r.stderr

    files, err = run(f"{td}/default")
    assert files == ["channel-x0_3.wav", "dup-channel.wav", "plain.wav", "zero-channel.wav"]
    assert "Processed 4 utterances out of 4" in err
    files, err = run(f"{td}/strict", "--strict-reference", "true")
    assert files == ["channel-x0_3.wav", "plain.wav"], files
    assert err.count("Raise linalg error") == 2 and "Processed 2 utterances out of 4" in err
    for gevd_dir in ("gev",):
        files, err = run(f"{td}/{gevd_dir}", "--strict-reference", "true", "--beamformer", "gevd")
        assert len(files) == 4 and "Raise linalg error" not in err   # the reference's GEV never raises
