"""
GPU parity of the geometry beamformers (DS / SD, libs/beamformer.py:
133-212, 343-512; apply_classic / apply_ds / apply_sd_beamformer.py): the python mirror
and the drop-in CLIs against the reference's stored doc outputs and against files the
unmodified reference CLIs wrote (tests/golden/ref_classic.npz, oracle/make_golden.py
gen_classic), with the PCM16 floor on both sides.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden, pcm16_rel_rms, rel_rms
from oracle import np_oracle as o
from test_oracle_golden import CLASSIC_RUNS, classic_online_doas

pytestmark = pytest.mark.gpu
STFT_KW = dict(frame_len=512, frame_hop=256, window="hann", center=True)


def _cli(name, args, stdin=None):
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "sptk", name)] + args
    r = subprocess.run(cmd, input=stdin, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return r


def test_mirror_classes_against_oracle():
    from setk_amd.libs import beamformer as B
    mix = o.synth_utterance(310, 4, 16000)
    obs = o.multichannel_stft(mix, transpose=False, **STFT_KW)
    cases = [
        (B.LinearDSBeamformer((0.0, 0.04, 0.08, 0.12)), dict(kind="ds", geometry="linear",
                                                             linear_topo=(0.0, 0.04, 0.08, 0.12)), 45.0),
        (B.LinearSDBeamformer((0.0, 0.05, 0.1, 0.2)), dict(kind="sd", geometry="linear",
                                                           linear_topo=(0.0, 0.05, 0.1, 0.2)), 120.0),
        (B.CircularDSBeamformer(0.05, 4), dict(kind="ds", geometry="circular", num_arounded=4), 250.0),
        (B.CircularSDBeamformer(0.04, 3, center=True),
         dict(kind="sd", geometry="circular", num_arounded=3, radius=0.04, circular_center=True), 33.0),
    ]
    for bf, okw, doa in cases:
        w = bf.weight(doa, 257, c=343, sr=16000)
        wo = o.classic_weight(doa=doa, num_bins=257, c=343, sr=16000, **okw)
        assert np.abs(w - wo).max() <= 1e-12 * np.abs(wo).max()
        enh = bf.run(doa, obs, c=343, sr=16000)
        assert enh.shape == (257, obs.shape[2])
        assert rel_rms(enh, o.beamform(wo, obs)) < 1e-5
    with pytest.raises(ValueError):
        B.CircularDSBeamformer(0.05, 6).run(0.0, obs)


def test_doc_ds_sd_clis(tmp_path):
    """doc/fixed_beamformer/README.md command lines (stdin scp, --utt2doa) against the
    stored ds.wav / sd.wav."""
    import scipy.io.wavfile
    g = load_golden("ref_classic.npz")
    td = str(tmp_path)
    scipy.io.wavfile.write(os.path.join(td, "egs.wav"), 16000, g["doc.egs"])
    with open(os.path.join(td, "doa.scp"), "w") as f:
        f.write("egs 100\n")
    common = ["--frame-len", "512", "--frame-hop", "256", "--geometry", "circular",
              "--circular-around", "4", "--circular-radius", "0.05", "--utt2doa",
              os.path.join(td, "doa.scp"), "--sr", "16000", "--speed", "340"]
    r = _cli("apply_ds_beamformer.py", common + ["-", os.path.join(td, "ds")],
             stdin=f"egs {td}/egs.wav\n")
    assert "Processed 1 utterances over 1" in r.stderr
    _cli("apply_sd_beamformer.py", common + ["--normalize", "true", "-", os.path.join(td, "sd")],
         stdin=f"egs {td}/egs.wav\n")
    for name in ("ds", "sd"):
        sr, y = scipy.io.wavfile.read(os.path.join(td, name, "egs.wav"))
        ref = g[f"doc.{name}"]
        assert sr == 16000 and y.dtype == np.int16 and y.shape == ref.shape
        err = rel_rms(y.astype(np.float64), ref.astype(np.float64))
        assert err < 1e-3, (name, err)


def test_classic_cli_against_reference_cli(tmp_path):
    import scipy.io.wavfile
    g = load_golden("ref_classic.npz")
    td = str(tmp_path)
    with open(os.path.join(td, "wav.scp"), "w") as ws, open(os.path.join(td, "doa.scp"), "w") as ds:
        for k in ("c0", "c1"):
            scipy.io.wavfile.write(os.path.join(td, f"{k}.wav"), 16000, g[f"{k}.pcm"])
            ws.write(f"{k} {td}/{k}.wav\n")
            ds.write(f"{k} " + " ".join(str(d) for d in classic_online_doas(g[f"{k}.pcm"].shape[0]))
                     + "\n")
    flags = {
        "ds.circular": ["--beamformer", "ds", "--geometry", "circular", "--circular-around", "4",
                        "--doa", "77.5"],
        "sd.circular.norm": ["--beamformer", "sd", "--geometry", "circular", "--circular-around", "4",
                             "--doa", "200", "--normalize", "true"],
        "ds.linear": ["--beamformer", "ds", "--geometry", "linear", "--linear-topo",
                      "0.0,0.04,0.08,0.12", "--doa", "60"],
        "sd.linear": ["--beamformer", "sd", "--geometry", "linear", "--linear-topo",
                      "0.0,0.05,0.1,0.2", "--doa", "135"],
        "sd.center.norm": ["--beamformer", "sd", "--geometry", "circular", "--circular-around", "3",
                           "--circular-center", "true", "--doa", "10", "--normalize", "true"],
        "ds.online": ["--beamformer", "ds", "--geometry", "circular", "--circular-around", "4",
                      "--chunk-len", "16", "--utt2doa", os.path.join(td, "doa.scp")],
    }
    assert set(flags) == set(CLASSIC_RUNS)
    for name, fl in flags.items():
        dst = os.path.join(td, name)
        _cli("apply_classic_beamformer.py", fl + [os.path.join(td, "wav.scp"), dst])
        for k in ("c0", "c1"):
            sr, y = scipy.io.wavfile.read(os.path.join(dst, k + ".wav"))
            ref = g[f"{k}.{name}"]
            assert y.shape == ref.shape and y.dtype == np.int16
            err = rel_rms(y.astype(np.float64), ref.astype(np.float64))
            assert err < 1e-3, (name, k, err)
    # an invalid direction is logged and skipped, like the reference (:94-96)
    r = _cli("apply_classic_beamformer.py",
             ["--geometry", "linear", "--linear-topo", "0,0.1,0.2,0.3", "--doa", "181",
              os.path.join(td, "wav.scp"), os.path.join(td, "bad")])
    assert "Invalid doa 181.00" in r.stderr and "Processed 0 utterances over 2" in r.stderr
