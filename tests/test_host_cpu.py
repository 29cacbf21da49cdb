"""
CPU-side tests (no GPU): the C-ABI library loads and exports every symbol of
include/setk_hip.h, host I/O (wave / Kaldi / scp) matches the reference's
vectors, work sharding, CLI surface.  No compute call is made.
"""
import ctypes
import io
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_golden


# ---------------------------------------------------------------------------
# C ABI
# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def built_lib():
    import __graft_entry__ as g
    return g.build(force=False)


def test_library_exports_header_symbols(built_lib):
    from setk_amd import _ffi
    header = open(os.path.join(ROOT, "include", "setk_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(setk_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no prototypes found in include/setk_hip.h"
    assert sorted(_ffi.exported_symbols()) == declared
    lib = ctypes.CDLL(built_lib)
    for name in declared:
        assert hasattr(lib, name), name
    lib.setk_abi_version.restype = ctypes.c_int
    assert lib.setk_abi_version() == 1


def test_rccl_load_failure_is_an_error_code_not_a_crash(built_lib):
    """A host whose librccl cannot be loaded: setk_comm_unique_id / setk_comm_create return
    SETK_ERR_UNSUPPORTED with the loader's message (round-5 advice: the message was built from
    two dlerror() calls, the second of which returns NULL -> a crash instead of the TCP
    fallback of dist.py).  SETK_RCCL_LIB forces the failure; no GPU work is involved."""
    code = ("import ctypes, sys\n"
            f"lib = ctypes.CDLL({built_lib!r})\n"
            "lib.setk_comm_last_error.restype = ctypes.c_char_p\n"
            "buf = ctypes.create_string_buffer(128)\n"
            "rc = lib.setk_comm_unique_id(buf)\n"
            "msg = lib.setk_comm_last_error().decode()\n"
            "comm = ctypes.c_void_p()\n"
            "rc2 = lib.setk_comm_create(ctypes.byref(comm), 0, buf.raw, 0, 1)\n"
            "print(rc, rc2, msg)\n")
    env = dict(os.environ, SETK_RCCL_LIB="/nonexistent/librccl.so.1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr[-1500:])
    rc, rc2, msg = r.stdout.strip().split(" ", 2)
    assert int(rc) == -2 and int(rc2) == -2 and "librccl not found" in msg and "/nonexistent" in msg


def test_star_rendezvous_refuses_strays_and_duplicates():
    """Rank 0 of the TCP star accepts exactly the job's ranks 1..W-1 once each: a connection
    that does not speak the handshake, a rank id out of range and a duplicate id are closed and
    the job still forms; established sockets carry no timeout."""
    import socket
    import struct
    import threading
    from setk_amd.dist import _Star, _MAGIC
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    res = {}

    def rank0():
        res[0] = _Star(0, 3, "127.0.0.1", port, timeout=60)

    th = threading.Thread(target=rank0)
    th.start()

    def raw(payload):
        for _ in range(200):
            try:
                c = socket.create_connection(("127.0.0.1", port), timeout=5)
                break
            except OSError:
                import time
                time.sleep(0.02)
        c.sendall(payload)
        return c
    strays = [raw(b"GET / HTTP/1.0\r\n\r\n"), raw(_MAGIC + struct.pack("<i", 7)), raw(_MAGIC + struct.pack("<i", 0))]
    res[1] = _Star(1, 3, "127.0.0.1", port, timeout=60)
    dup = raw(_MAGIC + struct.pack("<i", 1))
    res[2] = _Star(2, 3, "127.0.0.1", port, timeout=60)
    th.join(timeout=60)
    assert not th.is_alive() and len(res[0].peers) == 2
    assert all(p.gettimeout() is None for p in res[0].peers) and res[1].sock.gettimeout() is None
    out = {}
    ts = [threading.Thread(target=lambda r=r: out.__setitem__(r, res[r].allreduce([r + 1.0, 2.0]))) for r in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert out[0] == out[1] == out[2] == [6.0, 6.0]
    for c in strays + [dup]:
        c.close()
    for r in range(3):
        res[r].close()


def test_no_gpu_means_loud_failure(built_lib):
    """The product path must not fall back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from setk_amd import _ffi
    with pytest.raises(_ffi.SetkError):
        _ffi.Context(0)
    from setk_amd.libs import utils
    with pytest.raises(_ffi.SetkError):
        utils.forward_stft(np.zeros(4000, np.float32), frame_len=512)


def test_product_never_imports_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "setk_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M):
                    bad.append(f)
    assert not bad, bad


# ---------------------------------------------------------------------------
# wave codec
# ---------------------------------------------------------------------------
def test_wavio_roundtrip_and_extensible(tmp_path):
    from setk_amd.libs import wavio, utils
    import scipy.io.wavfile
    doc = load_golden("doc_adaptive_beamformer.npz")
    pcm = doc["egs"][:4000]  # N x 5 int16
    p = tmp_path / "a.wav"
    wavio.write_pcm16(str(p), pcm, 16000)
    sr, back = scipy.io.wavfile.read(str(p))
    assert sr == 16000 and np.array_equal(back, pcm)
    x = utils.read_wav(str(p))
    assert x.shape == (5, 4000) and x.dtype == np.float32
    assert np.array_equal(x, (pcm.astype(np.float32) / 32768.0).T)
    # chunked read + sample-rate check
    assert np.array_equal(utils.read_wav(str(p), beg=100, end=300), x[:, 100:300])
    with pytest.raises(RuntimeError):
        utils.read_wav(str(p), sr=8000)
    # WAVE_FORMAT_EXTENSIBLE header (what the doc assets use)
    import struct
    data = pcm.astype("<i2").tobytes()
    fmt = struct.pack("<HHIIHH", 0xFFFE, 5, 16000, 16000 * 10, 10, 16) + struct.pack(
        "<HHI", 22, 16, 0) + struct.pack("<H", 1) + b"\x00" * 14
    blob = b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(data)) + b"WAVE" + \
        b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(data)) + data
    y, sr = wavio.read(io.BytesIO(blob))
    assert sr == 16000 and np.array_equal(y, pcm.astype(np.float32) / 32768.0)
    # float -> PCM16 follows libsndfile (x * 32767, round to nearest)
    f = np.array([0.0, 0.5, -0.5, 1.0, -1.0, 3.0517578125e-05], np.float32)
    assert wavio.float_to_pcm16(f).tolist() == [0, 16384, -16384, 32767, -32767, 1]
    utils.write_wav(str(tmp_path / "sub" / "b.wav"), f)
    sr, back = scipy.io.wavfile.read(str(tmp_path / "sub" / "b.wav"))
    assert back.tolist() == [0, 16384, -16384, 32767, -32767, 1]


def test_wav_writes_are_atomic_and_skip_existing_checks_completeness(tmp_path):
    """--skip-existing trusts a {key}.wav only if its size is what its headers say; the writer
    works under a temporary name and renames, so an interrupted run leaves no partial file
    under the final name."""
    from setk_amd.libs import wavio
    from setk_amd.sptk.apply_adaptive_beamformer import _complete_wav
    pcm = (np.arange(3000) % 200 - 100).astype(np.int16)
    p = tmp_path / "k.wav"
    wavio.write_pcm16(str(p), pcm, 16000)
    assert sorted(os.listdir(tmp_path)) == ["k.wav"]  # no .part left behind
    assert _complete_wav(str(p))
    whole = p.read_bytes()
    (tmp_path / "cut.wav").write_bytes(whole[: len(whole) // 2])  # what a killed writer used to leave
    assert not _complete_wav(str(tmp_path / "cut.wav"))
    (tmp_path / "hdr.wav").write_bytes(whole[:44])
    assert not _complete_wav(str(tmp_path / "hdr.wav"))
    assert not _complete_wav(str(tmp_path / "missing.wav"))
    # a failing write must not clobber an existing good file
    class Boom(Exception):
        pass
    real = os.writev
    def bad(fd, bufs):
        raise Boom()
    os.writev = bad
    try:
        with pytest.raises(Boom):
            wavio.write_pcm16(str(p), pcm[:10], 16000)
    finally:
        os.writev = real
    assert p.read_bytes() == whole and sorted(os.listdir(tmp_path)) == ["cut.wav", "hdr.wav", "k.wav"]


# ---------------------------------------------------------------------------
# Kaldi I/O against the reference's vectors
# ---------------------------------------------------------------------------
def test_wave_reader_pcm16_frames(tmp_path):
    """read_pcm16: the stored frames of single 16-bit PCM entries (plain file and
    path.ark:offset), None for what the device ingest cannot take."""
    from setk_amd.libs import wavio
    from setk_amd.libs.data_handler import WaveReader
    rng = np.random.default_rng(2)
    pcm = rng.integers(-30000, 30000, size=(500, 4)).astype(np.int16)
    wavio.write_pcm16(str(tmp_path / "a.wav"), pcm, 16000)
    for c in range(2):
        wavio.write_pcm16(str(tmp_path / f"b.CH{c}.wav"), pcm[:, c], 16000)
    wavio.write(str(tmp_path / "f.wav"), (pcm[:, :2] / 32768.0).astype(np.float32), 16000)
    # an "ark" holding the wav at an offset
    blob = open(tmp_path / "a.wav", "rb").read()
    with open(tmp_path / "w.ark", "wb") as fd:
        fd.write(b"key ")
        off = fd.tell()
        fd.write(blob)
    (tmp_path / "wav.scp").write_text(
        f"a {tmp_path}/a.wav\nb {tmp_path}/b.CH*.wav\nf {tmp_path}/f.wav\n"
        f"k {tmp_path}/w.ark:{off}\np cat {tmp_path}/a.wav |\n")
    r = WaveReader(str(tmp_path / "wav.scp"), sr=16000)
    for key in ("a", "k"):
        got = r.read_pcm16(key)
        assert got.dtype == np.int16 and np.array_equal(got, pcm)
        assert np.array_equal(got.T.astype(np.float32) / np.float32(32768), r.read(key))
    assert r.read_pcm16("b") is None      # one file per channel
    assert r.read_pcm16("p") is None      # command pipe
    # float wav: write() stores PCM16 too (libsndfile default) -> frames available
    assert r.read_pcm16("f") is not None
    with pytest.raises(RuntimeError):
        WaveReader(str(tmp_path / "wav.scp"), sr=8000).read_pcm16("a")
    assert WaveReader(str(tmp_path / "wav.scp"), sr=16000, normalize=False).read_pcm16("a") is None


def test_kaldi_goldens():
    from setk_amd.libs import kaldi_io
    from setk_amd.libs.data_handler import ScriptReader, ArchiveReader
    g = load_golden("ref_kaldi.npz")
    cwd = os.getcwd()
    os.chdir(GOLDEN)
    try:
        rd = ScriptReader("kaldi_small.scp")
        assert len(rd) == 2 and "utt_fm" in rd and "nope" not in rd
        m = rd["utt_fm"]
        assert m.dtype == np.float32 and np.array_equal(m, g["utt_fm"])
        assert not m.flags.writeable  # read-only view like the reference
        assert np.array_equal(rd["utt_fv"], g["utt_fv"])
        assert np.array_equal(rd[0], g["utt_fm"])
        seq = dict(ArchiveReader("kaldi_small.ark"))
        assert set(seq) == {"utt_fm", "utt_fv"}
        with open("kaldi_dm.ark", "rb") as fd:
            (key, dm), = list(kaldi_io.read_float_ark(fd))
        assert key == "utt_dm" and dm.dtype == np.float64 and np.array_equal(dm, g["utt_dm"])
    finally:
        os.chdir(cwd)
    head = (float(g["cm.head"][0]), float(g["cm.head"][1]), 9, 6)
    cm = g["cm.pch"].tobytes() + g["cm.body"].tobytes()
    assert np.allclose(kaldi_io.uncompress(cm, "CM", head), g["cm.decoded"], atol=1e-6)
    assert np.allclose(kaldi_io.uncompress(g["cm2.body"].tobytes(), "CM2", head),
                       g["cm2.decoded"], atol=1e-6)
    assert np.allclose(kaldi_io.uncompress(g["cm3.body"].tobytes(), "CM3", head),
                       g["cm3.decoded"], atol=1e-6)


def test_probe_of_compressed_kaldi_matrices(tmp_path):
    """pipeline.probe_kaldi_compressed: the header of a CompressedMatrix entry (what the streaming
    pipeline needs to ship its body as stored) -- kind, global header, body range -- and None for
    everything else; the body it names decodes to what the archive reader returns."""
    import struct
    from setk_amd import pipeline
    from setk_amd.libs import kaldi_io
    from setk_amd.libs.data_handler import ScriptReader
    rng = np.random.default_rng(2)
    td = str(tmp_path)
    want = {}
    with open(f"{td}/m.ark", "wb") as ark, open(f"{td}/m.scp", "w") as scp:
        for key, kind, rows, cols in (("a", "CM2", 7, 5), ("b", "CM3", 3, 9), ("c", "CM", 6, 4)):
            if kind == "CM2":
                body = rng.integers(0, 65536, size=(rows, cols)).astype("<u2").tobytes()
            elif kind == "CM3":
                body = rng.integers(0, 256, size=(rows, cols)).astype(np.uint8).tobytes()
            else:
                body = np.sort(rng.integers(0, 65536, size=(cols, 4)).astype("<u2"), axis=1).tobytes() + \
                    rng.integers(0, 256, size=(cols, rows)).astype(np.uint8).tobytes()
            ark.write(f"{key} ".encode())
            scp.write(f"{key} {td}/m.ark:{ark.tell()}\n")
            ark.write(b"\0B" + kind.encode() + b" " + struct.pack("<ffii", -0.5, 2.0, rows, cols) + body)
            want[key] = (kind, rows, cols, body)
        ark.write(b"d ")
        scp.write(f"d {td}/m.ark:{ark.tell()}\n")
        ark.write(b"\0BFM " + b"\x04" + struct.pack("<i", 2) + b"\x04" + struct.pack("<i", 3) +
                  np.arange(6, dtype="<f4").tobytes())
    files = pipeline.OpenFiles()
    rd = ScriptReader(f"{td}/m.scp")
    try:
        for key, (kind, rows, cols, body) in want.items():
            path, off = rd.locate(key)
            hit = pipeline.probe_kaldi_compressed(files, path, off)
            assert hit[:5] == (kind, -0.5, 2.0, rows, cols) and hit[6] == len(body)
            with open(path, "rb") as f:
                f.seek(hit[5])
                assert f.read(hit[6]) == body
            assert pipeline.probe_kaldi_matrix(files, path, off) is None
            assert np.array_equal(rd[key], kaldi_io.uncompress(body, kind, (-0.5, 2.0, rows, cols)))
        path, off = rd.locate("d")
        assert pipeline.probe_kaldi_compressed(files, path, off) is None
        assert pipeline.probe_kaldi_matrix(files, path, off)[:2] == (2, 3)
    finally:
        files.close()


def test_kaldi_writer_roundtrip(tmp_path):
    from setk_amd.libs.data_handler import ArchiveWriter, ScriptReader
    rng = np.random.default_rng(0)
    mats = {f"k{i}": rng.uniform(size=(5 + i, 257)).astype(np.float32) for i in range(3)}
    ark, scp = str(tmp_path / "m.ark"), str(tmp_path / "m.scp")
    with ArchiveWriter(ark, scp) as w:
        for k, m in mats.items():
            w.write(k, m)
    rd = ScriptReader(scp)
    for k, m in mats.items():
        assert np.array_equal(rd[k], m)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not present")
def test_kaldi_doc_archive_matches_reference_reader():
    """doc/format_transform/asset/egs.ark (too large to commit) through both readers."""
    from setk_amd.libs.data_handler import ArchiveReader
    g = load_golden("ref_kaldi.npz")
    got = list(ArchiveReader("/root/reference/doc/format_transform/asset/egs.ark"))
    assert [k for k, _ in got] == list(g["doc_ark.keys"])
    for (k, m), shape, s, v in zip(got, g["doc_ark.shape"], g["doc_ark.sum"], g["doc_ark.m35"]):
        assert m.shape == tuple(shape)
        assert abs(float(np.sum(m, dtype=np.float64)) - s) < 1e-6 * abs(s)
        assert float(m[3, 5]) == v


# ---------------------------------------------------------------------------
# scp tables / readers
# ---------------------------------------------------------------------------
def test_scp_parsing(tmp_path):
    from setk_amd.libs.data_handler import parse_scps, ScpReader, WaveReader
    from setk_amd.libs import wavio
    p = tmp_path / "a.scp"
    p.write_text("k1 /x/a.wav\nk2 sox /x/b.wav -t wav - remix 1 |\n")
    d = parse_scps(str(p))
    assert list(d) == ["k1", "k2"] and d["k2"] == "sox /x/b.wav -t wav - remix 1 |"
    (tmp_path / "dup.scp").write_text("k1 a\nk1 b\n")
    with pytest.raises(ValueError):
        parse_scps(str(tmp_path / "dup.scp"))
    (tmp_path / "bad.scp").write_text("k1 a b\n")
    with pytest.raises(RuntimeError):
        parse_scps(str(tmp_path / "bad.scp"))
    with pytest.raises(FileNotFoundError):
        ScpReader(str(tmp_path / "missing.scp"))
    # multi-file channels (glob, sorted) and pipes
    rng = np.random.default_rng(1)
    chans = [(rng.uniform(-0.5, 0.5, 1000) * 32767).astype(np.int16) for _ in range(3)]
    for i, c in enumerate(chans):
        wavio.write_pcm16(str(tmp_path / f"u.CH{i + 1}.wav"), c, 16000)
    wavio.write_pcm16(str(tmp_path / "multi.wav"), np.stack(chans, 1), 16000)
    scp = tmp_path / "wav.scp"
    scp.write_text(f"glob {tmp_path}/u.CH*.wav\nmulti {tmp_path}/multi.wav\n"
                   f"pipe cat {tmp_path}/multi.wav |\n")
    rd = WaveReader(str(scp))
    ref = np.stack(chans).astype(np.float32) / 32768.0
    for k in ("glob", "multi", "pipe"):
        assert np.array_equal(rd[k], ref), k
    assert rd.maxabs("multi") == np.max(np.abs(ref))
    assert abs(rd.power("multi") - np.mean(ref[0]**2)) < 1e-6
    assert rd.nsamps("glob") == 1000 and abs(rd.duration("glob") - 1000 / 16000) < 1e-12
    keys = [k for k, _ in rd]
    assert keys == ["glob", "multi", "pipe"]
    with pytest.raises(KeyError):
        rd["nope"]
    with pytest.raises(IndexError):
        rd[1.5]


def test_cli_surface_matches_reference():
    from setk_amd.sptk.apply_adaptive_beamformer import build_parser
    a = build_parser().parse_args(["wav.scp", "mask.scp", "out"])
    expect = dict(wav_scp="wav.scp", tgt_mask="mask.scp", dst_dir="out", itf_mask="", fmt="kaldi",
                  beamformer="mvdr", pmwf_ref=-1, sr=16000, ban=False, rank1_appro="", mask=False,
                  vad_proportion=1, alpha=0.8, chunk_size=-1, channels=4, frame_len=512,
                  frame_hop=256, center=True, round_power_of_two=True, window="hann")
    for k, v in expect.items():
        assert getattr(a, k) == v, k
    b = build_parser().parse_args([
        "--frame-len", "400", "--frame-hop", "160", "--center", "false", "--window", "hamming",
        "--mask-format", "numpy", "--beamformer", "pmwf-1", "--pmwf-ref", "2", "--ban", "true",
        "--rank1-appro", "eig", "--post-masking", "true", "--vad-proportion", "0.9",
        "--online.alpha", "0.7", "--online.chunk-size", "64", "--online.channels", "6",
        "--itf-mask", "itf.scp", "--round-power-of-two", "false", "w", "m", "o"])
    assert (b.frame_len, b.frame_hop, b.center, b.window) == (400, 160, 0, "hamming")
    assert (b.fmt, b.beamformer, b.pmwf_ref, b.ban, b.rank1_appro) == ("numpy", "pmwf-1", 2, 1, "eig")
    assert (b.mask, b.vad_proportion, b.alpha, b.chunk_size, b.channels) == (1, 0.9, 0.7, 64, 6)
    assert b.itf_mask == "itf.scp" and b.round_power_of_two == 0
    with pytest.raises(SystemExit):
        build_parser().parse_args(["--beamformer", "nope", "w", "m", "o"])


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not present")
def test_cli_options_equal_reference_parser():
    """Every option string / default of the reference CLI exists here."""
    src = open("/root/reference/scripts/sptk/apply_adaptive_beamformer.py").read()
    ref_flags = set(re.findall(r'add_argument\("(--[a-z0-9.\-]+)"', src))
    from setk_amd.sptk.apply_adaptive_beamformer import build_parser
    ours = set()
    for act in build_parser()._actions:
        ours.update(o for o in act.option_strings if o.startswith("--"))
    assert ref_flags <= ours, ref_flags - ours


def test_vad_mask_matches_reference_loop():
    from setk_amd.engine import compute_vad_masks
    rng = np.random.default_rng(5)
    S = (rng.standard_normal((33, 40)) + 1j * rng.standard_normal((33, 40))).astype(np.complex64)

    def ref(spectrogram, proportion):  # apply_adaptive_beamformer.py:50-71
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

    for p in (0.6, 0.9, 0.99):
        m1, i1 = compute_vad_masks(S, p)
        m2, i2 = ref(S, p)
        assert i1 == i2 and np.array_equal(m1, m2)


def test_assign_keys_balances_and_partitions():
    from setk_amd.dist import assign_keys
    keys = [f"u{i}" for i in range(23)]
    dur = [((i * 7919) % 31) + 1 for i in range(23)]
    for world in (1, 2, 4, 8):
        parts = [assign_keys(keys, r, world, dur) for r in range(world)]
        flat = sorted(k for p in parts for k in p)
        assert flat == sorted(keys)  # a partition
        loads = [sum(dur[keys.index(k)] for k in p) for p in parts]
        assert max(loads) - min(loads) <= max(dur)
    assert assign_keys(keys, 1, 4) == keys[1::4]


# ---------------------------------------------------------------------------
# world_size 2 (gloo) : the N > 1 path of the CLI sharding
# ---------------------------------------------------------------------------
_WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r})
from setk_amd.dist import Shard
sh = Shard(backend="gloo")
keys = [f"u{{i}}" for i in range(11)]
dur = [((i * 13) % 7) + 1 for i in range(11)]
mine = sh.assign(keys, dur)
sh.barrier()
tot = sh.sum_counts([len(mine), sum(dur[keys.index(k)] for k in mine)])
sh.barrier()
print(json.dumps(dict(rank=sh.rank, world=sh.world, mine=mine, tot=tot)))
sh.close()
"""


def test_shard_world_size_two_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    procs = []
    port = 29600 + (os.getpid() % 300)
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(__import__("json").loads(o.strip().splitlines()[-1]))
    mine = sorted(outs[0]["mine"] + outs[1]["mine"])
    assert mine == sorted(f"u{i}" for i in range(11))
    assert not set(outs[0]["mine"]) & set(outs[1]["mine"])
    for o in outs:
        assert o["world"] == 2
        assert o["tot"][0] == 11 and o["tot"][1] == sum(((i * 13) % 7) + 1 for i in range(11))


_WORKER_RAGGED = r"""
import os, sys, json
sys.path.insert(0, {root!r})
from setk_amd.dist import Shard
from setk_amd.libs.data_handler import WaveReader
sh = Shard(backend="gloo")
reader = WaveReader({scp!r})
mine = sh.assign_by_duration(reader)
sec = sum(reader.peek_nsamps(k) for k in mine) / 16000.0
sh.barrier()
tot = sh.sum_counts([len(mine), sec])
print(json.dumps(dict(rank=sh.rank, mine=mine, sec=sec, tot=tot, w=sh.assigned_weight)))
sh.close()
"""


def test_duration_balanced_sharding_world_size_two_gloo(tmp_path):
    """What the CLIs do under torchrun: keys dealt by the lengths in the wave
    headers (no decode); ragged lengths must balance in SECONDS, not in counts
    (the reference's split_scp.pl balances counts, run_adapt_beamformer.sh:69-70)."""
    from setk_amd.libs import wavio
    lens = [160000, 8000, 8000, 8000, 480000, 16000, 16000, 240000, 32000, 8000, 100000]
    with open(tmp_path / "wav.scp", "w") as f:
        for i, n in enumerate(lens):
            wavio.write_pcm16(str(tmp_path / f"u{i}.wav"), np.zeros((n, 2), np.int16), 16000)
            f.write(f"u{i} {tmp_path}/u{i}.wav\n")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER_RAGGED.format(root=ROOT, scp=str(tmp_path / "wav.scp")))
    port = 29950 + (os.getpid() % 40)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(__import__("json").loads(o.strip().splitlines()[-1]))
    assert sorted(outs[0]["mine"] + outs[1]["mine"]) == sorted(f"u{i}" for i in range(len(lens)))
    total = sum(lens) / 16000.0
    for o in outs:
        assert abs(o["tot"][1] - total) < 1e-6 and o["tot"][0] == len(lens)
        assert abs(o["w"] / 16000.0 - o["sec"]) < 1e-9
    # seconds are balanced to within the longest utterance's share of a greedy deal;
    # a plain round robin on counts would put 480000 + 240000 + ... on one rank
    assert abs(outs[0]["sec"] - outs[1]["sec"]) <= max(lens) / 16000.0 * 0.5
    rr = [sum(lens[r::2]) / 16000.0 for r in range(2)]
    assert abs(outs[0]["sec"] - outs[1]["sec"]) < abs(rr[0] - rr[1])


def test_kaldi_sequential_vectors_across_buffer_boundaries(tmp_path):
    """Vector entries whose type token straddles an 8 KiB BufferedReader boundary
    (a peek-based dispatch saw one byte there and took the vector for a matrix)."""
    from setk_amd.libs import kaldi_io
    from setk_amd.libs.data_handler import ArchiveReader, ArchiveWriter
    rng = np.random.default_rng(0)
    path = str(tmp_path / "v.ark")
    items = []
    with ArchiveWriter(path) as w:
        for i in range(40):
            # payload sizes walk the header of the next entry across offset 8192 * k
            v = rng.standard_normal(2040 + i).astype(np.float32)
            items.append((f"k{i}", v))
            w.write(f"k{i}", v)
            m = rng.standard_normal((3, 5 + i)).astype(np.float32)
            items.append((f"m{i}", m))
            w.write(f"m{i}", m)
    got = list(ArchiveReader(path))
    assert [k for k, _ in got] == [k for k, _ in items]
    for (_, a), (_, b) in zip(got, items):
        assert a.shape == b.shape and np.array_equal(a, b)


def test_writer_family(tmp_path):
    from setk_amd.libs.data_handler import (ArchiveWriter, NumpyWriter, ScriptReader, WaveWriter,
                                            WaveReader)
    m = np.arange(12, dtype=np.float32).reshape(3, 4)
    with ArchiveWriter(str(tmp_path / "a.ark"), str(tmp_path / "a.scp")) as w:
        w.write("x", m)
        w.write("y", m[0])
        with pytest.raises(RuntimeError):
            w.write("z", [1, 2, 3])
    r = ScriptReader(str(tmp_path / "a.scp"))
    assert np.array_equal(r["x"], m) and np.array_equal(r["y"], m[0])
    with NumpyWriter(str(tmp_path / "npy"), str(tmp_path / "n.scp")) as w:
        w.write("x", m)
    assert np.array_equal(np.load(tmp_path / "npy" / "x.npy"), m)
    assert (tmp_path / "n.scp").read_text().split()[0] == "x"
    with WaveWriter(str(tmp_path / "wav" / "deep"), str(tmp_path / "w.scp")) as w:
        w.write("s", np.linspace(-0.5, 0.5, 100).astype(np.float32))
        w.write_pcm16("p", np.arange(-50, 50, dtype=np.int16))
    rd = WaveReader(str(tmp_path / "w.scp"))
    assert rd.peek_nsamps("s") == 100 and rd.peek_nsamps("p") == 100
    assert np.array_equal(rd.read_pcm16("p")[:, 0], np.arange(-50, 50, dtype=np.int16))


def test_payload_reads_by_name_and_holds_no_descriptor(tmp_path):
    """pipeline.Payload names its file: load_into opens, reads the byte range and closes (a run of
    n utterances must not hold 2 n descriptors: the descriptor table's growth stalls a
    multi-threaded process)."""
    from setk_amd.pipeline import OpenFiles, Payload, probe_npy
    blob = np.arange(5000, dtype=np.uint8).tobytes()
    paths = []
    for i in range(40):
        p = tmp_path / f"p{i}.bin"
        p.write_bytes(blob)
        paths.append(str(p))
    before = len(os.listdir("/proc/self/fd"))
    for p in paths:
        dst = np.zeros(1000, dtype=np.uint8)
        Payload(path=p, offset=123, nbytes=1000).load_into(dst)
        assert dst.tobytes() == blob[123:1123]
    assert len(os.listdir("/proc/self/fd")) == before
    with pytest.raises(IOError):
        Payload(path=paths[0], offset=4500, nbytes=1000).load_into(np.zeros(1000, dtype=np.uint8))
    assert len(os.listdir("/proc/self/fd")) == before
    # the header-probe cache stays small whatever the number of files
    files = OpenFiles()
    for i in range(100):
        q = tmp_path / f"m{i}.npy"
        np.save(q, np.zeros((3, 257), dtype=np.float32))
        shape, dt, fortran, off = probe_npy(files, str(q))
        assert shape == (3, 257) and not fortran and dt == np.dtype("<f4")
    assert len(files.fds) <= files.limit
    files.close()
    assert len(os.listdir("/proc/self/fd")) == before


def test_native_reader_pool_reads_what_the_interpreter_reads(built_lib, tmp_path):
    """setk_host_read_payloads (csrc/hostio.hip): the read stage of a batch in one call -- byte
    ranges of files into caller memory through the mapping (large payloads) and pread (small
    ones), per-payload errno, concurrent calls sharing the pool, no descriptor left open."""
    import errno
    import threading
    lib = ctypes.CDLL(built_lib)
    fn = lib.setk_host_read_payloads
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_longlong),
                   ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
                   ctypes.c_longlong, ctypes.POINTER(ctypes.c_int)]
    rng = np.random.default_rng(5)
    blobs, paths = [], []
    for i in range(24):
        b = rng.integers(0, 256, size=70_000 + 4099 * i, dtype=np.uint8)
        q = tmp_path / f"f{i}.bin"
        q.write_bytes(b.tobytes())
        blobs.append(b)
        paths.append(str(q))

    def call(idx, offs, sizes, threads, mmap_min):
        n = len(idx)
        dst = [np.full(max(sz, 1), 0xAB, dtype=np.uint8) for sz in sizes]
        P = (ctypes.c_char_p * n)(*[os.fsencode(paths[i]) if i >= 0 else b"/nonexistent/x.bin" for i in idx])
        O = (ctypes.c_longlong * n)(*offs)
        B = (ctypes.c_longlong * n)(*sizes)
        D = (ctypes.c_void_p * n)(*[d.ctypes.data for d in dst])
        S = (ctypes.c_int * n)(*([77] * n))
        assert fn(n, P, O, B, D, threads, mmap_min, S) == 0
        return dst, list(S)

    before = len(os.listdir("/proc/self/fd"))
    for mmap_min in (1, 1 << 40):  # everything through the mapping / everything through pread
        idx = list(range(24))
        offs = [37 * i + (4096 if i % 3 == 0 else 0) for i in idx]   # page-aligned and odd offsets
        sizes = [len(blobs[i]) - offs[i] - (i % 5) for i in idx]
        dst, st = call(idx, offs, sizes, 5, mmap_min)
        assert st == [0] * 24
        for i in idx:
            assert np.array_equal(dst[i], blobs[i][offs[i]:offs[i] + sizes[i]]), (mmap_min, i)
        # a missing file, a payload that runs past the end of its file, an empty payload
        dst, st = call([-1, 0, 1, 2], [0, 60_000, 0, 10], [100, 20_000, 0, 50], 3, mmap_min)
        assert st[0] == errno.ENOENT and st[1] == errno.EIO and st[2] == 0 and st[3] == 0
        assert np.array_equal(dst[3], blobs[2][10:60])
    # concurrent calls share the pool
    results = {}

    def worker(k):
        idx = [(k + 3 * r) % 24 for r in range(16)]
        dst, st = call(idx, [k] * 16, [50_000] * 16, 4, 1)
        results[k] = st == [0] * 16 and all(np.array_equal(d, blobs[i][k:k + 50_000]) for d, i in zip(dst, idx))
    th = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert results == {k: True for k in range(6)}
    assert len(os.listdir("/proc/self/fd")) == before
    # argument checks: nothing is read
    assert fn(-1, None, None, None, None, 1, 0, None) != 0
    assert fn(0, None, None, None, None, 1, 0, None) == 0


# ---------------------------------------------------------------------------
# the solve of the bin-resident CGMM, as a numpy model (tests/jacobi_model.py)
# ---------------------------------------------------------------------------
def test_parallel_jacobi_model_against_lapack():
    """The algorithm cgmm_bin_em_kernel runs between two passes -- parallel-order two-sided
    Jacobi with float32 angles, warm starts, the 'last sweep' rule, the eigenvalue floor --
    against LAPACK: the WHITENED error of R_eff^-1 (the relative error of any quadratic form
    x^H R_eff^-1 x) stays below 1e-6 on rank-deficient, ill-scaled, point-source and
    near-degenerate covariances, cold and warm started; and the fast path's certificate never
    accepts a matrix whose floor is active."""
    import jacobi_model as jm
    rng = np.random.default_rng(1)
    worst, cold, warm = 0.0, [], []
    for n in (6, 8, 3):
        for trial in range(120):
            kind = trial % 4
            T = int(rng.integers(1, 40)) if kind == 0 else 400
            X = rng.standard_normal((n, T)) + 1j * rng.standard_normal((n, T))
            if kind == 1:
                X = X * (10.0 ** rng.uniform(-4, 0, size=n))[:, None]
            if kind == 2:
                d = np.exp(1j * rng.uniform(0, 6.28, n))
                X = d[:, None] * (rng.standard_normal(T) + 1j * rng.standard_normal(T)) + \
                    10.0 ** rng.uniform(-4, -1) * X
            if kind == 3:
                X = X * (1 + 1e-6 * rng.standard_normal((n, 1)))
            R = X @ X.conj().T / T * 10.0 ** rng.uniform(-10, 2)
            P, ld, V, s = jm.effective_inverse(R)
            Pr, ldr, S = jm.lapack_reference(R)
            worst = max(worst, np.abs(S.conj().T @ (P - Pr) @ S).max(), abs(ld - ldr) / 100)
            cold.append(s)
            g = rng.uniform(0.9, 1.1, T)           # an EM-like reweighting of the frames
            R2 = (X * g) @ X.conj().T / T
            P2, ld2, _, s2 = jm.effective_inverse(R2, V0=V)
            Pr2, ldr2, S2 = jm.lapack_reference(R2)
            worst = max(worst, np.abs(S2.conj().T @ (P2 - Pr2) @ S2).max())
            warm.append(s2)
            if jm.certificate(R):
                w = np.linalg.eigvalsh((R + R.conj().T) / 2)
                assert w.min() >= jm.EPS * w.max(), "certificate accepted a floored matrix"
    assert worst < 1e-6, worst
    assert max(cold) <= 8 and np.mean(warm) < np.mean(cold)


def test_permu_aligner_realigns_flipped_bins():
    """libs/cluster.permu_aligner (--solve-permu): bins whose two classes were swapped come
    back aligned with their neighbours; identical to the unmodified reference where its
    tree is available (build container)."""
    from setk_amd.libs.cluster import permu_aligner
    doc = load_golden("doc_adaptive_beamformer.npz")
    m = doc["cgmm_mask"].astype(np.float64)          # T x F, the reference's own CGMM mask
    clean = np.stack([m, 1 - m])
    rng = np.random.default_rng(0)
    flip = rng.random(257) < 0.3
    masks = clean.copy()
    masks[:, :, flip] = masks[::-1][:, :, flip]
    out = permu_aligner(masks.copy())
    assert out.shape == masks.shape
    # every bin comes back in the clean orientation or (globally consistent) its mirror
    same = np.all(out == clean, axis=(0, 1))
    mirrored = np.all(out == clean[::-1], axis=(0, 1))
    assert np.all(same | mirrored) and (same.mean() > 0.95 or mirrored.mean() > 0.95)
    with pytest.raises(ValueError):
        permu_aligner(np.zeros((2, 10, 129)))
    with pytest.raises(RuntimeError):
        permu_aligner(np.zeros((10, 257)))
    from oracle import ref_harness as rh
    if rh.available():
        ref = rh.load().cluster.permu_aligner(masks.copy())
        assert np.array_equal(out, ref)
        m3 = rng.random((3, 40, 257))
        m3 /= m3.sum(0, keepdims=True)
        assert np.array_equal(permu_aligner(m3.copy()), rh.load().cluster.permu_aligner(m3.copy()))


def test_cosine_windows_equal_scipy_bit_for_bit():
    """libs.utils builds hann / hamming / blackman itself (no scipy.signal import on the CLI's
    start-up path): the same operations as scipy.signal.windows.general_cosine, same bits."""
    import scipy.signal
    from setk_amd.libs.utils import stft_window
    for name in ("hann", "hamming", "blackman"):
        for L in (2, 3, 25, 400, 512, 1024):
            ref = scipy.signal.get_window(name, L, fftbins=True).astype(np.float32)
            assert np.array_equal(stft_window(name, L), ref), (name, L)
    assert np.array_equal(stft_window("sqrthann", 512),
                          (scipy.signal.windows.hann(512, sym=False)**0.5).astype(np.float32))
    assert np.array_equal(stft_window("bartlett", 400),
                          scipy.signal.get_window("bartlett", 400, fftbins=True).astype(np.float32))


def test_cgmm_seeded_start_is_independent_of_the_sharding(tmp_path):
    """--num-classes 3 under several ranks: every rank walks the table and discards the draws of
    the utterances that are not its own, so each utterance starts as in the reference's
    one-process run.  Checked on the host logic alone: the frame count taken from the wave
    header equals the oracle's STFT frame count, and discarding K x F x T uniforms leaves the
    legacy generator where drawing the K x F x T array would have."""
    import argparse
    import scipy.io.wavfile
    from oracle import np_oracle as o
    from setk_amd.libs.data_handler import WaveReader
    from setk_amd.sptk.estimate_cgmm_masks import _num_frames
    td = str(tmp_path)
    lens = [16000, 5000, 12345, 700, 513]
    with open(f"{td}/wav.scp", "w") as f:
        for i, n in enumerate(lens):
            scipy.io.wavfile.write(f"{td}/u{i}.wav", 16000, np.zeros((n, 2), np.int16))
            f.write(f"u{i} {td}/u{i}.wav\n")
    reader = WaveReader(f"{td}/wav.scp")
    for center in (True, False):
        for frame_len, hop in ((512, 256), (400, 160), (512, 128)):
            args = argparse.Namespace(center=center, frame_hop=hop, frame_len=frame_len)
            n_fft = 512
            for i, n in enumerate(lens):
                if not center and n < n_fft:
                    continue
                want = o.forward_stft(np.zeros(n, np.float32), frame_len=frame_len, frame_hop=hop, center=center,
                                      window="hann", round_power_of_two=True, transpose=False).shape[1]
                assert _num_frames(reader, f"u{i}", args, n_fft) == want, (center, frame_len, hop, n)
    K, F = 3, 257
    frames = [63, 20, 49]
    np.random.seed(777)
    seq = [np.random.uniform(size=[K, F, t]) for t in frames]           # the one-process run
    for owner in range(3):                                              # a rank that owns one utterance
        np.random.seed(777)
        for j, t in enumerate(frames):
            if j == owner:
                assert np.array_equal(np.random.uniform(size=[K, F, t]), seq[j])
            else:
                np.random.uniform(size=K * F * t)


_WORKER_TORCH_FREE = r"""
import os, sys, json
sys.path.insert(0, {root!r})
from setk_amd import _ffi
from setk_amd.dist import Shard
_ffi.set_torch_free()                           # what the command lines do before the first Context
sh = Shard()                                    # default backend: rccl, else the TCP star
keys = [f"u{{i}}" for i in range(37)]
dur = [((i * 13) % 7) + 1 for i in range(37)]
mine = sh.assign(keys, dur)
sh.barrier()
tot = sh.sum_counts([len(mine), sum(dur[keys.index(k)] for k in mine)])
mx = sh.max_values([float(sh.rank), 3.5])
sh.barrier()
print(json.dumps(dict(rank=sh.rank, world=sh.world, mine=mine, tot=tot, mx=mx, backend=sh.backend,
                      torch="torch" in sys.modules)))
sh.close()
"""


@pytest.mark.parametrize("world", [2, 8])
def test_shard_without_torch(tmp_path, world):
    """The multi-rank control flow without `import torch` in any rank (round-5 review, item 6):
    on a GPU node the barrier and the counters go over RCCL through the library's own
    setk_comm_* entry points (rendezvous of the ncclUniqueId over a TCP socket on MASTER_PORT +
    1); here, without a GPU per rank, every rank agrees on the TCP star instead.  Same deal of
    the keys, same sums as the torch.distributed backends above."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER_TORCH_FREE.format(root=ROOT))
    port = 30400 + (os.getpid() % 500) + 10 * world
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.pop("SETK_DIST_BACKEND", None)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o_, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-2000:]
        outs.append(__import__("json").loads(o_.strip().splitlines()[-1]))
    allk = sorted(k for o_ in outs for k in o_["mine"])
    assert allk == sorted(f"u{i}" for i in range(37))
    total = sum(((i * 13) % 7) + 1 for i in range(37))
    for o_ in outs:
        assert o_["world"] == world and not o_["torch"] and o_["backend"] in ("rccl", "tcp")
        assert o_["tot"] == [37.0, float(total)] and o_["mx"] == [float(world - 1), 3.5]
    assert len({o_["backend"] for o_ in outs}) == 1
