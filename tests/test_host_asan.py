"""The HOST side of libsetk_hip.so under AddressSanitizer + UBSan, no GPU: the library is
rebuilt host-only (hipcc --cuda-host-only) against tools/hoststub/hip_stub.cpp, a host-memory
stand-in for the 34 HIP entry points it imports whose kernel launches only validate their
geometry and whose copies bound-check the device side; tests/host_asan_driver.py then calls
every entry point of include/setk_hip.h at the shapes the GPU tests use (1-16 channels, ragged
batches, 30 s utterances, the streaming and bin-resident CGMM paths, WPE)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang"


def _runtime():
    if not os.path.exists(CLANG):
        pytest.skip("no clang to build the sanitizer library with")
    rt = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True,
                        text=True).stdout.strip()
    if not os.path.isfile(rt):
        pytest.skip("no shared ASAN runtime in this toolchain")
    return rt


@pytest.fixture(scope="module")
def asan_env():
    rt = _runtime()
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "hoststub", "build.sh")], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    env = dict(os.environ)
    # (a 64 KB table-staging buffer: the library's page-locked ring wraps many times in a run)
    env.update(LD_PRELOAD=rt, SETK_LIB=os.path.join(ROOT, "_abl", "libsetk_hostasan.so"),
               SETK_PIN_CAP_KB="64", SETK_ALLOW_HOSTSTUB="1",
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:exitcode=99",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    return env


def _drive(env, *args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_asan_driver.py"), *args],
                          capture_output=True, text=True, env=env, timeout=900)


def test_stand_in_is_refused_unless_asked_for(asan_env):
    """SETK_LIB pointing at the stand-in build must not turn into a silent do-nothing run."""
    env = dict(asan_env)
    env.pop("SETK_ALLOW_HOSTSTUB")
    r = subprocess.run([sys.executable, "-c", "from setk_amd import _ffi; _ffi.Context(0)"], capture_output=True,
                       text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode != 0 and "stand-in" in r.stderr, r.stderr[-2000:]


def test_sanitizer_is_live(asan_env):
    r = _drive(asan_env, "--selftest-overflow")
    assert r.returncode == 99 and "heap-buffer-overflow" in r.stderr, r.stderr[-2000:]
    r = _drive(asan_env, "--selftest-device-deref")
    assert r.returncode == 99 and "use-after-poison" in r.stderr, r.stderr[-2000:]


def test_every_entry_point_is_clean_under_asan_ubsan(asan_env):
    r = _drive(asan_env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "runtime error" not in r.stderr, r.stderr[-4000:]          # UBSan
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["violations"] == 0, rec
    assert rec["launches"] > 500 and rec["copies"] > 1000
    assert rec["live_allocations"] == 0, "device memory still held after setk_destroy"


def test_streaming_cli_host_side_under_asan(asan_env, tmp_path):
    """The whole host pipeline of apply_adaptive_beamformer.py (table readers, reader threads,
    page-locked slabs, copy / compute / copy-out streams, writer threads) against the stand-in:
    it needs no torch and no GPU, so it runs here under ASAN + UBSan.  The kernels do nothing,
    the waveforms are silence of the right length."""
    import numpy as np
    import scipy.io.wavfile
    from setk_amd.libs import wavio
    rng = np.random.default_rng(0)
    td = str(tmp_path)
    lens = [16000, 23456, 8000, 30011, 16000, 12345, 4000]
    with open(f"{td}/wav.scp", "w") as ws, open(f"{td}/mask.scp", "w") as ms:
        for i, n in enumerate(lens):
            x = (rng.standard_normal((n, 4)) * 1000).astype(np.int16)
            wavio.write_pcm16(f"{td}/u{i}.wav", x, 16000)
            np.save(f"{td}/m{i}.npy", rng.random((1 + n // 256, 257)).astype(np.float32))
            ws.write(f"u{i} {td}/u{i}.wav\n")
            ms.write(f"u{i} {td}/m{i}.npy\n")
    # (the library's reader pool, csrc/hostio.hip: the wave payloads through its mapping path, the
    # shorter masks through pread)
    env = dict(asan_env, SETK_MMAP_MIN_KB="48")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "sptk", "apply_adaptive_beamformer.py"),
                        "--mask-format", "numpy", "--batch-utts", "3", "--profile", f"{td}/prof.json",
                        f"{td}/wav.scp", f"{td}/mask.scp", f"{td}/out"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert "Processed 7 utterances out of 7" in r.stderr
    import json
    with open(f"{td}/prof.json") as f:
        assert json.load(f)["stages"]["read_mode"] == "native"
    for i, n in enumerate(lens):
        sr, y = scipy.io.wavfile.read(f"{td}/out/u{i}.wav")
        assert sr == 16000 and y.dtype == np.int16 and y.shape == (256 * (n // 256),)


def test_streaming_cli_with_compressed_kaldi_masks_under_asan(asan_env, tmp_path):
    """The same pipeline fed from a Kaldi archive of CompressedMatrix masks (CM / CM2 / CM3, one stored
    F x T): the bodies travel as stored and setk_kaldi_cm_decode_batch expands them (round 6).
    Host side only here -- the table of the decode call, the slab layout with its device-only
    float32 homes -- under ASAN + UBSan; the arithmetic is the GPU test's."""
    import json
    import struct
    import numpy as np
    import scipy.io.wavfile
    from setk_amd.libs import wavio
    rng = np.random.default_rng(1)
    td = str(tmp_path)
    lens, kinds = [16000, 9000, 23456, 12000, 8000], ["CM2", "CM3", "CM2", "CM", "CM3"]
    with open(f"{td}/wav.scp", "w") as ws, open(f"{td}/masks.ark", "wb") as ark, open(f"{td}/mask.scp", "w") as ms:
        for i, (n, kind) in enumerate(zip(lens, kinds)):
            x = (rng.standard_normal((n, 4)) * 1000).astype(np.int16)
            wavio.write_pcm16(f"{td}/u{i}.wav", x, 16000)
            T = 1 + n // 256
            rows, cols = (257, T) if i == 2 else (T, 257)
            if kind == "CM2":
                body = rng.integers(0, 65536, size=(rows, cols)).astype("<u2").tobytes()
            elif kind == "CM3":
                body = rng.integers(0, 256, size=(rows, cols)).astype(np.uint8).tobytes()
            else:
                body = np.sort(rng.integers(0, 65536, size=(cols, 4)).astype("<u2"), axis=1).tobytes() + \
                    rng.integers(0, 256, size=(cols, rows)).astype(np.uint8).tobytes()
            ws.write(f"u{i} {td}/u{i}.wav\n")
            ark.write(f"u{i} ".encode())
            ms.write(f"u{i} {td}/masks.ark:{ark.tell()}\n")
            ark.write(b"\0B" + kind.encode() + b" " + struct.pack("<ffii", 0.0, 1.0, rows, cols) + body)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "sptk", "apply_adaptive_beamformer.py"),
                        "--mask-format", "kaldi", "--batch-utts", "2", "--profile", f"{td}/prof.json",
                        f"{td}/wav.scp", f"{td}/mask.scp", f"{td}/out"],
                       capture_output=True, text=True, env=asan_env, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert "Processed 5 utterances out of 5" in r.stderr
    with open(f"{td}/prof.json") as f:
        st = json.load(f)["stages"]
    # the archive's bytes went up, not their float32 expansion (4 B per cell)
    cells = sum((1 + n // 256) * 257 for n in lens)
    assert st["bytes_in"] < sum(2 * 4 * n for n in lens) + 3 * cells
    for i, n in enumerate(lens):
        sr, y = scipy.io.wavfile.read(f"{td}/out/u{i}.wav")
        assert sr == 16000 and y.dtype == np.int16 and y.shape == (256 * (n // 256),)


def test_cgmm_cli_host_side_under_asan(asan_env, tmp_path):
    """estimate_cgmm_masks.py (batched path: CgmmEstimator.estimate on the library's own slabs
    and stream, no torch) against the stand-in, under ASAN + UBSan."""
    import numpy as np
    from setk_amd.libs import wavio
    rng = np.random.default_rng(1)
    td = str(tmp_path)
    lens = [16000, 20011, 9000]
    with open(f"{td}/wav.scp", "w") as ws:
        for i, n in enumerate(lens):
            wavio.write_pcm16(f"{td}/u{i}.wav", (rng.standard_normal((n, 3)) * 900).astype(np.int16), 16000)
            ws.write(f"u{i} {td}/u{i}.wav\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "sptk", "estimate_cgmm_masks.py"),
                        "--num-iters", "3", "--batch-utts", "2", f"{td}/wav.scp", f"{td}/masks"],
                       capture_output=True, text=True, env=asan_env, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert "Train 3 utterances over 3" in r.stderr
    for i, n in enumerate(lens):
        assert np.load(f"{td}/masks/u{i}.npy").shape == (1 + n // 256, 257)


def test_fixed_beamformer_cli_host_side_under_asan(asan_env, tmp_path):
    """apply_fixed_beamformer.py (FixedBatchBeamformer on the library's own slabs and stream, no
    torch) against the stand-in, under ASAN + UBSan: two beams, ragged lengths."""
    import numpy as np
    import scipy.io.wavfile
    from setk_amd.libs import wavio
    rng = np.random.default_rng(2)
    td = str(tmp_path)
    lens = [16000, 12001, 30000, 8000, 5000]
    w = (rng.standard_normal((2, 257, 4)) + 1j * rng.standard_normal((2, 257, 4))).astype(np.complex64)
    np.save(f"{td}/w.npy", w)
    with open(f"{td}/wav.scp", "w") as ws, open(f"{td}/beam.scp", "w") as bs:
        for i, n in enumerate(lens):
            wavio.write_pcm16(f"{td}/u{i}.wav", (rng.standard_normal((n, 4)) * 900).astype(np.int16), 16000)
            ws.write(f"u{i} {td}/u{i}.wav\n")
            bs.write(f"u{i} {i % 2}\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "sptk", "apply_fixed_beamformer.py"),
                        "--beam", f"{td}/beam.scp", "--batch-utts", "2", f"{td}/wav.scp", f"{td}/w.npy",
                        f"{td}/out"], capture_output=True, text=True, env=asan_env, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert "Processed 5 utterances" in r.stderr
    for i, n in enumerate(lens):
        sr, y = scipy.io.wavfile.read(f"{td}/out/u{i}.wav")
        assert sr == 16000 and y.dtype == np.int16 and y.shape == (256 * (n // 256),)


def test_wpe_cli_host_side_under_asan(asan_env, tmp_path):
    """apply_wpe.py (BatchDereverb on the library's own slabs, scratch and stream, no torch)
    against the stand-in, under ASAN + UBSan: multi-channel PCM16 frames out."""
    import numpy as np
    import scipy.io.wavfile
    from setk_amd.libs import wavio
    rng = np.random.default_rng(3)
    td = str(tmp_path)
    lens = [16000, 9001, 12000]
    with open(f"{td}/wav.scp", "w") as ws:
        for i, n in enumerate(lens):
            wavio.write_pcm16(f"{td}/u{i}.wav", (rng.standard_normal((n, 2)) * 900).astype(np.int16), 16000)
            ws.write(f"u{i} {td}/u{i}.wav\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "sptk", "apply_wpe.py"),
                        "--frame-len", "512", "--frame-hop", "128", "--taps", "5", "--batch-utts", "2",
                        f"{td}/wav.scp", f"{td}/out"], capture_output=True, text=True, env=asan_env, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert "Processed 3 utterances over 3" in r.stderr
    for i, n in enumerate(lens):
        sr, y = scipy.io.wavfile.read(f"{td}/out/u{i}.wav")
        assert sr == 16000 and y.dtype == np.int16 and y.shape == (128 * (n // 128), 2)


def _make_table(td, lens, channels=4, seed=0):
    import numpy as np
    from setk_amd.libs import wavio
    rng = np.random.default_rng(seed)
    with open(f"{td}/wav.scp", "w") as ws, open(f"{td}/mask.scp", "w") as ms:
        for i, n in enumerate(lens):
            wavio.write_pcm16(f"{td}/u{i}.wav", (rng.standard_normal((n, channels)) * 1000).astype(np.int16), 16000)
            np.save(f"{td}/m{i}.npy", rng.random((1 + n // 256, 257)).astype(np.float32))
            ws.write(f"u{i} {td}/u{i}.wav\n")
            ms.write(f"u{i} {td}/m{i}.npy\n")


def _cli(env, td, *extra, timeout=300):
    return subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "sptk", "apply_adaptive_beamformer.py"),
                           "--mask-format", "numpy", *extra, f"{td}/wav.scp", f"{td}/mask.scp", f"{td}/out"],
                          capture_output=True, text=True, env=env, timeout=timeout)


def test_streaming_cli_failure_paths_do_not_hang(asan_env, tmp_path):
    """What the pipeline does when things go wrong, on the stand-in (a hang is the timeout):
    a writer that cannot create its file ends the run with that error after the other batches
    were drained; a key without a mask is skipped; a mask of the wrong length is the reference's
    ValueError; an empty table is a clean run."""
    import numpy as np
    td = str(tmp_path)
    lens = [8000 + 997 * i for i in range(9)]
    _make_table(td, lens)
    os.makedirs(f"{td}/out/u4.wav")                       # the writer's open() of u4.wav fails
    r = _cli(asan_env, td, "--batch-utts", "2", "--pipeline-depth", "2")
    assert r.returncode != 0 and "IsADirectoryError" in r.stderr, r.stderr[-3000:]
    assert "AddressSanitizer" not in r.stderr
    written = sorted(f for f in os.listdir(f"{td}/out") if os.path.isfile(f"{td}/out/{f}"))
    assert "u0.wav" in written and "u1.wav" in written    # batches before the failure came out
    # a key without a mask is skipped, the rest is processed
    td2 = str(tmp_path / "b")
    os.makedirs(td2)
    _make_table(td2, lens[:5])
    lines = open(f"{td2}/mask.scp").read().splitlines()
    open(f"{td2}/mask.scp", "w").write("\n".join(lines[:2] + lines[3:]) + "\n")
    r = _cli(asan_env, td2, "--batch-utts", "2")
    assert r.returncode == 0 and "Processed 4 utterances out of 5" in r.stderr, r.stderr[-3000:]
    assert not os.path.exists(f"{td2}/out/u2.wav")
    # a mask with too few rows: the reference's shape ValueError, raised by the planner
    td3 = str(tmp_path / "c")
    os.makedirs(td3)
    _make_table(td3, lens[:3])
    np.save(f"{td3}/m1.npy", np.ones((7, 257), np.float32))
    r = _cli(asan_env, td3)
    assert r.returncode != 0 and "do not match with mask" in r.stderr, r.stderr[-3000:]
    # an empty table
    td4 = str(tmp_path / "d")
    os.makedirs(td4)
    open(f"{td4}/wav.scp", "w").close()
    open(f"{td4}/mask.scp", "w").close()
    r = _cli(asan_env, td4)
    assert (r.returncode == 0 and "Processed 0 utterances" in r.stderr) or "empty" in r.stderr.lower(), r.stderr[-2000:]


def test_two_rank_cli_on_the_stand_in(tmp_path):
    """The command line as torchrun starts it, two ranks, on CPU: gloo for the barrier and the
    counters, the HIP stand-in (two pretended devices, no sanitizer: torch is in these
    processes) for everything else.  Every utterance is written exactly once, by the rank the
    duration-balanced deal gave it to, and rank 0 reports the total."""
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "hoststub", "build.sh")], capture_output=True,
                       text=True, timeout=900, env=dict(os.environ, PLAIN="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    td = str(tmp_path)
    lens = [16000, 64000, 8000, 30011, 16000, 12345, 48000, 9000]
    _make_table(td, lens)
    port = 29900 + (os.getpid() % 90)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HOSTSTUB_DEVICES="2", SETK_ALLOW_HOSTSTUB="1",
                   SETK_LIB=os.path.join(ROOT, "_abl", "libsetk_hoststub.so"))
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "scripts", "sptk", "apply_adaptive_beamformer.py"),
             "--mask-format", "numpy", "--batch-utts", "2", "--profile", f"{td}/prof.json",
             f"{td}/wav.scp", f"{td}/mask.scp", f"{td}/out"],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    errs = []
    for p in procs:
        _, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-3000:]
        errs.append(e)
    assert sorted(os.listdir(f"{td}/out")) == sorted(f"u{i}.wav" for i in range(len(lens)))
    assert any("Processed 8 utterances out of 8" in e for e in errs), errs[0][-1500:]
    import json
    per_rank = [json.load(open(f"{td}/prof.json.rank{k}")) for k in range(2)]
    assert sum(p["utts"] for p in per_rank) == 8 and all(p["utts"] >= 2 for p in per_rank)
    assert all(p["mode"] == "pipeline" and p["world"] == 2 for p in per_rank)
    # no rank imported torch: barrier and counters over the library's own exchange (RCCL on a GPU
    # node, the TCP star here where no rank has a GPU)
    assert all(p["dist_backend"] in ("rccl", "tcp") and not p["torch_loaded"] for p in per_rank)


def test_eight_rank_cli_on_the_stand_in_balances_ragged_lengths(tmp_path):
    """World size 8 as the driver's 8-GPU launch would run the CLI (BASELINE configs[2]: a batch
    sharded over 8 GPUs), on CPU: gloo for the barrier and the counters, the HIP stand-in with
    eight pretended devices for the rest.  128 utterances of ragged length (0.4 - 4 s): the
    longest-first deal on the header durations leaves every rank within 2 % of the mean
    number of samples, every utterance is written exactly once, and rank 0 reports the total
    (reference mode: scripts/run_adapt_beamformer.sh:69-92, run.pl JOB=1:nj over split scps)."""
    import json
    import numpy as np
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "hoststub", "build.sh")], capture_output=True,
                       text=True, timeout=900, env=dict(os.environ, PLAIN="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    td = str(tmp_path)
    rng = np.random.default_rng(8)
    lens = [int(n) for n in rng.integers(6400, 64000, size=128)]
    _make_table(td, lens, channels=2)
    world = 8
    port = 29700 + (os.getpid() % 190)
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HOSTSTUB_DEVICES=str(world),
                   SETK_ALLOW_HOSTSTUB="1", SETK_LIB=os.path.join(ROOT, "_abl", "libsetk_hoststub.so"),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "scripts", "sptk", "apply_adaptive_beamformer.py"),
             "--mask-format", "numpy", "--batch-utts", "4", "--profile", f"{td}/prof.json",
             f"{td}/wav.scp", f"{td}/mask.scp", f"{td}/out"],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    errs = []
    for p in procs:
        _, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        errs.append(e)
    assert sorted(os.listdir(f"{td}/out")) == sorted(f"u{i}.wav" for i in range(len(lens)))
    assert any(f"Processed {len(lens)} utterances out of {len(lens)}" in e for e in errs), errs[0][-1500:]
    per_rank = [json.load(open(f"{td}/prof.json.rank{k}")) for k in range(world)]
    assert sum(p["utts"] for p in per_rank) == len(lens)
    assert all(p["mode"] == "pipeline" and p["world"] == world for p in per_rank)
    assert all(p["dist_backend"] in ("rccl", "tcp") and not p["torch_loaded"] for p in per_rank)
    loads = np.array([p["assigned_samples"] for p in per_rank], dtype=np.float64)
    assert loads.sum() == sum(lens)
    assert np.abs(loads / loads.mean() - 1).max() < 0.02, loads


def test_launcher_requeues_what_a_dead_rank_left(tmp_path):
    """`python -m setk_amd.launch --nproc 2`: in the first attempt rank 1 dies (os._exit, no
    clean-up) after its second wave file, which takes the attempt down; the launcher starts the
    job again with --skip-existing --requeue, the new attempt deals ONLY the missing utterances
    over both ranks, and at the end every utterance exists exactly once and complete."""
    import json
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "hoststub", "build.sh")], capture_output=True,
                       text=True, timeout=900, env=dict(os.environ, PLAIN="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    td = str(tmp_path)
    lens = [16000, 64000, 8000, 30011, 16000, 12345, 48000, 9000, 20000, 21000, 22000, 23000]
    _make_table(td, lens)
    env = dict(os.environ, HOSTSTUB_DEVICES="2", SETK_ALLOW_HOSTSTUB="1", OMP_NUM_THREADS="1",
               SETK_LIB=os.path.join(ROOT, "_abl", "libsetk_hoststub.so"), SETK_FAULT_INJECT="1:2:0", SETK_TESTING="1",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "setk_amd.launch", "--nproc", "2", "--retries", "2",
                        os.path.join(ROOT, "scripts", "sptk", "apply_adaptive_beamformer.py"),
                        "--mask-format", "numpy", "--batch-utts", "1", "--pipeline-depth", "1",
                        "--profile", f"{td}/prof.json", f"{td}/wav.scp", f"{td}/mask.scp", f"{td}/out"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "attempt 1 of 3" in r.stderr and "re-queueing what is missing" in r.stderr
    assert "attempt 2 of 3" in r.stderr and "attempt 3 of 3" not in r.stderr
    assert "SETK_FAULT_INJECT: rank 1 dies after 2 files" in r.stderr
    out = sorted(os.listdir(f"{td}/out"))
    assert out == sorted(f"u{i}.wav" for i in range(len(lens))), out     # no .part left, nothing missing
    from setk_amd.sptk.apply_adaptive_beamformer import _complete_wav
    assert all(_complete_wav(f"{td}/out/{f}") for f in out)
    # the second attempt dealt only what was missing, over BOTH ranks
    per_rank = [json.load(open(f"{td}/prof.json.rank{k}")) for k in range(2)]
    redone = sum(p["utts"] for p in per_rank)
    assert 0 < redone <= len(lens) - 2 and all(p["utts"] >= 1 for p in per_rank), per_rank
    assert "already in" in r.stderr   # --skip-existing saw the first attempt's files
