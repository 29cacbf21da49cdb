"""
The reference's own CPU numbers for BASELINE's metric: the UNMODIFIED
scripts/sptk/apply_adaptive_beamformer.py (loaded from /root/reference through
oracle/ref_harness.py) on synthetic PCM16 wav + numpy masks,

  * one process, BLAS threads pinned to 1            -> the "1-core" number
  * nj = nproc processes over disjoint scp shards    -> the reference's parallel mode
    (scripts/run_adapt_beamformer.sh:69-92, `run.pl JOB=1:nj`)

wall clock from the first scp read to the last wav close.

*** TEST / MEASUREMENT INFRASTRUCTURE -- NOT PRODUCT CODE ***
Only bench.py's `cpu_baseline` leg and tools/ call this, and only where
/root/reference exists (the build container); on the GPU box bench.py reports
"reference absent on this box" and quotes the record committed under profiles/.

    python -m oracle.ref_cpu_leg --utts 8 --out profiles/r04_ref_cpu_leg.json
"""
import argparse
import json
import os
import platform
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SR = 16000

_WORKER = r"""
import os, sys, time, json, argparse
for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[v] = "1"
sys.path.insert(0, sys.argv[1])
from oracle import ref_harness as rh
cli = rh.load_cli("apply_adaptive_beamformer")
td, shard, kind, start_at = sys.argv[2], sys.argv[3], sys.argv[4], float(sys.argv[5])
args = argparse.Namespace(
    wav_scp=os.path.join(td, f"wav.{shard}.scp"), tgt_mask=os.path.join(td, f"mask.{shard}.scp"),
    dst_dir=os.path.join(td, f"enh.{shard}"), itf_mask="", fmt="numpy", beamformer=kind,
    pmwf_ref=-1, sr=16000, ban=False, rank1_appro="", mask=False, vad_proportion=1, alpha=0.8,
    chunk_size=-1, channels=4, frame_len=512, frame_hop=256, center=True,
    round_power_of_two=True, window="hann")
ready = time.time()
while time.time() < start_at:
    time.sleep(0.005)
t0 = time.time()
cli.run(args)
print(json.dumps(dict(t0=t0, t1=time.time(), ready=ready)))
"""


def _host():
    cpu = platform.processor()
    try:
        with open("/proc/cpuinfo") as f:
            cpu = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")][0]
    except Exception:
        pass
    import scipy
    return {"cpu": cpu, "logical_cores": os.cpu_count(), "numpy": np.__version__,
            "scipy": scipy.__version__}


def _write_inputs(td, n, C, N):
    import scipy.io.wavfile
    from . import np_oracle as o
    nd = min(n, 2)
    for i in range(nd):
        mix, sp, nz = o.synth_utterance(i, C, N, return_parts=True)
        scipy.io.wavfile.write(os.path.join(td, f"u{i}.wav"), SR,
                               np.rint(mix.T.astype(np.float64) * 32767).astype(np.int16))
        np.save(os.path.join(td, f"u{i}.npy"), o.irm_mask(sp, nz))
    return nd


_PORT_WORKER = r"""
import os, sys, time, json
for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[v] = "1"
sys.path.insert(0, sys.argv[1])
from oracle import np_oracle as o
n, C, N, kind = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
utts = []
for i in range(min(n, 2)):
    mix, sp, nz = o.synth_utterance(i, C, N, return_parts=True)
    utts.append((mix, o.irm_mask(sp, nz)))
o.enhance_utterance(*utts[0], kind=kind)
t0 = time.time()
for i in range(n):
    o.enhance_utterance(*utts[i % len(utts)], kind=kind)
print(json.dumps(dict(wall=time.time() - t0)))
"""


def _port_one_core(n, C, N, kind):
    """bench.py's `cpu_baseline` port (oracle/np_oracle.enhance_utterance, compute only, 1 thread)
    on the SAME cores in the SAME run: the ratio port : reference that bench.py applies on the GPU
    box, where only the port can run."""
    r = subprocess.run([sys.executable, "-c", _PORT_WORKER, ROOT, str(n), str(C), str(N), kind],
                       capture_output=True, text=True, timeout=3600)
    if r.returncode != 0:
        raise RuntimeError("port worker failed: " + r.stderr[-800:])
    return json.loads(r.stdout.strip().splitlines()[-1])["wall"]


def _run(td, shards, kind):
    start_at = time.time() + 6.0 + 0.05 * len(shards)
    procs = [subprocess.Popen([sys.executable, "-c", _WORKER, ROOT, td, str(s), kind, repr(start_at)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for s in shards]
    res = []
    for p in procs:
        o_, e_ = p.communicate(timeout=3600)
        if p.returncode != 0:
            raise RuntimeError("reference CLI failed: " + e_[-800:])
        res.append(json.loads(o_.strip().splitlines()[-1]))
    return max(r["t1"] for r in res) - min(r["t0"] for r in res), sum(r["ready"] > start_at for r in res)


def measure(utts=8, channels=8, seconds=30.0, kind="mvdr", nj=None, per_proc=2):
    """Returns the record (dict).  Raises if /root/reference is absent."""
    from . import ref_harness as rh
    if not rh.available():
        raise RuntimeError("reference absent on this box")
    C, N = channels, int(round(seconds * SR))
    nj = nj or (os.cpu_count() or 1)
    with tempfile.TemporaryDirectory(prefix="setk_refleg_") as td:
        nd = _write_inputs(td, utts, C, N)

        def shard(name, n):
            with open(os.path.join(td, f"wav.{name}.scp"), "w") as ws, \
                    open(os.path.join(td, f"mask.{name}.scp"), "w") as ms:
                for i in range(n):
                    ws.write(f"{name}_{i} {td}/u{i % nd}.wav\n")
                    ms.write(f"{name}_{i} {td}/u{i % nd}.npy\n")
        shard("one", utts)
        wall1, _ = _run(td, ["one"], kind)
        assert len(os.listdir(os.path.join(td, "enh.one"))) == utts
        for j in range(nj):
            shard(f"j{j}", per_proc)
        walln, late = _run(td, [f"j{j}" for j in range(nj)], kind)
    wallp = _port_one_core(utts, C, N, kind)
    ref1 = utts * seconds / wall1
    port1 = utts * seconds / wallp
    return {
        "port_one_core": {"value": round(port1, 2), "cores": 1, "utts": utts, "wall_s": round(wallp, 2),
                          "what": "oracle/np_oracle.enhance_utterance, compute only, same cores, same run"},
        "port_over_reference": {"one_core": round(port1 / ref1, 3)},
        "what": "unmodified reference scripts/sptk/apply_adaptive_beamformer.py through "
                "oracle/ref_harness.py (five import shims, no source change), PCM16 wav + numpy "
                "masks in, PCM16 wav out, first scp read to last wav close",
        "workload": f"{C}-ch {seconds:g} s, {kind}, STFT 512/256/hann/center",
        "unit": "x real time (audio seconds per wall second)",
        "one_core": {"value": round(utts * seconds / wall1, 2), "cores": 1, "utts": utts,
                     "wall_s": round(wall1, 2), "s_per_utt": round(wall1 / utts, 3)},
        "all_cores": {"value": round(nj * per_proc * seconds / walln, 2), "cores": nj,
                      "utts": nj * per_proc, "wall_s": round(walln, 2), "late_workers": late},
        "host": _host(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=8)
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--beamformer", default="mvdr")
    ap.add_argument("--nj", type=int, default=0)
    ap.add_argument("--per-proc", type=int, default=2)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    rec = measure(a.utts, a.channels, a.seconds, a.beamformer, a.nj or None, a.per_proc)
    txt = json.dumps(rec, indent=1)
    print(txt)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
