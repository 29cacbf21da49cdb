"""
Contract tests of the driver-facing entry points: `__graft_entry__.smoke()` and bench.py's
JSON line (single rank, two ranks, eight ranks sharing the one GPU).  They spawn subprocesses,
a profiler child and multi-rank launches, so they sort LAST: `pytest -m gpu -x` reaches every
parity test before them.  They assert the CONTRACT only -- keys, types, identities between the
fields, exact-parity flags -- and never a clock, a roofline fraction or a timing bound: those
depend on the box (round 5's record went red on a clock estimate of a 6-utterance launch).
Counter-derived fields are checked only when the profiler delivered them.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["--steps", "2", "--warmup", "1", "--utts", "6", "--seconds", "4", "--full-batch", "12"]
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                 "scaling", "vs_baseline", "dtype", "data", "config", "roofline")


def _json_line(r):
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


def test_bench_contract_single_rank():
    """bench.py prints ONE JSON line with the contract's fields, the roofline and cpu_baseline
    objects, and the flat scalars of the record."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--e2e-utts", "6", "--cpu-sample", "2"] + SMALL,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    one = _json_line(r)
    for k in CONTRACT_KEYS:
        assert k in one, k
    assert one["n_gpus"] == 1 and one["steps"] == 2 and one["warmup"] == 1 and one["scaling"] == "weak"
    assert one["vs_baseline"] is None and one["higher_is_better"] is True and one["dtype"] == "f32"
    audio = 6 * 4.0 * 2
    assert abs(one["value"] * one["ms_per_step"] * 2 / 1e3 - audio) / audio < 1e-3
    roof = one["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "alg_bytes_per_launch", "kernel_ms"):
        assert k in roof, k
    assert roof["peak"] == 8000.0 and roof["unit"] == "GB/s" and roof["kernel_ms"] > 0
    assert roof["bound"] in ("hbm", "mfma")     # the contract's vocabulary; what binds first: roof["co_limit"]
    assert abs(roof["achieved"] - roof["alg_bytes_per_launch"] / (roof["kernel_ms"] * 1e-3) / 1e9) \
        <= 0.01 * roof["achieved"] + 0.1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    if "pmc_method" in roof:
        # counters were collected in this very run (rocprofv3 --pmc on a child of the same workload)
        for key in ("pass1", "pass2"):
            assert roof[key]["alg_bytes_per_launch"] > 0
            if "traffic" in roof[key]["hbm"]:
                assert roof[key]["hbm"]["traffic"] > 0
        if roof["traffic"] is not None:
            assert roof["traffic"] == roof["pass1"]["hbm"]["traffic"]
    else:
        assert "pmc" in roof   # says why (profiler absent / refused on this box)
    cpu = one["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cpu, k
    assert cpu["cores"] == 1 and cpu["kind"] in ("port", "reference") and cpu["value"] > 0
    # the oracle check of the timed configuration ran (bench.py exits non-zero on a mismatch)
    pc = cpu["parity_check"]
    assert pc["worst_rel_rms_vs_oracle"] < pc["tol"] == 1e-3 and pc["copies_bit_identical_to_their_source"] is True
    # the same shard as 16-bit PCM frames: its own block, exact parity with the float32 path
    i16 = one["int16_ingest"]
    assert i16["status"] == "ok" and i16["bit_identical_to_float32_path_on_pcm_over_32768"] is True
    assert "2 C N" in i16["roofline"]["algorithmic_bytes"] and i16["roofline"]["bound"] == "hbm"
    assert one["full_batch"]["utts"] == 12
    e2e = one["end_to_end"]
    assert "error" not in e2e, e2e
    assert e2e["sizes"]["6"]["written"] == [6, 6]
    # the flat scalars: at the top level and inside the contract's roofline object
    for k in ("stage1_ms", "stage3_ms", "steady_state_ms_per_step", "pcm16_from_frames_ms_per_step", "pcm16_from_frames_value",
              "pcm16_from_frames_roofline_frac", "pass2_traffic_over_algorithmic", "e2e_process_rtf"):
        assert k in one and k in roof, k
        assert one[k] == roof[k]
    assert one["stage1_ms"] == one["stage_ms"]["stft_covar"] and one["e2e_process_rtf"] > 0
    assert one["pcm16_from_frames_ms_per_step"] == i16["ms_per_step"]


def test_bench_contract_aux_legs():
    """`--aux 1` (tools/bench_aux.py): the same line with the auxiliary legs added."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--aux", "1", "--e2e-utts", "6", "--cpu-sample", "1",
                        "--cpu-allcore-per-proc", "0", "--other-configs", "0", "--sustain-sec", "0.2", "--pmc", "0"]
                       + SMALL, capture_output=True, text=True, timeout=900, cwd=ROOT)
    one = _json_line(r)
    assert one["sustained"]["steps"] >= 50 and one["uncached_call"]["ms_per_step"] > 0
    e2e = one["end_to_end"]
    assert "error" not in e2e, e2e
    assert e2e["sizes"]["6"]["written"] == [6, 6, 6] and e2e["sizes"]["96"]["written"] == [96] * 5
    for k in ("median", "min", "max", "spread"):
        assert k in e2e["sizes"]["96"]["process_rtf"]
    # (the marginal rate is a difference of two process clocks: present, and None when these
    #  miniature sizes drown in the noise -- its VALUE is never asserted)
    assert "marginal_GBps_in" in e2e and "marginal_ms_per_utt" in e2e
    assert e2e["host_copy_GBps"]["1"] > 0
    assert one["roofline"]["pmc"] is None and one["roofline"]["traffic"] is None   # --pmc 0


def test_bench_contract_two_ranks():
    """The two-rank launch (as the driver starts it, but both ranks on this box's single GPU
    with a gloo rendezvous) aggregates over ranks and reports from rank 0 only."""
    env = dict(os.environ, SETK_BENCH_SHARE_GPU="1", SETK_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "2"] + SMALL
    two = _json_line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env))
    for k in CONTRACT_KEYS:
        assert k in two, k
    assert two["n_gpus"] == 2 and two["value"] > 0
    # whole-job aggregate: audio of both ranks over the slower rank's time
    audio = 2 * 6 * 4.0 * 2
    assert abs(two["value"] * two["ms_per_step"] * 2 / 1e3 - audio) / audio < 1e-3
    assert len(two["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in two["per_rank_ms_per_step"])
    assert two["ms_per_step"] >= max(two["per_rank_ms_per_step"]) - 1e-3

    # `python bench.py --gpus 2` with no launcher around it starts its two ranks itself
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, capture_output=True,
                       text=True, timeout=900, cwd=ROOT, env=env)
    assert _json_line(r)["n_gpus"] == 2
    assert "[bench rank 0/2]" in r.stderr and "[bench rank 1/2]" in r.stderr
    # a launcher whose world size disagrees with --gpus is refused
    bad = subprocess.run(cmd[:-len(SMALL) - 1] + ["3"] + SMALL, capture_output=True, text=True,
                         timeout=900, cwd=ROOT, env=env)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in bad.stderr


def test_bench_eight_ranks_at_the_full_shard_size():
    """World size 8 without 8 GPUs: `python bench.py --gpus 8` exactly as the driver would
    start it (default shard: 125 utterances of 8-ch 30 s per rank = BASELINE configs[2]'s 1000
    over the job), the eight ranks sharing this box's one GPU behind a gloo rendezvous.  The
    contract's aggregation is what is checked: eight per-rank times, the job's time never less
    than the slowest rank's, the value = audio of all ranks over it."""
    env = dict(os.environ, SETK_BENCH_SHARE_GPU="1", SETK_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "3", "--warmup", "1", "--distinct", "2"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    rec = _json_line(r)
    assert rec["n_gpus"] == 8 and rec["steps"] == 3 and rec["scaling"] == "weak"
    per = rec["per_rank_ms_per_step"]
    assert len(per) == 8 and all(v > 0 for v in per)
    assert rec["ms_per_step"] >= max(per) - 1e-3
    cfg = rec["config"]
    assert cfg["utts_per_gpu"] == 125 and cfg["channels"] == 8 and cfg["seconds"] == 30.0
    assert "1000 at 8 GPUs" in cfg["workload"] and cfg["parallelism"] == "utterance-sharded x8"
    audio = 8 * 125 * 30.0
    assert abs(rec["value"] * rec["ms_per_step"] / 1e3 - audio) / audio < 1e-3
    assert abs(rec["per_gpu_value"] * 8 - rec["value"]) / rec["value"] < 1e-3
    for k in range(8):
        assert f"[bench rank {k}/8]" in r.stderr
    # nothing that belongs to one GPU's record leaks into the multi-rank line
    for key in ("cpu_baseline", "full_batch", "other_configs", "end_to_end", "int16_ingest"):
        assert key not in rec
