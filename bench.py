#!/usr/bin/env python
"""
bench.py -- real-time factor of the mask-based MVDR hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
8-ch 16 kHz oracle-mask MVDR on 30 s utterances, STFT 512/256/hann/center.
Every rank owns the per-GPU shard of that configuration -- 125 utterances, i.e.
1000 utterances at 8 GPUs -- so the scaling is WEAK and N = 8 reproduces
configs[2] exactly.  A "step" is one pass of the whole hot path (STFT ->
covariance -> MVDR solve -> beamform -> iSTFT -> renorm) over the rank's shard,
inputs resident in HBM as float32 C x N when the timed region starts.  Utterances
shard independently: RCCL carries only the barriers and the max-over-ranks reduction.

What the default command runs besides the K timed steps (N = 1, rank 0):
  int16_ingest  the same shard resident as the wave files' interleaved 16-bit frames
                (what every CLI user starts from); its step time is the headline's
                sibling `pcm16_from_frames_*`
  roofline      HIP-event kernel times of the timed steps + ONE profiled child
                (rocprofv3 --pmc, three counter passes over float32 and PCM16 steps)
  cpu_baseline  the numpy oracle on one host core over a bounded sample, and the
                parity check of every distinct utterance of the timed shard
  full_batch    all 1000 utterances of configs[2] on the one GPU
  end_to_end    192 files, disk -> wav through the drop-in CLI (process wall clock)
Everything else (sustained stepping, power, issue rates, the other BASELINE configs, the
all-core CPU leg, the three-repeat end-to-end marginal rate) is `--aux 1`: tools/bench_aux.py.

The scalars a reader needs first are flat, at the top level AND inside `roofline` (the
driver's record keeps the scalars of the contract's objects and only the names of the rest).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
SR = 16000
F = 257
SIMDS = 256 * 4            # 256 CUs x 4 SIMD-32
XCDS = 8
VALU_CYCLES_PER_INST = 2   # a wave64 VALU instruction occupies its SIMD-32 for two cycles
# measured issue rate of a SIMD shared by n waves, cycles per plain fp32 VALU instruction
# (profiles/r04f_valu_rate_pinned.txt; --aux 1 re-measures it in the run)
ISSUE_CYCLES_AT_WAVES = {1: 7.6, 2: 3.6, 3: 2.67, 4: 2.24}
# kernels of the step: key -> (name fragments, 16-bit PCM input form? None = either)
KERNELS = {"pass1": (["stft_covar_mc_kernel", "stft_covar_kernel"], False),
           "pass2": (["beamform_istft_mc_kernel", "beamform_istft_kernel"], False),
           "solve": (["solve_kernel"], None),
           "pass1_pcm": (["stft_covar_kernel"], True),
           "pass2_pcm": (["beamform_istft_mc_kernel"], True),
           "ingest": (["pcm16_deinterleave_batch_kernel"], None)}
WAVES_PER_SIMD = {"stft_covar_mc_kernel": 4, "stft_covar_kernel": 4, "beamform_istft_mc_kernel": 4,
                  "beamform_istft_kernel": 2, "solve_kernel": 2}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.1 s of warm-up and 0.4 s timed (a 40 ms region sits on the clock ramp of an idle GPU)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--utts", type=int, default=125, help="utterances per GPU")
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--beamformer", default="mvdr", choices=["mvdr", "gevd", "pmwf-0"])
    ap.add_argument("--distinct", type=int, default=16,
                    help="distinct synthetic utterances generated per rank (others are copies)")
    ap.add_argument("--cpu-sample", type=int, default=64,
                    help="utterances timed on the CPU oracle (0 = skip)")
    ap.add_argument("--full-batch", type=int, default=1000,
                    help="N=1 only: also time the whole configs[2] batch of this many utterances on the "
                         "one GPU (strong-scaling anchor; 0 = skip)")
    ap.add_argument("--e2e-utts", type=int, default=192,
                    help="N=1 only: files of the end-to-end CLI leg (0 = skip); --aux 1 adds 16 x as many, "
                         "three repeats each")
    ap.add_argument("--pmc", type=int, default=1,
                    help="N=1 only: HBM traffic and VALU instruction counts of the streaming kernels IN "
                         "THIS RUN: a few steps re-run as a child under `rocprofv3 --pmc` (counters only, "
                         "one pass per counter group; 0 = skip)")
    ap.add_argument("--child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--int16-ingest", type=int, default=1,
                    help="N=1 only: the same shard fed as 16-bit PCM frames (0 = skip)")
    ap.add_argument("--aux", type=int, default=0, help="N=1 only: the auxiliary legs (tools/bench_aux.py)")
    ap.add_argument("--sustain-sec", type=float, default=3.0, help="--aux: seconds of sustained stepping")
    ap.add_argument("--cpu-allcore-per-proc", type=int, default=3, help="--aux: utterances per CPU worker")
    ap.add_argument("--other-configs", type=int, default=1, help="--aux: BASELINE configs[1], [3], [4]")
    return ap.parse_args()


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks
    ourselves, one per GPU -- what scripts/run_adapt_beamformer.sh:80-92 does with
    run.pl JOB=1:nj -- and hand back the launcher's exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def stage_dict(sm):
    return {"stft_covar": round(sm[0], 4), "reduce_solve": round(sm[1], 4),
            "beamform_istft": round(sm[2], 4), "renorm": round(sm[3], 4)}


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(relaunch_under_torchrun(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} "
                         "ranks (one rank per GPU is the contract)")
    t_start = time.perf_counter()
    import torch
    import torch.distributed as dist
    from setk_amd import build as _build
    _build.build_library(force=False)  # no-op when the in-tree library is current
    from setk_amd import _ffi, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # SETK_BENCH_SHARE_GPU=1 + SETK_BENCH_BACKEND=gloo: every rank on cuda:0 with a CPU
    # rendezvous -- only for exercising the multi-rank control flow on a 1-GPU box
    # (tests/test_gpu_zz_contract.py); the real launch is one rank per GPU over RCCL.
    share = os.environ.get("SETK_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("SETK_BENCH_BACKEND", "nccl")
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    C, U = args.channels, args.utts
    N = int(round(args.seconds * SR))
    ctx = _ffi.Context(dev_index)
    ctx.stft_plan(512, 256, 512, True)
    T = ctx.num_frames(N)
    L = ctx.istft_num_samples(T)

    # ---- synthetic shard, resident in HBM --------------------------------
    audio, masks = [], []
    nd = max(1, min(args.distinct, U))
    for i in range(nd):
        mix, sp, nz = synth.synth_utterance(rank * U + i, C, N, return_parts=True)
        parts = torch.from_numpy(np.stack([sp[0], nz[0]])).to(dev)
        spec = torch.empty((2, T, F), dtype=torch.complex64, device=dev)
        ctx.stft(parts, spec)  # oracle (IRM) mask from the device STFT
        s, v = spec[0].abs(), spec[1].abs()
        audio.append(torch.from_numpy(mix).to(dev))
        masks.append((s / torch.sqrt(s * s + v * v + synth.EPSILON)).contiguous())
    for i in range(nd, U):
        audio.append(audio[i % nd].clone())
        masks.append(masks[i % nd].clone())
    waves = [torch.empty(L, dtype=torch.float32, device=dev) for _ in range(U)]
    aptr, mptr, wptr = ([t.data_ptr() for t in x] for x in (audio, masks, waves))
    ns = [N] * U
    kind = {"mvdr": _ffi.BF_MVDR, "gevd": _ffi.BF_GEVD, "pmwf-0": _ffi.BF_PMWF}[args.beamformer]
    opts = _ffi.BfOpts(kind=kind, flags=_ffi.FLAG_CLAMP_MASK, pmwf_beta=0.0, pmwf_ref=-1, rank1=0)

    def step():
        ctx.enhance_batch(opts, C, aptr, ns, mptr, None, wptr, want_status=False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    st = ctx.enhance_batch(opts, C, aptr, ns, mptr, None, wptr, want_status=True)
    if any(st) and not os.environ.get("SETK_BENCH_NOCHECK"):
        raise SystemExit(f"numerical status {st}")
    ctx.set_profiling(True)   # HIP events around each stage, on the stream the kernels run on
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.set_profiling(False)
    stage_ms = ctx.last_stage_ms()
    if args.child:
        # profiled child of pmc_leg(): the launches above and the PCM16 step's are all it is for
        int16_ingest_leg(args, ctx, _ffi, torch, opts, audio, masks, waves, C, N, T, L, U, nd, None)
        print(json.dumps({"child": True, "stage_ms": [round(x, 4) for x in stage_ms]}), flush=True)
        return
    # every rank's own clock (its device work only, no barrier): an imbalance shows here
    per_rank_ms = [round(1e3 * own_elapsed / args.steps, 4)]
    if world > 1:
        cdev = dev if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([own_elapsed], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [round(1e3 * float(x.item()) / args.steps, 4) for x in allr]
    print(f"[bench rank {rank}/{world}] cuda:{dev_index} ms_per_step {per_rank_ms[rank]:.4f}",
          file=sys.stderr, flush=True)
    value = world * U * (N / SR) * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    if rank == 0:
        b_k1 = U * (4.0 * C * N + 4.0 * T * F)            # algorithmic bytes / launch, pass 1
        b_k2 = U * (4.0 * C * N + 4.0 * L)                # pass 2: audio again + the wave
        out = {
            "metric": "real-time-factor (audio-sec/wall-sec), 8-ch 16 kHz MVDR",
            "value": round(value, 1),
            "unit": "x real time (audio seconds per wall second, all GPUs)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "per_gpu_value": round(value / world, 1),
            "per_rank_ms_per_step": per_rank_ms,
            "config": {
                "workload": f"{C}-ch 16 kHz oracle-mask {args.beamformer.upper()}, "
                            f"{args.seconds:g} s utterances, {U} per GPU "
                            f"(BASELINE configs[2] shard: {U * 8} at 8 GPUs), "
                            "STFT 512/256/hann/center, inputs resident in HBM",
                "input_format": "float32 C x N resident in HBM; a wave file's interleaved PCM16 frames pay "
                                "the de-interleave pass: pcm16_from_frames_ms_per_step",
                "utts_per_gpu": U, "channels": C, "seconds": args.seconds,
                "frames": T, "beamformer": args.beamformer,
                "parallelism": f"utterance-sharded x{world}",
            },
            "stage_ms": stage_dict(stage_ms),
        }
    if rank == 0 and world == 1:
        single_gpu_legs(args, out, ctx, _ffi, torch, synth, opts, step, audio, masks, waves,
                        C, N, T, L, U, nd, stage_ms, ms_per_step, b_k1, b_k2)
        out["bench_wall_s"] = round(time.perf_counter() - t_start, 1)
    elif rank == 0:
        out["roofline"] = build_roofline(C, b_k1, b_k2, stage_ms, None)
        out["roofline"]["pipeline_achieved"] = round(
            U * (4.0 * C * N + 4.0 * T * F + 4.0 * L) / (ms_per_step * 1e-3) / 1e9, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def single_gpu_legs(args, out, ctx, _ffi, torch, synth, opts, step, audio, masks, waves,
                    C, N, T, L, U, nd, stage_ms, ms_per_step, b_k1, b_k2):
    """Everything of the N = 1 record outside the contract's timed region."""
    dev = audio[0].device
    # outputs of the timed configuration for the oracle check in cpu_baseline(): every DISTINCT
    # utterance of the shard; the others are copies whose outputs must equal their source's bit for
    # bit -- compared on the device
    wave0 = {"waves": [waves[i].cpu().numpy() for i in range(nd)],
             "clones_bit_identical": all(bool(torch.equal(waves[i], waves[i % nd])) for i in range(nd, U)),
             "clones": U - nd}
    # the same step once the clocks have settled (the driver's 5 + 20 steps last 45 ms and sit on the
    # ramp of a GPU that idled while the host synthesised the shard): 200 more steps, outside the contract's region
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        step()
    torch.cuda.synchronize()
    steady_ms = 1e3 * (time.perf_counter() - t0) / 200
    aux = None
    if args.aux:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_aux as aux
        out.update(aux.stepping_legs(args, ctx, _ffi, torch, opts, step, audio, masks, C, N, L, U))
    int16_leg = None
    if args.int16_ingest:
        try:
            int16_leg = int16_ingest_leg(args, ctx, _ffi, torch, opts, audio, masks, waves, C, N, T, L, U, nd,
                                         stage_dict(stage_ms))
        except Exception as e:  # a side leg never takes the headline line down with it
            int16_leg = {"error": repr(e)}
    full_batch = time_full_batch(args, ctx, opts, torch, audio, masks, C, N, L) if args.full_batch > U else None
    pmc = pmc_leg(args) if args.pmc else None
    rates = aux.issue_rates_leg() if (aux and args.pmc) else None
    roof = build_roofline(C, b_k1, b_k2, stage_ms, pmc, rates)
    roof["pipeline_achieved"] = round(U * (4.0 * C * N + 4.0 * T * F + 4.0 * L) / (ms_per_step * 1e-3) / 1e9, 1)
    if rates is not None:
        roof["issue_rates"] = rates
    out["roofline"] = roof
    if args.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(args, C, N, 0, wave0)
        if aux and args.cpu_allcore_per_proc > 0:
            allc = aux.cpu_allcore(args, C, N)
            if allc:
                out["cpu_baseline"]["all_cores"] = dict(allc, unit=out["cpu_baseline"]["unit"])
    flat = {"stage1_ms": round(stage_ms[0], 4), "stage3_ms": round(stage_ms[2], 4),
            "steady_state_ms_per_step": round(steady_ms, 4)}
    p2 = (roof.get("pass2") or {}).get("hbm", {})
    flat["pass2_traffic_over_algorithmic"] = p2.get("traffic_over_algorithmic")
    if int16_leg is not None:
        if "error" not in int16_leg:
            add_pcm_counters(int16_leg, pmc, U, C, N, T, L)
            flat.update(pcm16_from_frames_ms_per_step=int16_leg["ms_per_step"],
                        pcm16_from_frames_value=int16_leg["value"],
                        pcm16_from_frames_roofline_frac=int16_leg["roofline"]["frac"],
                        pcm16_pass2_traffic_over_algorithmic=int16_leg["roofline"]["pass2"].get(
                            "traffic_over_algorithmic"))
        out["int16_ingest"] = int16_leg
    if full_batch is not None:
        out["full_batch"] = full_batch
        flat["full_batch_value"] = full_batch["value"]
    del audio[:], masks[:], waves[:]
    torch.cuda.empty_cache()
    if aux and args.other_configs:
        try:
            out["other_configs"] = aux.other_configs(torch, _ffi, synth, dev, args if args.pmc else None, rates)
        except Exception as e:
            out["other_configs"] = {"error": repr(e)}
    if args.e2e_utts > 0:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_aux
            e2e = bench_aux.end_to_end(args, C, N, full=bool(args.aux))
        except Exception as e:
            e2e = {"error": repr(e)}
        out["end_to_end"] = e2e
        flat["e2e_process_rtf"] = e2e.get("process_rtf")
        flat["e2e_marginal_GBps_in"] = (e2e.get("marginal_GBps_in") or {}).get("median") \
            if isinstance(e2e.get("marginal_GBps_in"), dict) else None
    out.update(flat)
    roof.update(flat)   # (scalars of the contract's objects survive in the driver's record)


def pmc_leg(args, child=None, kernels=None):
    """Counters of THIS run's workload (or of `child`, a command line, for the kernels
    `kernels` = {key: (name fragments, pcm)}): a few steps of the same configuration
    re-run as a child under `rocprofv3 --pmc`, one pass per counter group (counters
    only -- never mixed with API tracing), parsed from the counter_collection csv.
    HBM bytes follow MI355X_MICROARCH.md (HBM section): read = 2 x FETCH_SIZE KB (gfx950
    tallies the 128-byte requests of a coalesced stream at 64 bytes), write = WRITE_SIZE KB.
    The effective clock of the profiled launches is GRBM_GUI_ACTIVE / 8 XCDs / kernel time."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    kernels = {k: (v if isinstance(v, tuple) else (v, None)) for k, v in (kernels or KERNELS).items()}
    child = child or [sys.executable, os.path.abspath(__file__), "--child", "1", "--gpus", "1",
                      "--steps", "3", "--warmup", "1", "--utts", str(args.utts), "--channels", str(args.channels),
                      "--seconds", str(args.seconds), "--beamformer", args.beamformer,
                      "--distinct", str(min(args.distinct, 2))]
    groups = [["FETCH_SIZE"], ["WRITE_SIZE"],
              ["SQ_INSTS_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES",
               "SQ_VALU_MFMA_COEXEC_CYCLES"]]
    acc = {}
    td = tempfile.mkdtemp(prefix="setk_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.perf_counter()
    try:
        for gi, g in enumerate(groups):
            out_dir = os.path.join(td, f"g{gi}")
            cmd = [exe, "--pmc"] + g + ["--output-format", "csv", "-d", out_dir, "--"] + child
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd="/tmp", env=env)
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"error": f"rocprofv3 --pmc {' '.join(g)} failed (rc {r.returncode}): "
                                 + (r.stderr or r.stdout)[-300:]}
            for fn in files:
                for row in csv.DictReader(open(fn)):
                    name = row["Kernel_Name"]
                    if ", true, false>" in name:   # stft_covar_kernel<C, true, false>: setk_stft's dump
                        continue
                    # the streaming kernels' last template argument: 16-bit PCM input
                    is_pcm = ", true>(" in name or name.rstrip().endswith(", true>")
                    for key, (knames, pcm) in kernels.items():
                        kname = next((k for k in knames if k + "<" in name), None)
                        if kname and (pcm is None or pcm == is_pcm):
                            acc.setdefault(key, {})["__kernel__"] = kname
                            acc[key].setdefault(row["Counter_Name"], []).append(
                                (float(row["Counter_Value"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    finally:
        shutil.rmtree(td, ignore_errors=True)
    res = {"method": "in-run: this workload re-run for a few steps under rocprofv3 --pmc, one pass per "
                     "counter group; read = 2 x FETCH_SIZE KB, write = WRITE_SIZE KB "
                     "(MI355X_MICROARCH.md HBM section); clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time",
           "seconds_spent": round(time.perf_counter() - t0, 1)}
    for key in kernels:
        c = acc.get(key, {})
        mean = lambda name: (sum(v for v, _ in c[name]) / len(c[name])) if c.get(name) else None  # noqa: E731
        dur = lambda name: (sum(d for _, d in c[name]) / len(c[name])) if c.get(name) else None  # noqa: E731
        rd, wr = mean("FETCH_SIZE"), mean("WRITE_SIZE")
        gui, dns = mean("GRBM_GUI_ACTIVE"), dur("GRBM_GUI_ACTIVE")
        res[key] = {
            "kernel": c.get("__kernel__"),
            "mfma_insts": mean("SQ_INSTS_MFMA"), "mfma_busy_cycles": mean("SQ_VALU_MFMA_BUSY_CYCLES"),
            "mfma_valu_coexec_cycles": mean("SQ_VALU_MFMA_COEXEC_CYCLES"),
            "hbm_read_bytes": None if rd is None else 2.0 * rd * 1024.0,
            "hbm_write_bytes": None if wr is None else wr * 1024.0,
            "valu_insts": mean("SQ_INSTS_VALU"), "waves": mean("SQ_WAVES"),
            "profiled_kernel_ms": None if dns is None else round(dns / 1e6, 4),
            # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs of the part
            "clock_ghz": None if not (gui and dns) else round(gui / XCDS / dns, 3),
            "launches": len(c.get("SQ_INSTS_VALU", [])),
        }
    return res


def build_roofline(C, b_k1, b_k2, stage_ms, pmc, rates=None):
    """The `roofline` object: the contract's HBM figures for the STFT+covariance kernel
    (algorithmic bytes per launch / mean HIP-event kernel time of the timed steps / 8 TB/s),
    the counter traffic, and for both streaming kernels the VALU-issue floor they sit under
    (DESIGN section 5): wave-instructions / 1024 SIMDs x cycles per instruction / clock.
    `ceiling_ms` is the co-limit: the larger of the HBM time of the algorithmic bytes and the
    issue time of the kernel's own instruction stream at its occupancy (four waves per SIMD:
    2.24 cycles per instruction, measured); `frac_of_ceiling` = ceiling_ms / kernel_ms says how
    much of what this instruction stream allows the kernel reaches, `frac` stays the HBM figure."""
    k1_ms, solve_ms, k2_ms = stage_ms[0], stage_ms[1], stage_ms[2]
    achieved = b_k1 / (k1_ms * 1e-3) / 1e9
    k1name = ((pmc or {}).get("pass1") or {}).get("kernel") or "stft_covar_kernel"
    roof = {"kernel": f"{k1name}<{C}>" if "_mc_" in k1name else f"{k1name}<{C}, false>", "bound": "hbm",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "alg_bytes_per_launch": b_k1, "kernel_ms": round(k1_ms, 4)}
    if not pmc or "error" in pmc:
        roof["pmc"] = pmc
        return roof
    roof["pmc_method"] = pmc["method"]
    roof["pmc_seconds_spent"] = pmc["seconds_spent"]
    measured = (rates or {}).get("cycles_per_inst_at_waves") or {}
    for key, alg, kms in (("pass1", b_k1, k1_ms), ("pass2", b_k2, k2_ms), ("solve", None, solve_ms)):
        p = pmc.get(key) or {}
        kname = p.get("kernel") or KERNELS[key][0][-1]
        if alg is None:
            # the solve: 32 125 small dense problems, no streaming traffic to speak of; its stage
            # time includes covar_finalize_kernel, the floor is priced on the profiled duration
            if not p.get("valu_insts"):
                continue
            ent = {"kernel": kname, "stage_ms_with_partial_reduce": round(solve_ms, 4)}
        else:
            ent = {"kernel": kname, "kernel_ms": round(kms, 4), "alg_bytes_per_launch": alg,
                   "hbm": {"achieved": round(alg / (kms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                           "frac": round(alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
            if p.get("hbm_read_bytes") is not None and p.get("hbm_write_bytes") is not None:
                ent["hbm"]["traffic"] = round(p["hbm_read_bytes"] + p["hbm_write_bytes"])
                ent["hbm"]["traffic_over_algorithmic"] = round(ent["hbm"]["traffic"] / alg, 3)
        if p.get("valu_insts") and p.get("clock_ghz") and p.get("profiled_kernel_ms"):
            pk = p["profiled_kernel_ms"]
            floor_ms = p["valu_insts"] / SIMDS * VALU_CYCLES_PER_INST / (p["clock_ghz"] * 1e9) * 1e3
            waves = WAVES_PER_SIMD[kname]
            cpi = measured.get(waves, ISSUE_CYCLES_AT_WAVES[waves])
            occ_floor = floor_ms * cpi / VALU_CYCLES_PER_INST
            ent["valu_issue"] = {"insts": round(p["valu_insts"]), "clock_ghz": p["clock_ghz"],
                                 "floor_ms": round(floor_ms, 4), "profiled_kernel_ms": pk,
                                 "frac": round(floor_ms / pk, 4),
                                 "at_occupancy": {"waves_per_simd": waves, "cycles_per_inst": cpi,
                                                  "floor_ms": round(occ_floor, 4), "frac": round(occ_floor / pk, 4)}}
            if p.get("mfma_insts"):
                # the matrix pipe of a SIMD is busy SQ_VALU_MFMA_BUSY_CYCLES / 1024 cycles; what of
                # it overlaps VALU issue is SQ_VALU_MFMA_COEXEC_CYCLES
                mf_ms = p["mfma_busy_cycles"] / SIMDS / (p["clock_ghz"] * 1e9) * 1e3
                co_ms = (p.get("mfma_valu_coexec_cycles") or 0.0) / SIMDS / (p["clock_ghz"] * 1e9) * 1e3
                ent["mfma"] = {"insts": round(p["mfma_insts"]), "op": "v_mfma_f32_16x16x32_f16",
                               "busy_ms_per_simd": round(mf_ms, 4), "coexec_with_valu_ms": round(co_ms, 4),
                               "valu_plus_mfma_floor_ms": round(occ_floor + mf_ms - co_ms, 4),
                               "frac_of_that_floor": round((occ_floor + mf_ms - co_ms) / pk, 4)}
            if alg is not None:
                hbm_ms = alg / (HBM_PEAK_GBS * 1e9) * 1e3
                ceil_ms = max(hbm_ms, occ_floor + (ent.get("mfma", {}).get("busy_ms_per_simd", 0.0)
                                                   - ent.get("mfma", {}).get("coexec_with_valu_ms", 0.0)))
                ent["ceiling_ms"] = round(ceil_ms, 4)
                ent["ceiling_is"] = "hbm" if ceil_ms == hbm_ms else "valu_issue at occupancy (+ exposed MFMA)"
                ent["frac_of_ceiling"] = round(ceil_ms / pk, 4)
        roof[key] = ent
    p1 = roof.get("pass1", {})
    if "traffic" in p1.get("hbm", {}):
        roof["traffic"] = p1["hbm"]["traffic"]
        roof["traffic_over_algorithmic"] = p1["hbm"]["traffic_over_algorithmic"]
    if "ceiling_ms" in p1:
        roof.update(ceiling_ms=p1["ceiling_ms"], ceiling_is=p1["ceiling_is"], frac_of_ceiling=p1["frac_of_ceiling"],
                    valu_insts=p1["valu_issue"]["insts"], clock_ghz=p1["valu_issue"]["clock_ghz"])
        if p1["ceiling_is"] != "hbm":
            # (`bound` keeps the contract's vocabulary -- hbm | mfma -- and the HBM figures; what binds
            #  the kernel before HBM does is named beside it)
            roof["co_limit"] = "valu_issue"
            roof["bound_note"] = ("bound / achieved / peak / frac are the contract's HBM figures; the kernel's own "
                                  "instruction stream needs ceiling_ms at full issue, more than the HBM time of "
                                  "its bytes (DESIGN section 5)")
    return roof


def int16_ingest_leg(args, ctx, _ffi, torch, opts, audio, masks, waves, C, N, T, L, U, nd, f32_stage_ms):
    """The timed configuration with its audio as 16-bit PCM, as a wave file stores it (SURVEY
    8f-3 "int16 ingest on device", 8d "report that variant separately with 2 C N").  Resident
    in HBM: the interleaved frames [N][C] of every utterance.  One step = de-interleave into
    planar int16 (setk_pcm16_deinterleave_batch) + the four stages with SETK_FLAG_IN_PCM16
    (both streaming kernels read 2 bytes per sample; read_wav's / 32768 is folded into their
    window tables).  `enhance_only`: the planar samples already resident.  The parity check is
    exact: the outputs equal the float32 path's on pcm / 32768 bit for bit."""
    dev = audio[0].device
    stride = ctx.pcm16_channel_stride(N)
    frames, f32q = [], []
    for i in range(nd):
        q = torch.clamp(torch.round(audio[i].T * 32767.0), -32768, 32767).to(torch.int16).contiguous()  # [N][C]
        frames.append(q)
        f32q.append((q.T.to(torch.float32) / 32768.0).contiguous())
    frames += [frames[i % nd].clone() for i in range(nd, U)]
    planar = [torch.empty((C, stride), dtype=torch.int16, device=dev) for _ in range(U)]
    fptr, pptr, mptr, wptr = ([t.data_ptr() for t in x] for x in (frames, planar, masks, waves))
    ns = [N] * U
    po = _ffi.BfOpts(kind=opts.kind, flags=opts.flags | _ffi.FLAG_IN_PCM16, pmwf_beta=opts.pmwf_beta,
                     pmwf_ref=opts.pmwf_ref, rank1=opts.rank1)

    def step(ingest=True):
        if ingest:
            ctx.pcm16_deinterleave_batch(C, fptr, ns, pptr)
        ctx.enhance_batch(po, C, pptr, ns, mptr, None, wptr, want_status=False)

    def timed(fn, warm, k):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / k

    warm, k = (1, 3) if args.child else (max(10, args.warmup // 2), max(20, args.steps // 2))
    step()
    st = ctx.enhance_batch(po, C, pptr, ns, mptr, None, wptr, want_status=True)
    ms_full = timed(step, warm, k)
    if args.child:
        return None
    ctx.set_profiling(True)
    ms_enh = timed(lambda: step(False), warm, k)
    ctx.set_profiling(False)
    stage = ctx.last_stage_ms()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        ctx.pcm16_deinterleave_batch(C, fptr, ns, pptr)
    ev1.record()
    torch.cuda.synchronize()
    ms_ingest = ev0.elapsed_time(ev1) / 20
    # exact parity against the float32 path on the dequantised samples (distinct utterances)
    step()
    torch.cuda.synchronize()
    got = [waves[i].clone() for i in range(nd)]
    # (the same batch of U utterances: the work list -- hence the order in which an utterance's
    #  partial covariance slabs are summed -- depends on the batch)
    ctx.enhance_batch(opts, C, [f32q[i % nd].data_ptr() for i in range(U)], ns, mptr, None, wptr, want_status=True)
    torch.cuda.synchronize()
    identical = all(bool(torch.equal(got[i], waves[i])) for i in range(nd))
    b_k1 = U * (2.0 * C * N + 4.0 * T * F)
    b_k2 = U * (2.0 * C * N + 4.0 * L)
    return {
        "what": f"{U} x {C}-ch x {N / SR:g} s, audio resident as the wave files' interleaved 16-bit frames; "
                "step = de-interleave (planar int16, no float32 copy) + the four stages reading int16",
        "status": "ok" if st is not None and not any(st) else str(st),
        "ms_per_step": round(ms_full, 4),
        "value": round(U * (N / SR) / (ms_full * 1e-3), 1),
        "enhance_only_ms": round(ms_enh, 4),
        "ingest_ms": round(ms_ingest, 4),
        "ingest_gbs": round(U * 4.0 * C * N / (ms_ingest * 1e-3) / 1e9, 1),
        "stage_ms": stage_dict(stage),
        "float32_stage_ms": f32_stage_ms,
        "bit_identical_to_float32_path_on_pcm_over_32768": identical,
        "roofline": {
            "kernel": "stft_covar_kernel<C,false,PCM>", "bound": "hbm",
            "algorithmic_bytes": "U x (2 C N + 4 T F): int16 audio, float32 mask",
            "alg_bytes_per_launch": b_k1,
            "achieved": round(b_k1 / (stage[0] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(b_k1 / (stage[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
            "pass2": {"kernel": "beamform_istft_mc_kernel<C,PCM>", "alg_bytes_per_launch": b_k2,
                      "achieved": round(b_k2 / (stage[2] * 1e-3) / 1e9, 1),
                      "frac": round(b_k2 / (stage[2] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "pipeline_achieved": round(U * (2.0 * C * N + 4.0 * T * F + 4.0 * L) / (ms_enh * 1e-3) / 1e9, 1),
            "note": "the kernels are VALU-issue bound (DESIGN section 5): halving the audio bytes halves this "
                    "fraction's numerator while the time stays; what the variant buys is HBM traffic (power) "
                    "and no float32 copy of the audio",
        },
    }


def add_pcm_counters(leg, pmc, U, C, N, T, L):
    """Counter traffic of the PCM16 step's kernels (same profiled child as the float32 step's)."""
    if not pmc or "error" in pmc:
        return
    rf = leg["roofline"]
    rf["pmc"] = {}
    for key, dst in (("pass1_pcm", rf), ("pass2_pcm", rf["pass2"]), ("ingest", None)):
        p = pmc.get(key) or {}
        if p.get("hbm_read_bytes") is None or p.get("hbm_write_bytes") is None:
            continue
        tr = round(p["hbm_read_bytes"] + p["hbm_write_bytes"])
        rf["pmc"][key] = {"hbm_read_bytes": round(p["hbm_read_bytes"]), "hbm_write_bytes": round(p["hbm_write_bytes"]),
                          "valu_insts": p.get("valu_insts"), "profiled_kernel_ms": p.get("profiled_kernel_ms")}
        if dst is not None:
            dst["traffic"] = tr
            dst["traffic_over_algorithmic"] = round(tr / dst["alg_bytes_per_launch"], 3)


def time_full_batch(args, ctx, opts, torch, audio, masks, C, N, L):
    """All `--full-batch` utterances of configs[2] resident on ONE GPU (19.2 GB):
    the strong-scaling anchor next to the weak-scaling per-GPU shard."""
    n = args.full_batch
    dev = audio[0].device
    nd = len(audio)
    big_a = [audio[i % nd] if i < nd else audio[i % nd].clone() for i in range(n)]
    big_m = [masks[i % nd] if i < nd else masks[i % nd].clone() for i in range(n)]
    big_w = [torch.empty(L, dtype=torch.float32, device=dev) for _ in range(n)]
    ap, mp, wp = ([t.data_ptr() for t in x] for x in (big_a, big_m, big_w))
    ns = [N] * n
    for _ in range(2):
        ctx.enhance_batch(opts, C, ap, ns, mp, None, wp, want_status=False)
    torch.cuda.synchronize()
    k = 5
    t0 = time.perf_counter()
    for _ in range(k):
        ctx.enhance_batch(opts, C, ap, ns, mp, None, wp, want_status=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    return {"utts": n, "ms_per_step": round(1e3 * dt, 3),
            "value": round(n * (N / SR) / dt, 1), "unit": "x real time, one GPU, whole batch"}


def cpu_baseline(args, C, N, first_index, wave0):
    """The oracle (a numpy restatement of the reference path, oracle/np_oracle.py) on ONE host
    core over a bounded sample of the same synthetic workload.  Also the checker of the timed
    configuration: the GPU output of every distinct utterance of the shard against the oracle's
    (the remaining utterances are copies: bit-identical outputs, checked on the device).
    `kind` is "reference" only where the unmodified reference was timed in this run (the build
    container); a Python reference may not travel to the GPU box in any form, so there the
    port is the baseline and `reference_estimate` = port / (port : reference ratio measured
    on the same cores in the build container, profiles/round6_ref_cpu_leg.json)."""
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    import platform
    import scipy
    from oracle import np_oracle as o
    kind = args.beamformer
    n = args.cpu_sample
    utts = []
    for i in range(min(n, 4)):
        mix, sp, nz = o.synth_utterance(first_index + i, C, N, return_parts=True)
        utts.append((mix, o.irm_mask(sp, nz)))

    def run():
        t0 = time.perf_counter()
        for i in range(n):
            mix, mask = utts[i % len(utts)]
            o.enhance_utterance(mix, mask, kind=kind)
        return time.perf_counter() - t0

    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            dt = run()
    else:
        dt = run()
    # parity of the timed configuration (gauge fixed on both sides; the GPU mask is
    # the IRM of the DEVICE spectrograms, the oracle's of its own: < 1e-6 apart)
    parity = None
    if wave0 is not None:
        errs = []
        for i, w in enumerate(wave0["waves"]):
            if i < len(utts):
                mix, mask = utts[i]
            else:
                mix, sp, nz = o.synth_utterance(first_index + i, C, N, return_parts=True)
                mask = o.irm_mask(sp, nz)
            ref = o.enhance_utterance(mix, mask, kind=kind, gauge=True)
            errs.append(float(np.sqrt(np.mean((w - ref)**2)) / np.sqrt(np.mean(ref**2))))
        err = max(errs)
        parity = {"tol": 1e-3, "distinct_utterances_checked": len(errs), "worst_rel_rms_vs_oracle": float(f"{err:.3e}"),
                  "copies": wave0["clones"], "copies_bit_identical_to_their_source": wave0["clones_bit_identical"]}
        if not (err < 1e-3 and wave0["clones_bit_identical"]) and not os.environ.get("SETK_BENCH_NOCHECK"):
            raise SystemExit(f"timed configuration differs from the oracle: worst rel rms {err:.3e}, "
                             f"copies identical: {wave0['clones_bit_identical']}")
    cpu_model = platform.processor()
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = [ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")][0]
    except Exception:
        pass
    one = round(n * (N / SR) / dt, 2)
    out = {
        "value": one, "unit": "x real time (audio seconds per wall second)", "cores": 1, "kind": "port",
        "sample": f"{n} utterances of the same {C}-ch {N / SR:g} s workload, compute only "
                  "(STFT -> covariance -> MVDR -> iSTFT), numpy oracle, 1 thread, "
                  f"{dt:.1f} s wall; host has {os.cpu_count()} cores",
        "host_cpu": cpu_model, "host_logical_cores": os.cpu_count(),
        "numpy": np.__version__, "scipy": scipy.__version__,
        "parity_worst_rel_rms_vs_oracle": None if parity is None else parity["worst_rel_rms_vs_oracle"],
        "parity_check": parity,
    }
    ref = reference_leg(args, C, N)
    if ref.get("present"):
        # the reference itself was timed in this run: it is the baseline, the port a second entry
        ref1 = ref["one_core"]
        out["port_value"] = one
        out.update(value=ref1["value"], kind="reference",
                   sample=f"{ref1['utts']} utterances of the same {C}-ch {N / SR:g} s workload through "
                          "the UNMODIFIED reference CLI (oracle/ref_harness.py), first scp read to "
                          f"last wav close, 1 thread, {ref1['wall_s']} s wall")
    else:
        ratio = ((ref.get("recorded_in_build_container") or {}).get("port_over_reference") or {}).get("one_core")
        if ratio and C == 8 and N == 480000 and kind == "mvdr":
            out["port_over_reference_measured_in_build_container"] = ratio
            out["reference_estimate"] = round(one / ratio, 2)
            allc = ((ref.get("recorded_in_build_container") or {}).get("all_cores") or {})
            if allc.get("value") and (ref["recorded_in_build_container"].get("one_core") or {}).get("value"):
                scale = allc["value"] / ref["recorded_in_build_container"]["one_core"]["value"] / allc["cores"]
                out["reference_estimate_all_cores"] = round(out["reference_estimate"] * scale * (os.cpu_count() or 1), 1)
    out["reference"] = ref
    return out


def reference_leg(args, C, N):
    """the unmodified reference CLI on this host's cores, when the reference
    tree is on this box (the build container); on the GPU box /root/reference does not
    exist, and the record says so and quotes the committed measurement."""
    from oracle import ref_cpu_leg, ref_harness
    if ref_harness.available():
        try:
            rec = ref_cpu_leg.measure(utts=max(4, min(args.cpu_sample, 16)), channels=C,
                                      seconds=N / SR, kind=args.beamformer)
            rec["present"] = True
            return rec
        except Exception as e:  # pragma: no cover
            return {"present": False, "note": f"reference leg failed: {e}"}
    rec = {"present": False, "note": "reference absent on this box (/root/reference is not shipped to the GPU "
                                     "box, and a Python reference may not travel); cpu_baseline.kind stays 'port'"}
    for name in ("round6_ref_cpu_leg.json", "r04_ref_cpu_leg.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec["recorded_in_build_container"] = json.load(f)
            rec["recorded_in"] = "profiles/" + name
            break
        except Exception:
            pass
    return rec


if __name__ == "__main__":
    main()
