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
inputs resident in HBM when the timed region starts.  Utterances shard
independently: RCCL carries only the barriers and the max-over-ranks reduction.

Outside the timed K steps (and reported in the same JSON line, N = 1 only unless
noted): `sustained` (every rank keeps stepping for a few seconds so that an
external GPU-busy sampler sees the run), `uncached_call` (the step with new buffer
addresses and a status read-back every call: the timed steps reuse one descriptor
block and skip the read-back), `full_batch` (all 1000 utterances of
configs[2] on the one GPU: the strong-scaling anchor), `cpu_baseline` (one core,
all cores in the reference's process-per-shard mode, and the oracle check of the
timed configuration's output), `other_configs` (configs[1] 4-ch MVDR, configs[3]
8-ch GEV and configs[4] 6-ch CGMM -> MVDR at their BASELINE sizes, each with its
own stage times) and `end_to_end` (disk -> wav through the CLI, plus the host's
RAM copy rate that bounds it).

One JSON line on rank 0:
  value      aggregate real-time factor (audio seconds / wall second, all GPUs)
  roofline   the fused STFT+covariance kernel: algorithmic bytes per launch
             (4*C*N + 4*T*F per utterance) / mean kernel time (HIP
             events on the launch stream, over the timed steps) vs 8 TB/s
  cpu_baseline  the numpy oracle (a port of the reference path) on one host
             core over a bounded sample of the same workload (N=1, rank 0)
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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
SR = 16000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.1 s of warm-up and 0.4 s timed -- with 3 + 20 steps (46 ms in all, rounds 1 - 3)
    # the timed region still sat on the clock / power ramp of a GPU that had been idle: the
    # same build measured 1.83 - 1.86 ms per step there and 1.75 - 1.80 over the 3 s of `sustained`
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--utts", type=int, default=125, help="utterances per GPU")
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--beamformer", default="mvdr", choices=["mvdr", "gevd", "pmwf-0"])
    ap.add_argument("--distinct", type=int, default=16,
                    help="distinct synthetic utterances generated per rank (others are copies)")
    ap.add_argument("--cpu-sample", type=int, default=96,
                    help="utterances timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-allcore-per-proc", type=int, default=3,
                    help="utterances per worker process of the all-core CPU leg (0 = skip)")
    ap.add_argument("--sustain-sec", type=float, default=3.0,
                    help="after the timed steps keep stepping for this long so that an "
                         "external GPU-busy sampler can see the run (N=1 rank 0 reports it)")
    ap.add_argument("--full-batch", type=int, default=1000,
                    help="N=1 only: also time the whole configs[2] batch of this many "
                         "utterances on the one GPU (strong-scaling anchor; 0 = skip)")
    ap.add_argument("--other-configs", type=int, default=1,
                    help="N=1 only: also time BASELINE configs[1], [3] and [4] (short runs) and "
                         "report them in `other_configs` (0 = skip)")
    ap.add_argument("--e2e-utts", type=int, default=192,
                    help="N=1 only: utterances of the end-to-end CLI leg: runs of n and 8 n (0 = skip)")
    ap.add_argument("--pmc", type=int, default=1,
                    help="N=1 only: collect HBM traffic and VALU instruction counts of the two "
                         "streaming kernels IN THIS RUN by re-running three timed steps under "
                         "`rocprofv3 --pmc` (separate passes, counters only; 0 = skip)")
    ap.add_argument("--pmc-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--pcm-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--int16-ingest", type=int, default=1,
                    help="N=1 only: the same shard fed as 16-bit PCM (SETK_FLAG_IN_PCM16: 2 C N audio bytes "
                         "per pass), reported apart in `int16_ingest` with its own roofline block (0 = skip)")
    ap.add_argument("--aux", type=int, default=0,
                    help="N=1 only: the auxiliary legs -- `sustained`, `uncached_call`, `power`, "
                         "`roofline.issue_rates`, `cpu_baseline.all_cores` (default off: the line stays "
                         "short enough for the driver's record to keep stage_ms, roofline.pass1/pass2 and "
                         "cpu_baseline whole)")
    args = ap.parse_args()
    if not args.aux:
        args.sustain_sec = 0.0
        args.cpu_allcore_per_proc = 0
    return args


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
    import torch
    import torch.distributed as dist
    from setk_amd import build as _build
    _build.build_library(force=False)  # no-op when the in-tree library is current
    from setk_amd import _ffi, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # SETK_BENCH_SHARE_GPU=1 + SETK_BENCH_BACKEND=gloo: every rank on cuda:0 with a CPU
    # rendezvous -- only for exercising the multi-rank control flow on a 1-GPU box
    # (tests/test_gpu_api.py); the real launch is one rank per GPU over RCCL.
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
    F = 257

    # ---- synthetic shard, resident in HBM --------------------------------
    audio, masks, waves = [], [], []
    nd = max(1, min(args.distinct, U))
    for i in range(nd):
        mix, sp, nz = synth.synth_utterance(rank * U + i, C, N, return_parts=True)
        a = torch.from_numpy(mix).to(dev)
        parts = torch.from_numpy(np.stack([sp[0], nz[0]])).to(dev)
        spec = torch.empty((2, T, F), dtype=torch.complex64, device=dev)
        ctx.stft(parts, spec)  # oracle (IRM) mask from the device STFT
        s, v = spec[0].abs(), spec[1].abs()
        m = (s / torch.sqrt(s * s + v * v + synth.EPSILON)).contiguous()
        audio.append(a)
        masks.append(m)
    for i in range(nd, U):
        audio.append(audio[i % nd].clone())
        masks.append(masks[i % nd].clone())
    waves = [torch.empty(L, dtype=torch.float32, device=dev) for _ in range(U)]
    aptr = [t.data_ptr() for t in audio]
    mptr = [t.data_ptr() for t in masks]
    wptr = [t.data_ptr() for t in waves]
    ns = [N] * U
    kind = {"mvdr": _ffi.BF_MVDR, "gevd": _ffi.BF_GEVD, "pmwf-0": _ffi.BF_PMWF}[args.beamformer]
    opts = _ffi.BfOpts(kind=kind, flags=_ffi.FLAG_CLAMP_MASK, pmwf_beta=0.0, pmwf_ref=-1, rank1=0)

    def step():
        ctx.enhance_batch(opts, C, aptr, ns, mptr, None, wptr, want_status=False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.pcm_child:
        # profiled child of pcm_traffic(): the int16 step's launches are all it is for
        print(json.dumps(int16_ingest_leg(args, ctx, _ffi, torch, opts, audio, masks, waves, C, N, T, L, U, nd,
                                          None)), flush=True)
        return
    for _ in range(args.warmup):
        step()
    st = ctx.enhance_batch(opts, C, aptr, ns, mptr, None, wptr, want_status=True)
    if any(st) and not os.environ.get("SETK_BENCH_NOCHECK"):
        raise SystemExit(f"numerical status {st}")
    ctx.set_profiling(True)
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

    audio_sec = world * U * (N / SR) * args.steps
    value = audio_sec / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    if args.pmc_child:
        # profiled child of pmc_leg(): the launches above are all it is for
        print(json.dumps({"pmc_child": True, "ms_per_step": round(ms_per_step, 4),
                          "stage_ms": [round(x, 4) for x in stage_ms]}), flush=True)
        return

    # ---- outside the contract's timed region ---------------------------------
    # (a) sustained stepping: the K timed steps last ~40 ms, invisible to a GPU-busy
    #     sampler with a period of seconds
    sustained = None
    if args.sustain_sec > 0:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_sus = 0
        while time.perf_counter() - t0 < args.sustain_sec:
            for _ in range(50):
                step()
            torch.cuda.synchronize()
            n_sus += 50
        sustained = {"steps": n_sus, "ms_per_step": round(1e3 * (time.perf_counter() - t0) / n_sus, 4)}
    # (a') the same step as a caller sees it who does NOT repeat himself: the utterance
    #      tables alternate between two sets of buffers (so the descriptor block is rebuilt
    #      and uploaded every call) and the per-utterance status words are read back
    fresh = None
    if rank == 0 and args.aux:
        audio_b = [t.clone() for t in audio]
        masks_b = [t.clone() for t in masks]
        waves_b = [torch.empty(L, dtype=torch.float32, device=dev) for _ in range(U)]
        sets = [(aptr, mptr, wptr),
                ([t.data_ptr() for t in audio_b], [t.data_ptr() for t in masks_b],
                 [t.data_ptr() for t in waves_b])]
        for i in range(4):
            ctx.enhance_batch(opts, C, sets[i & 1][0], ns, sets[i & 1][1], None, sets[i & 1][2],
                              want_status=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kf = 20
        for i in range(kf):
            ctx.enhance_batch(opts, C, sets[i & 1][0], ns, sets[i & 1][1], None, sets[i & 1][2],
                              want_status=True)
        torch.cuda.synchronize()
        fresh = {"steps": kf, "ms_per_step": round(1e3 * (time.perf_counter() - t0) / kf, 4),
                 "what": "new buffer addresses every call (descriptors rebuilt + uploaded), "
                         "status words read back (one stream synchronisation per call)"}
        del audio_b, masks_b, waves_b
    # one output of the timed configuration, for the oracle check in cpu_baseline()
    # (every DISTINCT utterance of the shard; the others are copies whose outputs must equal
    #  their source's bit for bit -- compared on the device)
    wave0 = None
    if rank == 0:
        wave0 = {"waves": [waves[i].cpu().numpy() for i in range(nd)],
                 "clones_bit_identical": all(bool(torch.equal(waves[i], waves[i % nd])) for i in range(nd, U)),
                 "clones": U - nd}
    # (a'') the shard as 16-bit PCM (its own roofline block; the float32 headline is untouched)
    int16_leg = None
    if rank == 0 and world == 1 and args.int16_ingest:
        try:
            int16_leg = int16_ingest_leg(args, ctx, _ffi, torch, opts, audio, masks, waves, C, N, T, L, U, nd,
                                         {"stft_covar": round(stage_ms[0], 4), "beamform_istft": round(stage_ms[2], 4)})
        except Exception as e:  # an auxiliary leg never takes the headline line down with it
            int16_leg = {"error": repr(e)}
        if args.pmc and "error" not in int16_leg:
            tr = pcm_traffic(args)
            int16_leg["roofline"]["pmc"] = {k: (None if not isinstance(v, dict) else {
                "hbm_read_bytes": v.get("hbm_read_bytes"), "hbm_write_bytes": v.get("hbm_write_bytes"),
                "valu_insts": v.get("valu_insts"), "profiled_kernel_ms": v.get("profiled_kernel_ms")})
                for k, v in tr.items() if k in ("pass1", "pass2", "ingest")} if "error" not in tr else tr
            p1 = tr.get("pass1") if isinstance(tr, dict) else None
            if isinstance(p1, dict) and p1.get("hbm_read_bytes") is not None:
                int16_leg["roofline"]["traffic"] = round(p1["hbm_read_bytes"] + (p1.get("hbm_write_bytes") or 0.0))
    # (b) strong-scaling anchor: the whole configs[2] batch on this one GPU
    full_batch = None
    if world == 1 and args.full_batch > U:
        full_batch = time_full_batch(args, ctx, opts, torch, audio, masks, C, N, L)

    if rank == 0:
        b_k1 = U * (4.0 * C * N + 4.0 * T * F)            # algorithmic bytes / launch, pass 1
        b_k2 = U * (4.0 * C * N + 4.0 * L)                # pass 2: audio again + the wave
        k1_ms, k2_ms = stage_ms[0], stage_ms[2]
        achieved = b_k1 / (k1_ms * 1e-3) / 1e9
        pmc = pmc_leg(args) if (world == 1 and args.pmc) else None
        rates = issue_rates_leg() if (world == 1 and args.pmc and args.aux) else None
        roof = build_roofline(C, b_k1, b_k2, k1_ms, k2_ms, pmc, rates, solve_ms=stage_ms[1])
        if rates is not None:
            roof["issue_rates"] = rates
        out = {
            "metric": "real-time-factor (audio-sec/wall-sec), 8-ch 16 kHz MVDR",
            "value": round(value, 1),
            "unit": "x real time (audio seconds per wall second, all GPUs)",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "per_gpu_value": round(value / world, 1),
            "per_rank_ms_per_step": per_rank_ms,
            "config": {
                "workload": f"{C}-ch 16 kHz oracle-mask {args.beamformer.upper()}, "
                            f"{args.seconds:g} s utterances, {U} per GPU "
                            f"(BASELINE configs[2] shard: {U * 8} at 8 GPUs), "
                            "STFT 512/256/hann/center, inputs resident in HBM",
                "utts_per_gpu": U, "channels": C, "seconds": args.seconds,
                "frames": T, "beamformer": args.beamformer,
                "parallelism": f"utterance-sharded x{world}",
            },
            "stage_ms": {"stft_covar": round(stage_ms[0], 4),
                         "reduce_solve": round(stage_ms[1], 4),
                         "beamform_istft": round(stage_ms[2], 4),
                         "renorm": round(stage_ms[3], 4)},
            "roofline": dict(roof, pipeline_achieved=round(
                U * (4.0 * C * N + 4.0 * T * F + 4.0 * L) / (ms_per_step * 1e-3) / 1e9, 1)),
        }
        if world == 1 and args.cpu_sample > 0:
            # (ahead of the auxiliary legs: the driver's record keeps the head of the line)
            out["cpu_baseline"] = cpu_baseline(args, C, N, rank * U, wave0)
        if int16_leg is not None:
            out["int16_ingest"] = int16_leg
        if sustained is not None:
            out["sustained"] = sustained
        if world == 1 and args.sustain_sec > 0 and args.pmc and args.aux:
            out["power"] = power_leg(step, torch, min(3.0, max(1.0, args.sustain_sec)))
        if fresh is not None:
            out["uncached_call"] = fresh
        if full_batch is not None:
            out["full_batch"] = full_batch
        if world == 1 and args.other_configs:
            del audio, masks, waves
            audio = masks = waves = []
            torch.cuda.empty_cache()
            try:  # (an auxiliary leg never takes the headline line down with it)
                out["other_configs"] = other_configs(torch, _ffi, synth, dev, args if args.pmc else None, rates)
            except Exception as e:
                out["other_configs"] = {"error": repr(e)}
        if world == 1 and args.e2e_utts > 0:
            # free the resident shard first: the CLI leg is its own process
            audio = masks = waves = None
            torch.cuda.empty_cache()
            try:
                out["end_to_end"] = end_to_end(args, C, N)
            except Exception as e:
                out["end_to_end"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


SIMDS = 256 * 4            # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32
XCDS = 8
VALU_CYCLES_PER_INST = 2   # a wave64 VALU instruction occupies its SIMD-32 for two cycles
# measured issue rate of a SIMD shared by n waves (plain fp32 VALU, profiles/r02t_valu_rate_pinned.txt)
# fallback only (profiles/r04f_valu_rate_pinned.txt); the record carries the rates measured in
# THIS run by tools/ubench/valu_rate3 --fma-only (issue_rates_leg)
ISSUE_CYCLES_AT_WAVES = {1: 7.6, 2: 3.6, 3: 2.67, 4: 2.24}
# kernel-name fragments per stage, most specific first; the form that ran is reported
KERNELS = {"pass1": ["stft_covar_mc_kernel", "stft_covar_kernel"],
           "pass2": ["beamform_istft_mc_kernel", "beamform_istft_kernel"],
           "solve": ["solve_kernel"]}
WAVES_PER_SIMD = {"stft_covar_mc_kernel": 4, "stft_covar_kernel": 4,
                  "beamform_istft_mc_kernel": 4, "beamform_istft_kernel": 2, "solve_kernel": 2}
WAVES_WHY = {"stft_covar_kernel": "one 1024-thread workgroup per CU at the 128-VGPR budget",
             "stft_covar_mc_kernel": "one 1024-thread workgroup per CU at the 128-VGPR budget",
             "beamform_istft_mc_kernel": "two 512-thread workgroups per CU at the 128-VGPR budget "
                                         "(54 KB of LDS each: weights + operand tiles)",
             "beamform_istft_kernel": "two 256-thread workgroups per CU: 63.6 KB of LDS each, 256 VGPRs per wave",
             "solve_kernel": "226 VGPRs per wave: two waves per SIMD (32 125 problems x 8 lanes = 4 016 waves, two rounds)"}


def issue_rates_leg():
    """VALU issue rate of a SIMD shared by 1 / 2 / 3 / 4 waves, measured in THIS run with
    pinned instruction streams (tools/ubench/valu_rate3 --fma-only, ~1 s; built on demand)."""
    import re
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(here, "tools", "ubench", "valu_rate3")
    src = exe + ".hip"
    try:
        if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, src],
                           check=True, capture_output=True, timeout=300)
        r = subprocess.run([exe, "--fma-only"], capture_output=True, text=True, timeout=120)
        rates = {}
        for m in re.finditer(r"v_fma_f32\s+waves/SIMD (\d+):.*?= ([\d.]+) per SIMD", r.stdout):
            rates[int(m.group(1))] = float(m.group(2))
        if len(rates) >= 3:
            return {"cycles_per_inst_at_waves": rates, "how": "tools/ubench/valu_rate3 --fma-only in this run"}
        return {"error": "valu_rate3 output not understood: " + r.stdout[-200:] + r.stderr[-200:]}
    except Exception as e:  # noqa: BLE001 - a missing compiler must not take the bench down
        return {"error": f"valu_rate3: {e}"}


def power_leg(step, torch, seconds=2.0):
    """Board power and shader clock while the timed step repeats (rocm-smi sampled from a
    side thread): tells a power cap (clock well under 2.4 GHz at the cap) from a clock the
    kernels simply do not need."""
    import re
    import subprocess
    import threading
    smi = "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return {"error": "rocm-smi not found"}
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                r = subprocess.run([smi, "--showpower", "--showclocks", "-d", "0"], capture_output=True,
                                   text=True, timeout=10)
                pw = re.search(r"Power \(W\):\s*([\d.]+)", r.stdout)
                ck = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", r.stdout)
                samples.append((float(pw.group(1)) if pw else None, int(ck.group(1)) if ck else None))
            except Exception:  # noqa: BLE001
                samples.append((None, None))
    th = threading.Thread(target=sampler, daemon=True)
    t0 = time.perf_counter()
    th.start()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            step()
        torch.cuda.synchronize()
        n += 50
    stop.set()
    th.join(timeout=15)
    pw = [p for p, _ in samples if p is not None]
    ck = [c for _, c in samples if c is not None]
    cap = None
    try:
        r = subprocess.run([smi, "--showmaxpower", "-d", "0"], capture_output=True, text=True, timeout=10)
        m = re.search(r"Max Graphics Package Power \(W\):\s*([\d.]+)", r.stdout)
        cap = float(m.group(1)) if m else None
    except Exception:  # noqa: BLE001
        pass
    return {"steps": n, "samples": len(samples), "power_w_mean": round(sum(pw) / len(pw), 1) if pw else None,
            "power_w_max": max(pw) if pw else None, "power_cap_w": cap,
            "sclk_mhz_mean": round(sum(ck) / len(ck)) if ck else None,
            "how": "rocm-smi --showpower --showclocks polled while the step repeats"}


def pmc_leg(args, child=None, kernels=None, pcm=False):
    """Counters of THIS run's workload (or of `child`, a command line, for the kernels
    `kernels` = {key: [name fragments]}): three timed steps of the same configuration
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
    kernels = kernels or KERNELS
    child = child or [sys.executable, os.path.abspath(__file__), "--pmc-child", "1", "--gpus", "1",
                      "--steps", "3", "--warmup", "1", "--utts", str(args.utts), "--channels", str(args.channels),
                      "--seconds", str(args.seconds), "--beamformer", args.beamformer,
                      "--distinct", str(args.distinct)]
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
                    for key, knames in kernels.items():
                        kname = next((k for k in knames if k + "<" in row["Kernel_Name"]), None)
                        # the streaming kernels' last template argument: 16-bit PCM input
                        is_pcm = ", true>(" in row["Kernel_Name"] or row["Kernel_Name"].rstrip().endswith(", true>")
                        # (stft_covar_kernel<C, true, false> is the spectrogram dump of setk_stft)
                        if kname and ", true, false>" not in row["Kernel_Name"] and \
                                (key not in ("pass1", "pass2") or is_pcm == pcm):
                            acc.setdefault(key, {})["__kernel__"] = kname
                            d = acc[key].setdefault(row["Counter_Name"], [])
                            d.append((float(row["Counter_Value"]),
                                      int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    finally:
        shutil.rmtree(td, ignore_errors=True)
    res = {"method": "in-run: this workload re-run for 3 steps under rocprofv3 --pmc, one pass per "
                     "counter group; read = 2 x FETCH_SIZE KB, write = WRITE_SIZE KB "
                     "(MI355X_MICROARCH.md HBM section); clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time",
           "seconds_spent": round(time.perf_counter() - t0, 1)}
    for key in kernels:
        c = acc.get(key, {})
        mean = lambda name: (sum(v for v, _ in c[name]) / len(c[name])) if c.get(name) else None
        dur = lambda name: (sum(d for _, d in c[name]) / len(c[name])) if c.get(name) else None
        rd, wr = mean("FETCH_SIZE"), mean("WRITE_SIZE")
        insts, gui, dns = mean("SQ_INSTS_VALU"), mean("GRBM_GUI_ACTIVE"), dur("GRBM_GUI_ACTIVE")
        res[key] = {
            "kernel": c.get("__kernel__"),
            "mfma_insts": mean("SQ_INSTS_MFMA"), "mfma_busy_cycles": mean("SQ_VALU_MFMA_BUSY_CYCLES"),
            "mfma_valu_coexec_cycles": mean("SQ_VALU_MFMA_COEXEC_CYCLES"),
            "hbm_read_bytes": None if rd is None else 2.0 * rd * 1024.0,
            "hbm_write_bytes": None if wr is None else wr * 1024.0,
            "valu_insts": insts, "waves": mean("SQ_WAVES"),
            "profiled_kernel_ms": None if dns is None else round(dns / 1e6, 4),
            # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs of the part
            "clock_ghz": None if not (gui and dns) else round(gui / XCDS / dns, 3),
            "launches": len(c.get("SQ_INSTS_VALU", [])),
        }
    return res


def build_roofline(C, b_k1, b_k2, k1_ms, k2_ms, pmc, rates=None, solve_ms=None):
    """The `roofline` object: the HBM numbers of the contract for the STFT+covariance
    kernel, and for BOTH streaming kernels the VALU-issue roofline they actually sit
    under: floor = wave-instructions / 1024 SIMDs x 2 cycles / clock.
    `bound` names the tighter of the two for pass 1."""
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
    for key, alg, kms in (("pass1", b_k1, k1_ms), ("pass2", b_k2, k2_ms), ("solve", None, solve_ms)):
        p = pmc.get(key) or {}
        kname = p.get("kernel") or KERNELS[key][-1]
        if alg is None:
            # the solve: 32 125 small dense problems, no streaming traffic to speak of (0.09 GB);
            # its stage time includes covar_finalize_kernel, the floor is priced on the kernel's
            # own (profiled) duration
            if not p.get("valu_insts"):
                continue
            kms = kms if kms else p.get("profiled_kernel_ms")
            ent = {"kernel": kname, "stage_ms_with_partial_reduce": None if solve_ms is None else round(solve_ms, 4)}
        else:
            ent = {"kernel": kname, "kernel_ms": round(kms, 4), "alg_bytes_per_launch": alg,
                   "hbm": {"achieved": round(alg / (kms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                           "frac": round(alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
        if alg is not None and p.get("hbm_read_bytes") is not None and p.get("hbm_write_bytes") is not None:
            ent["hbm"]["traffic"] = round(p["hbm_read_bytes"] + p["hbm_write_bytes"])
            ent["hbm"]["traffic_over_algorithmic"] = round(ent["hbm"]["traffic"] / alg, 3)
        if p.get("valu_insts") and p.get("clock_ghz"):
            floor_ms = p["valu_insts"] / SIMDS * VALU_CYCLES_PER_INST / (p["clock_ghz"] * 1e9) * 1e3
            ent["valu_issue"] = {"insts": round(p["valu_insts"]), "simds": SIMDS,
                                 "cycles_per_inst": VALU_CYCLES_PER_INST,
                                 "clock_ghz": p["clock_ghz"], "floor_ms": round(floor_ms, 4),
                                 "profiled_kernel_ms": p["profiled_kernel_ms"],
                                 "frac": round(floor_ms / p["profiled_kernel_ms"], 4)
                                 if p.get("profiled_kernel_ms") else None,
                                 "frac_unprofiled": round(floor_ms / kms, 4)}
            # what the kernel's OCCUPANCY lets a SIMD issue: measured with pinned instruction
            # streams (tools/ubench/valu_rate2.hip, profiles/r02t_valu_rate_pinned.txt): one
            # plain fp32 VALU instruction per 7.6 / 3.6 / 2.24 cycles with 1 / 2 / 4 waves
            waves = WAVES_PER_SIMD[kname]
            measured = (rates or {}).get("cycles_per_inst_at_waves") or {}
            cpi = measured.get(waves, ISSUE_CYCLES_AT_WAVES[waves])
            occ_floor = floor_ms * cpi / VALU_CYCLES_PER_INST
            ent["valu_issue"]["at_occupancy"] = {
                "waves_per_simd": waves, "cycles_per_inst": cpi,
                "cycles_per_inst_source": "measured in this run" if waves in measured else "profiles/r04f_valu_rate_pinned.txt",
                "floor_ms": round(occ_floor, 4),
                "frac": round(occ_floor / p["profiled_kernel_ms"], 4) if p.get("profiled_kernel_ms") else None,
                "why": WAVES_WHY[kname]}
            if p.get("mfma_insts"):
                # the matrix pipe of a SIMD is busy SQ_VALU_MFMA_BUSY_CYCLES / 1024 cycles; what of
                # it overlaps VALU issue is SQ_VALU_MFMA_COEXEC_CYCLES
                mf_ms = p["mfma_busy_cycles"] / SIMDS / (p["clock_ghz"] * 1e9) * 1e3
                co_ms = (p.get("mfma_valu_coexec_cycles") or 0.0) / SIMDS / (p["clock_ghz"] * 1e9) * 1e3
                ent["mfma"] = {"insts": round(p["mfma_insts"]), "op": "v_mfma_f32_16x16x32_f16",
                               "busy_ms_per_simd": round(mf_ms, 4), "coexec_with_valu_ms": round(co_ms, 4),
                               "valu_plus_mfma_floor_ms": round(occ_floor + mf_ms - co_ms, 4),
                               "frac_of_that_floor": round((occ_floor + mf_ms - co_ms) / p["profiled_kernel_ms"], 4)
                               if p.get("profiled_kernel_ms") else None}
        roof[key] = ent
    p1 = roof.get("pass1", {})
    if "traffic" in p1.get("hbm", {}):
        roof["traffic"] = p1["hbm"]["traffic"]
    vi = p1.get("valu_issue")
    if vi and vi.get("frac") and vi["frac"] > roof["frac"]:
        # closer to its VALU-issue ceiling than to the HBM ceiling: that is the binding one
        roof["bound"] = "valu_issue"
        roof["bound_note"] = ("`achieved`/`peak`/`frac` stay the contract's HBM figures; the kernel "
                              "is nearer its VALU-issue floor (pass1.valu_issue.frac)")
    return roof


def other_configs(torch, _ffi, synth, dev, pmc_args=None, rates=None):
    """The other GPU configurations BASELINE.json names, timed briefly (inputs resident
    in HBM, 10 steps each) so that one record carries all of them:
    configs[1] 4-ch 10 s MVDR (500 utterances), configs[3] 8-ch 30 s GEV (125),
    configs[4] 6-ch 30 s CGMM (20 EM iterations) -> MVDR (125)."""
    from setk_amd.engine import CgmmEstimator
    F = 257
    res = {}

    def shard(ctx, C, N, U, nd=8):
        T = ctx.num_frames(N)
        L = ctx.istft_num_samples(T)
        audio, masks = [], []
        for i in range(nd):
            mix, sp, nz = synth.synth_utterance(1000 + i, C, N, return_parts=True)
            a = torch.from_numpy(mix).to(dev)
            parts = torch.from_numpy(np.stack([sp[0], nz[0]])).to(dev)
            spec = torch.empty((2, T, F), dtype=torch.complex64, device=dev)
            ctx.stft(parts, spec)
            sa, va = spec[0].abs(), spec[1].abs()
            audio.append(a)
            masks.append((sa / torch.sqrt(sa * sa + va * va + synth.EPSILON)).contiguous())
        for i in range(nd, U):
            audio.append(audio[i % nd].clone())
            masks.append(masks[i % nd].clone())
        waves = [torch.empty(L, dtype=torch.float32, device=dev) for _ in range(U)]
        return audio, masks, waves, T, L

    def run(label, C, seconds, U, kind):
        ctx = _ffi.Context(dev.index)
        ctx.stft_plan(512, 256, 512, True)
        N = int(round(seconds * SR))
        audio, masks, waves, T, L = shard(ctx, C, N, U)
        ap, mp, wp = ([t.data_ptr() for t in x] for x in (audio, masks, waves))
        ns = [N] * U
        opts = _ffi.BfOpts(kind=kind, flags=_ffi.FLAG_CLAMP_MASK, pmwf_beta=0.0, pmwf_ref=-1, rank1=0)
        for _ in range(40):   # (steady clocks, as the headline's warm-up)
            ctx.enhance_batch(opts, C, ap, ns, mp, None, wp, want_status=False)
        st = ctx.enhance_batch(opts, C, ap, ns, mp, None, wp, want_status=True)
        ctx.set_profiling(True)
        torch.cuda.synchronize()
        k = 100
        t0 = time.perf_counter()
        for _ in range(k):
            ctx.enhance_batch(opts, C, ap, ns, mp, None, wp, want_status=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
        ctx.set_profiling(False)
        sm = ctx.last_stage_ms()
        b_k1 = U * (4.0 * C * N + 4.0 * T * F)
        res[label] = {
            "workload": f"{C}-ch {seconds:g} s x {U} utterances, "
                        f"{'GEV' if kind == _ffi.BF_GEVD else 'MVDR'}, inputs resident in HBM",
            "ms_per_step": round(1e3 * dt, 4), "value": round(U * seconds / dt, 1),
            "status_ok": not any(st),
            "stage_ms": {"stft_covar": round(sm[0], 4), "reduce_solve": round(sm[1], 4),
                         "beamform_istft": round(sm[2], 4), "renorm": round(sm[3], 4)},
            "stft_covar_roofline_frac": round(b_k1 / (sm[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        }
        ctx.close()
        del audio, masks, waves
        torch.cuda.empty_cache()

    run("configs[1] 4-ch MVDR", 4, 10.0, 500, _ffi.BF_MVDR)
    run("configs[3] 8-ch GEV", 8, 30.0, 125, _ffi.BF_GEVD)
    # configs[4]: CGMM mask estimation feeding MVDR
    ctx = _ffi.Context(dev.index)
    est = CgmmEstimator(num_iters=20, ctx=ctx)
    est._plan()
    C, N, U = 6, 30 * SR, 125
    audio = [torch.from_numpy(synth.synth_utterance(2000 + (i % 8), C, N)).to(dev) for i in range(8)]
    audio += [audio[i % 8].clone() for i in range(8, U)]
    L = ctx.istft_num_samples(ctx.num_frames(N))
    waves = [torch.empty(L, dtype=torch.float32, device=dev) for _ in range(U)]
    opts = _ffi.BfOpts(kind=_ffi.BF_MVDR, flags=_ffi.FLAG_CLAMP_MASK, pmwf_ref=-1)

    def cg_step():
        m = est.estimate_device(audio)
        ctx.enhance_batch(opts, C, [t.data_ptr() for t in audio], [N] * U,
                          [t.data_ptr() for t in m], None, [w.data_ptr() for w in waves],
                          want_status=False)
        torch.cuda.synchronize()

    cg_step()
    cg_step()
    t0 = time.perf_counter()
    for _ in range(6):
        cg_step()
    dt = (time.perf_counter() - t0) / 6
    res["configs[4] 6-ch CGMM->MVDR"] = {
        "workload": "6-ch 30 s x 125 utterances, CGMM (K = 2, 20 EM iterations) -> MVDR, "
                    "inputs resident in HBM",
        "ms_per_step": round(1e3 * dt, 3), "value": round(U * 30.0 / dt, 1)}
    T4 = ctx.num_frames(N)
    ctx.close()
    del audio, waves
    torch.cuda.empty_cache()
    if pmc_args is not None:
        res["configs[4] 6-ch CGMM->MVDR"]["roofline"] = cgmm_roofline(pmc_args, C, T4, U, rates)
    # the paths around the fused hot path (SURVEY 8f-4 consumers, unfused engine)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_consumers
        res["consumers_and_unfused"] = bench_consumers.run()
    except Exception as e:  # pragma: no cover
        res["consumers_and_unfused"] = {"error": repr(e)[:300]}
    return res


def cgmm_roofline(args, C, T, U, rates):
    """The EM kernel of configs[4] (cgmm_bin_em_kernel: all iterations of one (utterance, bin)
    in one workgroup) under the same two ceilings as the streaming kernels, from counters
    collected in this run: tools/bench_cgmm.py at the configs[4] shape as a child of
    `rocprofv3 --pmc`.  Algorithmic bytes: the bin-major spectrogram read once + the masks
    written once."""
    child = [sys.executable, os.path.join(ROOT, "tools", "bench_cgmm.py"), "--utts", str(U),
             "--channels", str(C), "--seconds", "30", "--iters", "20", "--steps", "1"]
    pmc = pmc_leg(args, child=child, kernels={"em": ["cgmm_bin_em_kernel"]})
    if "error" in pmc or not (pmc.get("em") or {}).get("valu_insts"):
        return {"pmc": pmc}
    p = pmc["em"]
    F = 257
    alg = U * (8.0 * C * T * F + 4.0 * T * F)
    kms = p["profiled_kernel_ms"]
    ent = {"kernel": "cgmm_bin_em_kernel", "profiled_kernel_ms": kms, "launches_profiled": p["launches"],
           "alg_bytes_per_launch": alg, "seconds_spent": pmc.get("seconds_spent"),
           "hbm": {"achieved": round(alg / (kms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                   "frac": round(alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
    if p.get("hbm_read_bytes") is not None and p.get("hbm_write_bytes") is not None:
        ent["hbm"].update(read_bytes=round(p["hbm_read_bytes"]), write_bytes=round(p["hbm_write_bytes"]),
                          traffic_over_algorithmic=round((p["hbm_read_bytes"] + p["hbm_write_bytes"]) / alg, 3))
    floor_ms = p["valu_insts"] / SIMDS * VALU_CYCLES_PER_INST / (p["clock_ghz"] * 1e9) * 1e3
    measured = (rates or {}).get("cycles_per_inst_at_waves") or {}
    cpi = measured.get(3, ISSUE_CYCLES_AT_WAVES[3])
    ent["valu_issue"] = {"insts": round(p["valu_insts"]), "clock_ghz": p["clock_ghz"],
                         "floor_ms": round(floor_ms, 3), "frac": round(floor_ms / kms, 4),
                         "at_occupancy": {"waves_per_simd": 3, "cycles_per_inst": cpi,
                                          "floor_ms": round(floor_ms * cpi / VALU_CYCLES_PER_INST, 3),
                                          "frac": round(floor_ms * cpi / VALU_CYCLES_PER_INST / kms, 4),
                                          "why": "three 256-thread workgroups per CU: 46.5 KB of LDS each, 168 VGPRs"}}
    ent["bound"] = "valu_issue"
    return ent


def int16_ingest_leg(args, ctx, _ffi, torch, opts, audio, masks, waves, C, N, T, L, U, nd, f32_stage_ms):
    """The timed configuration with its audio as 16-bit PCM, as a wave file stores it (SURVEY
    8f-3 "int16 ingest on device", 8d "report that variant separately with 2 C N").  Resident
    in HBM: the interleaved frames [N][C] of every utterance.  One step = de-interleave into
    planar int16 (setk_pcm16_deinterleave_batch: 4 C N bytes; the float32 conversion it
    replaces moved 6 C N) + the four stages with SETK_FLAG_IN_PCM16 (both streaming kernels
    read 2 bytes per sample; read_wav's / 32768 is folded into their window tables).
    `enhance_only`: the planar samples already resident (what a caller who stores int16
    [C][N] pays).  The parity check is exact: the outputs equal the float32 path's on
    pcm / 32768 bit for bit."""
    dev = audio[0].device
    F = 257
    stride = ctx.pcm16_channel_stride(N)
    frames, f32q = [], []
    for i in range(nd):
        q = torch.clamp(torch.round(audio[i].T * 32767.0), -32768, 32767).to(torch.int16).contiguous()  # [N][C]
        frames.append(q)
        f32q.append((q.T.to(torch.float32) / 32768.0).contiguous())
    frames += [frames[i % nd].clone() for i in range(nd, U)]
    planar = [torch.empty((C, stride), dtype=torch.int16, device=dev) for _ in range(U)]
    fptr = [t.data_ptr() for t in frames]
    pptr = [t.data_ptr() for t in planar]
    mptr = [t.data_ptr() for t in masks]
    wptr = [t.data_ptr() for t in waves]
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

    warm, k = (1, 3) if args.pcm_child else (max(10, args.warmup // 2), max(20, args.steps // 2))
    step()
    st = ctx.enhance_batch(po, C, pptr, ns, mptr, None, wptr, want_status=True)
    ms_full = timed(step, warm, k)
    ctx.set_profiling(True)
    ms_enh = timed(lambda: step(False), warm, k)
    ctx.set_profiling(False)
    stage = ctx.last_stage_ms()
    if args.pcm_child:
        return {"pcm_child": True, "ms_per_step": round(ms_full, 4), "stage_ms": [round(x, 4) for x in stage]}
    # ingest kernel alone
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
    b_all = U * (2.0 * C * N + 4.0 * T * F + 4.0 * L)
    out = {
        "what": f"{U} x {C}-ch x {N / SR:g} s, audio resident as the wave files' interleaved 16-bit frames; "
                "step = de-interleave (planar int16, no float32 copy) + the four stages reading int16",
        "status": "ok" if st is not None and not any(st) else str(st),
        "ms_per_step": round(ms_full, 4),
        "value": round(U * (N / SR) / (ms_full * 1e-3), 1),
        "enhance_only_ms": round(ms_enh, 4),
        "ingest_ms": round(ms_ingest, 4),
        "ingest_gbs": round(U * 4.0 * C * N / (ms_ingest * 1e-3) / 1e9, 1),
        "stage_ms": {"stft_covar": round(stage[0], 4), "reduce_solve": round(stage[1], 4),
                     "beamform_istft": round(stage[2], 4), "renorm": round(stage[3], 4)},
        "float32_stage_ms": f32_stage_ms,
        "bit_identical_to_float32_path_on_pcm_over_32768": identical,
        "roofline": {
            "kernel": "stft_covar_kernel<C,false,PCM>", "bound": "hbm",
            "algorithmic_bytes": "U x (2 C N + 4 T F): int16 audio, float32 mask",
            "achieved": round(b_k1 / (stage[0] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(b_k1 / (stage[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
            "pass2": {"kernel": "beamform_istft_mc_kernel<C,PCM>",
                      "achieved": round(b_k2 / (stage[2] * 1e-3) / 1e9, 1),
                      "frac": round(b_k2 / (stage[2] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "pipeline_achieved": round(b_all / (ms_enh * 1e-3) / 1e9, 1),
            "note": "the kernels are VALU-issue bound (section 5 of DESIGN.md): halving the audio bytes "
                    "halves this fraction's numerator while the time stays; what the variant buys is HBM "
                    "traffic (power) and the ingest pass",
        },
    }
    return out


def pcm_traffic(args):
    """HBM traffic of the int16 step's kernels: a child of this script under rocprofv3 --pmc
    (counters only), as pmc_leg does for the float32 step."""
    child = [sys.executable, os.path.abspath(__file__), "--pcm-child", "1", "--gpus", "1",
             "--steps", "3", "--warmup", "1", "--utts", str(args.utts), "--channels", str(args.channels),
             "--seconds", str(args.seconds), "--beamformer", args.beamformer, "--distinct", str(args.distinct)]
    try:
        return pmc_leg(args, child=child, pcm=True, kernels={
            "pass1": ["stft_covar_kernel"], "pass2": ["beamform_istft_mc_kernel"],
            "ingest": ["pcm16_deinterleave_batch_kernel"]})
    except Exception as e:  # the leg is auxiliary: report, do not fail the bench
        return {"error": repr(e)}


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


_ALLCORE_WORKER = r"""
import os, sys, time, json
os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = "1"
sys.path.insert(0, sys.argv[1])
idx, n, C, N, kind = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
from oracle import np_oracle as o
mix, sp, nz = o.synth_utterance(idx % 4, C, N, return_parts=True)
mask = o.irm_mask(sp, nz)
o.enhance_utterance(mix, mask, kind=kind)   # warm up caches / imports
ready = time.time()
while time.time() < float(sys.argv[7]):     # common start line
    time.sleep(0.005)
t0 = time.time()
for _ in range(n):
    o.enhance_utterance(mix, mask, kind=kind)
print(json.dumps(dict(t0=t0, t1=time.time(), ready=ready)))
"""


def cpu_allcore(args, C, N):
    """The reference's own parallel mode on the host: nj single-threaded processes
    over disjoint shards (scripts/run_adapt_beamformer.sh:69-92, run.pl JOB=1:nj),
    here nj = the host's cores (bounded by free memory), each running the oracle."""
    import subprocess
    nj = os.cpu_count() or 1
    try:
        with open("/proc/meminfo") as f:
            avail_kb = [int(l.split()[1]) for l in f if l.startswith("MemAvailable")][0]
        nj = max(1, min(nj, int(avail_kb / 1024 / 1024 * 0.5 / 0.6)))  # ~0.6 GB per worker
    except Exception:
        pass
    per = args.cpu_allcore_per_proc
    start_at = time.time() + 40.0   # workers import numpy/scipy and synthesise first
    procs = [subprocess.Popen([sys.executable, "-c", _ALLCORE_WORKER, ROOT, str(i), str(per), str(C),
                               str(N), args.beamformer, repr(start_at)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for i in range(nj)]
    res = []
    for p in procs:
        o_, e_ = p.communicate(timeout=900)
        if p.returncode == 0 and o_.strip():
            res.append(json.loads(o_.strip().splitlines()[-1]))
    if not res:
        return None
    late = sum(1 for r in res if r["ready"] > start_at)
    wall = max(r["t1"] for r in res) - min(r["t0"] for r in res)
    n_utts = len(res) * per
    return {"value": round(n_utts * (N / SR) / wall, 1), "cores": len(res),
            "wall_s": round(wall, 2), "utts": n_utts, "late_workers": late}


def cpu_baseline(args, C, N, first_index, wave0):
    """The oracle (a numpy port of the reference path, oracle/np_oracle.py) on the
    host: one core over a bounded sample of the same synthetic workload, then all
    cores in the reference's process-per-shard mode.  Also the checker of the
    timed configuration: the GPU output of every distinct utterance of the shard against the
    oracle's (the remaining utterances are copies: bit-identical outputs, checked on the device)."""
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
        parity = {"utterance": first_index, "rel_rms_vs_oracle": float(f"{errs[0]:.3e}"), "tol": 1e-3,
                  "distinct_utterances_checked": len(errs), "worst_rel_rms_vs_oracle": float(f"{err:.3e}"),
                  "copies": wave0["clones"], "copies_bit_identical_to_their_source": wave0["clones_bit_identical"]}
        if not (err < 1e-3 and wave0["clones_bit_identical"]) and not os.environ.get("SETK_BENCH_NOCHECK"):
            raise SystemExit(f"timed configuration differs from the oracle: worst rel rms {err:.3e}, "
                             f"copies identical: {wave0['clones_bit_identical']}")
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")][0]
    except Exception:
        cpu_model = platform.processor()
    one = round(n * (N / SR) / dt, 2)
    out = {
        "value": one,
        "unit": "x real time (audio seconds per wall second)",
        "cores": 1,
        "kind": "port",
        "sample": f"{n} utterances of the same {C}-ch {N / SR:g} s workload, compute only "
                  "(STFT -> covariance -> MVDR -> iSTFT), numpy oracle, 1 thread, "
                  f"{dt:.1f} s wall; host has {os.cpu_count()} cores",
        "host": {"cpu": cpu_model, "logical_cores": os.cpu_count(), "numpy": np.__version__,
                 "scipy": scipy.__version__},
        # SURVEY 6 [probe]: the unmodified reference CLI needs 0.59 s per 8-ch/30-s
        # utterance on one 2.1 GHz Xeon core (51 x real time); the port skips its two
        # extra wav decodes and np.stack copy
        "ratio_to_reference_probe": round(one / (30.0 / 0.59), 2) if C == 8 and N == 480000 else None,
        "parity_check": parity,
    }
    out["reference"] = reference_leg(args, C, N)
    if out["reference"].get("present"):
        # the reference itself was timed in this run: it is the baseline, the port a second entry
        ref1 = out["reference"]["one_core"]
        out["port"] = {"value": out["value"], "cores": 1, "sample": out["sample"]}
        out.update(value=ref1["value"], kind="reference",
                   sample=f"{ref1['utts']} utterances of the same {C}-ch {N / SR:g} s workload through "
                          "the UNMODIFIED reference CLI (oracle/ref_harness.py), first scp read to "
                          f"last wav close, 1 thread, {ref1['wall_s']} s wall")
    if args.cpu_allcore_per_proc > 0:
        allc = cpu_allcore(args, C, N)
        if allc:
            allc["unit"] = out["unit"]
            allc["sample"] = (f"{allc['cores']} single-threaded oracle processes x "
                              f"{args.cpu_allcore_per_proc} utterances each (run.pl JOB=1:nj style), "
                              "compute only, common start line")
            out["all_cores"] = allc
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
    rec = {"present": False, "note": "reference absent on this box (/root/reference is not shipped "
                                     "to the GPU box); cpu_baseline.kind stays 'port'"}
    try:
        with open(os.path.join(ROOT, "profiles", "r04_ref_cpu_leg.json")) as f:
            rec["recorded_in_build_container"] = json.load(f)
    except Exception:
        pass
    return rec


def host_copy_rate(threads=(1, 8), nbytes=64 << 20, reps=4):
    """RAM -> RAM copy rate of this host (numpy, GIL released), per thread count: the
    ceiling of any path that stages file bytes through a page-locked buffer."""
    import threading
    out = {}
    for nt in threads:
        src = [np.ones(nbytes, dtype=np.uint8) for _ in range(nt)]
        dst = [np.ones(nbytes, dtype=np.uint8) for _ in range(nt)]   # ones: pages touched

        def work(k):
            for _ in range(reps):
                np.copyto(dst[k], src[k])
        th = [threading.Thread(target=work, args=(k,)) for k in range(nt)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        out[str(nt)] = round(nt * reps * nbytes / (time.perf_counter() - t0) / 1e9, 2)
    return out


def end_to_end(args, C, N):
    """disk -> wav through the drop-in CLI (scripts/sptk/apply_adaptive_beamformer.py),
    PCM16 wav + numpy masks in, PCM16 wav out, on files written to /dev/shm (or
    TMPDIR).  Two runs (n and 8n utterances) give the marginal cost per utterance;
    each run reports two wall clocks: the whole process (python + torch import +
    plan + pinned pools) and the CLI's own clock from its first scp read to the
    last wav close."""
    import shutil
    import subprocess
    import tempfile
    from setk_amd import synth
    from setk_amd.libs import wavio
    n1 = args.e2e_utts
    # (8 x: the difference of two process clocks carries ~0.1 s of start-up noise; at 4 x the
    #  marginal rate came out anywhere between 21 and 43 GB/s on the same build)
    n2 = 8 * n1
    T = 1 + N // 256
    need = n2 * (2 * C * N + 4 * T * 257 + 2 * N) * 1.1
    base = None
    for cand in ("/dev/shm", os.environ.get("TMPDIR", "/tmp")):
        try:
            if os.path.isdir(cand) and os.access(cand, os.W_OK) and \
                    shutil.disk_usage(cand).free > need:
                base = cand
                break
        except OSError:
            pass
    if base is None:
        return {"error": "no scratch directory with %.1f GB free" % (need / 1e9)}
    d = tempfile.mkdtemp(prefix="setk_e2e_", dir=base)
    try:
        rng = np.random.default_rng(0)
        os.makedirs(f"{d}/wav")
        os.makedirs(f"{d}/mask")
        nd = 4
        for i in range(nd):
            mix = synth.synth_utterance(i, C, N)
            wavio.write_pcm16(f"{d}/wav/u{i}.wav", wavio.float_to_pcm16(mix.T), SR)
            np.save(f"{d}/mask/u{i}.npy", rng.uniform(0.05, 0.95, size=(T, 257)).astype(np.float32))
        for i in range(nd, n2):
            shutil.copyfile(f"{d}/wav/u{i % nd}.wav", f"{d}/wav/u{i}.wav")
            shutil.copyfile(f"{d}/mask/u{i % nd}.npy", f"{d}/mask/u{i}.npy")

        def run_cli(n):
            with open(f"{d}/wav.scp", "w") as ws, open(f"{d}/mask.scp", "w") as ms:
                for i in range(n):
                    ws.write(f"u{i} {d}/wav/u{i}.wav\n")
                    ms.write(f"u{i} {d}/mask/u{i}.npy\n")
            shutil.rmtree(f"{d}/enh", ignore_errors=True)
            cmd = [sys.executable,
                   os.path.join(ROOT, "scripts", "sptk", "apply_adaptive_beamformer.py"),
                   "--mask-format", "numpy", "--beamformer", args.beamformer,
                   "--profile", f"{d}/prof.json", f"{d}/wav.scp", f"{d}/mask.scp", f"{d}/enh"]
            t0 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)
            wall = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": r.stderr[-500:]}
            done = len([f for f in os.listdir(f"{d}/enh") if f.endswith(".wav")])
            prof = {}
            try:
                with open(f"{d}/prof.json") as f:
                    prof = json.load(f)
            except Exception:
                pass
            inner = prof.get("wall_s")
            st = prof.get("stages") or {}
            return {"utts": n, "written": done, "audio_s": n * N / SR,
                    "wall_s_process": round(wall, 3),
                    "value_process": round(n * N / SR / wall, 1),
                    "wall_s_first_read_to_last_write": None if inner is None else round(inner, 3),
                    "value_first_read_to_last_write":
                        None if not inner else round(n * N / SR / inner, 1),
                    "pipeline_wall_s": None if "wall_s" not in st else round(st["wall_s"], 3),
                    "stages": {k: (round(v, 4) if isinstance(v, float) else v)
                               for k, v in st.items()}}

        # each size twice, the faster run counts: the first read of freshly written page-cache
        # pages varies by 2x between otherwise identical runs
        def best(n):
            runs = [run_cli(n) for _ in range(2)]
            good = [r for r in runs if "error" not in r]
            if not good:
                return runs[0]
            r = min(good, key=lambda r_: r_["wall_s_process"])
            r["wall_s_process_all"] = [r_["wall_s_process"] for r_ in good]
            return r
        r1, r2 = best(n1), best(n2)
        out = {"workload": f"{C}-ch {N / SR:g} s PCM16 wav + float32 numpy masks on {base}, "
                           f"{args.beamformer}, PCM16 wav out, through "
                           "scripts/sptk/apply_adaptive_beamformer.py",
               "unit": "x real time (audio seconds per wall second)",
               "runs": [r1, r2]}
        if "error" not in r1 and "error" not in r2:
            dm = (r2["wall_s_process"] - r1["wall_s_process"]) / (n2 - n1)
            out["marginal_ms_per_utt"] = round(1e3 * dm, 4)
            out["marginal_value"] = round((N / SR) / dm, 1) if dm > 0 else None
            out["marginal_GBps_in"] = round((2 * C * N + 4 * T * 257) / dm / 1e9, 2) if dm > 0 else None
            # the same from the CLI's own clocks (no interpreter start-up noise in the difference)
            for name, key in (("marginal_ms_per_utt_first_read_to_last_write",
                               "wall_s_first_read_to_last_write"),
                              ("marginal_ms_per_utt_pipeline", "pipeline_wall_s")):
                if r1.get(key) is not None and r2.get(key) is not None:
                    out[name] = round(1e3 * (r2[key] - r1[key]) / (n2 - n1), 4)
        # what the host can copy at all (threads -> GB/s): the input bytes are copied once
        # from the page cache into page-locked slabs before the DMA
        out["host_copy_GBps"] = host_copy_rate()
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
