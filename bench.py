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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=125, help="utterances per GPU")
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--beamformer", default="mvdr", choices=["mvdr", "gevd", "pmwf-0"])
    ap.add_argument("--distinct", type=int, default=16,
                    help="distinct synthetic utterances generated per rank (others are copies)")
    ap.add_argument("--cpu-sample", type=int, default=96,
                    help="utterances timed on the CPU oracle (0 = skip)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.set_profiling(False)
    stage_ms = ctx.last_stage_ms()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    audio_sec = world * U * (N / SR) * args.steps
    value = audio_sec / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    if rank == 0:
        b_k1 = U * (4.0 * C * N + 4.0 * T * F)            # algorithmic bytes / launch
        k1_ms = stage_ms[0]
        achieved = b_k1 / (k1_ms * 1e-3) / 1e9
        traffic = None
        try:
            with open(args.traffic_json) as f:
                tj = json.load(f)
            if tj.get("utts") == U and tj.get("channels") == C and tj.get("samples") == N:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            pass
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
            "roofline": {
                "kernel": f"stft_covar_kernel<{C}, false>",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "alg_bytes_per_launch": b_k1,
                "kernel_ms": round(k1_ms, 4),
                "pipeline_achieved": round(
                    U * (4.0 * C * N + 4.0 * T * F + 4.0 * L) / (ms_per_step * 1e-3) / 1e9, 1),
            },
        }
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args, C, N)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, C, N):
    """The oracle (a numpy port of the reference path, oracle/np_oracle.py) on
    one host core, over a bounded sample of the same synthetic workload."""
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    from oracle import np_oracle as o
    kind = args.beamformer
    n = args.cpu_sample
    utts = []
    for i in range(min(n, 4)):
        mix, sp, nz = o.synth_utterance(i, C, N, return_parts=True)
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
    return {
        "value": round(n * (N / SR) / dt, 2),
        "unit": "x real time (audio seconds per wall second)",
        "cores": 1,
        "kind": "port",
        "sample": f"{n} utterances of the same {C}-ch {N / SR:g} s workload, compute only "
                  "(STFT -> covariance -> MVDR -> iSTFT), numpy oracle, 1 thread, "
                  f"{dt:.1f} s wall; host has {os.cpu_count()} cores",
    }


if __name__ == "__main__":
    main()
