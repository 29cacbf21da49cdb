#!/usr/bin/env python
"""
The auxiliary legs of bench.py (`python bench.py --aux 1`, N = 1 only): everything that is not the
contract's timed steps, their counters, the CPU baseline and the whole-batch anchor.  Kept apart so
that the driver's command stays short; every leg reports into the same JSON line.

  sustained / uncached_call / power / roofline.issue_rates   the step repeated for seconds, with new
        buffer addresses every call, under a rocm-smi sampler, and the SIMD issue rates of this part
  cpu_baseline.all_cores   the oracle in the reference's process-per-shard mode on every host core
  other_configs   BASELINE configs[1], [3], [4] at size (+ the EM kernel's counters, the consumers)
  end_to_end      disk -> wav through the drop-in CLI, three repeats per size: median, min, max
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
SR = 16000
HBM_PEAK_GBS = 8000.0
SIMDS = 256 * 4
VALU_CYCLES_PER_INST = 2
ISSUE_CYCLES_AT_WAVES = {1: 7.6, 2: 3.6, 3: 2.67, 4: 2.24}   # profiles/r04f_valu_rate_pinned.txt

def issue_rates_leg():
    """VALU issue rate of a SIMD shared by 1 / 2 / 3 / 4 waves, measured in THIS run with
    pinned instruction streams (tools/ubench/valu_rate3 --fma-only, ~1 s; built on demand)."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "ubench", "valu_rate3")
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


def other_configs(torch, _ffi, synth, dev, pmc_args=None, rates=None):
    """The other GPU configurations BASELINE.json names, timed briefly (inputs resident
    in HBM, 10 steps each) so that one record carries all of them:
    configs[1] 4-ch 10 s MVDR (500 utterances), configs[3] 8-ch 30 s GEV (125),
    configs[4] 6-ch 30 s CGMM (20 EM iterations) -> MVDR (125)."""
    from setk_amd.engine import CgmmEstimator
    F = 257
    res = {}
    t_leg = time.perf_counter()

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
    from bench import pmc_leg
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
    ent["note"] = ("since round 6 the frames phases issue packed fp32 (v_pk_*_f32): about two thirds of the count are "
                   "instructions that hold a SIMD's pipe for FOUR cycles, so this 2-cycle floor undercounts the pipe time "
                   "(the plain form of the same kernel: 8.95e9 instructions, frac 0.55)")
    return ent


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


def stepping_legs(args, ctx, _ffi, torch, opts, step, audio, masks, C, N, L, U):
    """`sustained` (the K timed steps last ~40 ms, invisible to a GPU-busy sampler with a period of
    seconds), `power` (board power and shader clock meanwhile) and `uncached_call` (the step as a
    caller sees it who does NOT repeat himself: the utterance tables alternate between two sets of
    buffers, so the descriptor block is rebuilt and uploaded every call, and the per-utterance
    status words are read back)."""
    out = {}
    dev = audio[0].device
    if args.sustain_sec > 0:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_sus = 0
        while time.perf_counter() - t0 < args.sustain_sec:
            for _ in range(50):
                step()
            torch.cuda.synchronize()
            n_sus += 50
        out["sustained"] = {"steps": n_sus, "ms_per_step": round(1e3 * (time.perf_counter() - t0) / n_sus, 4)}
        out["power"] = power_leg(step, torch, min(3.0, max(1.0, args.sustain_sec)))
    ns = [N] * U
    audio_b = [t.clone() for t in audio]
    masks_b = [t.clone() for t in masks]
    sets = []
    for a_, m_ in ((audio, masks), (audio_b, masks_b)):
        w_ = [torch.empty(L, dtype=torch.float32, device=dev) for _ in range(U)]
        sets.append(([t.data_ptr() for t in a_], [t.data_ptr() for t in m_], [t.data_ptr() for t in w_], w_))
    for i in range(4):
        ctx.enhance_batch(opts, C, sets[i & 1][0], ns, sets[i & 1][1], None, sets[i & 1][2], want_status=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kf = 20
    for i in range(kf):
        ctx.enhance_batch(opts, C, sets[i & 1][0], ns, sets[i & 1][1], None, sets[i & 1][2], want_status=True)
    torch.cuda.synchronize()
    out["uncached_call"] = {"steps": kf, "ms_per_step": round(1e3 * (time.perf_counter() - t0) / kf, 4),
                            "what": "new buffer addresses every call (descriptors rebuilt + uploaded), "
                                    "status words read back (one stream synchronisation per call)"}
    return out


def end_to_end(args, C, N, full=False):
    """disk -> wav through the drop-in CLI (scripts/sptk/apply_adaptive_beamformer.py), PCM16 wav +
    numpy masks in, PCM16 wav out, on files written to /dev/shm (or TMPDIR); the reference's
    boundary is apply_adaptive_beamformer.py:125-180 (first read to last close).  Every run reports
    the whole process's wall clock (interpreter + plan + page-locked slabs + the work).
    Default: n files, two runs -> `process_rtf` (the faster).  full: n and 16 n files, three and
    FIVE runs -> median / min / max of the process-level rate at both sizes and of the marginal
    input rate (each 16n run against the median n run), with the spread in the record.  (At 8 n
    the difference of two process clocks still carried +-20 % of run-to-run noise: the same
    +-0.05 s on a 0.9 s run.)"""
    import shutil
    import subprocess
    import tempfile
    from setk_amd import synth
    from setk_amd.libs import wavio
    n1 = args.e2e_utts
    n2 = 16 * n1 if full else n1
    reps = 3 if full else 2
    T = 1 + N // 256
    in_bytes = 2 * C * N + 4 * T * 257
    need = n2 * (in_bytes + 2 * N) * 1.1
    base = None
    for cand in ("/dev/shm", os.environ.get("TMPDIR", "/tmp")):
        try:
            if os.path.isdir(cand) and os.access(cand, os.W_OK) and shutil.disk_usage(cand).free > need:
                base = cand
                break
        except OSError:
            pass
    if base is None:
        return {"error": "no scratch directory with %.1f GB free" % (need / 1e9)}
    d = tempfile.mkdtemp(prefix="setk_e2e_", dir=base)
    t_leg = time.perf_counter()
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
            cmd = [sys.executable, os.path.join(ROOT, "scripts", "sptk", "apply_adaptive_beamformer.py"),
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
            st = prof.get("stages") or {}
            return {"utts": n, "written": done, "wall_s_process": round(wall, 3),
                    "wall_s_first_read_to_last_write": prof.get("wall_s"),
                    "pipeline_wall_s": st.get("wall_s"),
                    "stages": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}}

        def stats(vals):
            v = sorted(vals)
            med = v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
            return {"median": round(med, 3), "min": round(v[0], 3), "max": round(v[-1], 3),
                    "spread": round((v[-1] - v[0]) / med, 3) if med else None}

        out = {"workload": f"{C}-ch {N / SR:g} s PCM16 wav + float32 numpy masks on {base}, {args.beamformer}, "
                           "PCM16 wav out, through scripts/sptk/apply_adaptive_beamformer.py; whole-process "
                           "wall clock", "unit": "x real time (audio seconds per wall second)", "sizes": {}}
        walls = {}
        for n in ([n1, n2] if full else [n1]):
            run_cli(n)   # (first touch of freshly written page-cache pages: not counted)
            runs = [run_cli(n) for _ in range(5 if (full and n == n2) else reps)]
            bad = [r for r in runs if "error" in r]
            if bad:
                out["error"] = bad[0]["error"]
                return out
            walls[n] = [r["wall_s_process"] for r in runs]
            out["sizes"][str(n)] = {"written": [r["written"] for r in runs], "audio_s": n * N / SR,
                                    "wall_s_process": stats(walls[n]),
                                    "process_rtf": stats([n * N / SR / w for w in walls[n]]),
                                    "last_run_stages": runs[-1]["stages"],
                                    "wall_s_first_read_to_last_write": [r["wall_s_first_read_to_last_write"] for r in runs]}
        out["process_rtf"] = out["sizes"][str(n1)]["process_rtf"]["median" if full else "max"]
        out["process_wall_s"] = out["sizes"][str(n1)]["wall_s_process"]["median" if full else "min"]
        if full:
            out["process_rtf_16n"] = out["sizes"][str(n2)]["process_rtf"]["median"]
            w1 = out["sizes"][str(n1)]["wall_s_process"]["median"]
            dms = [(w2 - w1) / (n2 - n1) for w2 in walls[n2]]
            # (None when a difference of two process clocks is not positive: sizes too small for the noise)
            ok = min(dms) > 0
            out["marginal_ms_per_utt"] = stats([1e3 * dm for dm in dms]) if ok else None
            out["marginal_GBps_in"] = stats([in_bytes / dm / 1e9 for dm in dms]) if ok else None
            out["marginal_value"] = stats([(N / SR) / dm for dm in dms]) if ok else None
            # what the host can copy at all (threads -> GB/s): the input bytes are copied once
            # from the page cache into page-locked slabs before the DMA
            out["host_copy_GBps"] = host_copy_rate()
        out["seconds_spent"] = round(time.perf_counter() - t_leg, 1)
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)
