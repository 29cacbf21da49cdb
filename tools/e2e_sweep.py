#!/usr/bin/env python
"""End-to-end CLI sweep on one set of files: python tools/e2e_sweep.py N "flags a" "flags b" ...
(8-ch 30 s PCM16 wavs + float32 npy masks in /dev/shm; prints the pipeline's own clock)."""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from setk_amd import synth  # noqa: E402
from setk_amd.libs import wavio  # noqa: E402

n = int(sys.argv[1])
C, N, T = 8, 480000, 1876
d = tempfile.mkdtemp(prefix="setk_sweep_", dir="/dev/shm")
try:
    os.makedirs(f"{d}/wav"); os.makedirs(f"{d}/mask")
    rng = np.random.default_rng(0)
    for i in range(4):
        wavio.write_pcm16(f"{d}/wav/u{i}.wav", wavio.float_to_pcm16(synth.synth_utterance(i, C, N).T), 16000)
        np.save(f"{d}/mask/u{i}.npy", rng.uniform(0.05, 0.95, size=(T, 257)).astype(np.float32))
    with open(f"{d}/wav.scp", "w") as ws, open(f"{d}/mask.scp", "w") as ms:
        for i in range(n):
            if i >= 4:
                shutil.copyfile(f"{d}/wav/u{i % 4}.wav", f"{d}/wav/u{i}.wav")
                shutil.copyfile(f"{d}/mask/u{i % 4}.npy", f"{d}/mask/u{i}.npy")
            ws.write(f"u{i} {d}/wav/u{i}.wav\n"); ms.write(f"u{i} {d}/mask/u{i}.npy\n")
    for flags in sys.argv[2:]:
        shutil.rmtree(f"{d}/enh", ignore_errors=True)
        env = dict(os.environ)
        # "KEY=VALUE ..." tokens in front of the flags go to the environment
        toks = flags.split()
        while toks and "=" in toks[0] and not toks[0].startswith("-"):
            k, v = toks.pop(0).split("=", 1)
            env[k] = v
        cmd = [sys.executable, os.path.join(ROOT, "scripts/sptk/apply_adaptive_beamformer.py"),
               "--mask-format", "numpy", "--profile", f"{d}/prof.json"] + toks + \
              [f"{d}/wav.scp", f"{d}/mask.scp", f"{d}/enh"]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        wall = time.perf_counter() - t0
        if r.returncode:
            print(flags, "FAILED", r.stderr[-400:]); continue
        st = json.load(open(f"{d}/prof.json"))["stages"]
        keep = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()
                if k in ("t_read", "t_launch", "t_write", "t_slot_wait", "t_alloc", "wall_s",
                         "zero_copy_payloads", "staged_payloads", "read_threads")}
        print(f"[{flags}] process {wall:.2f} s, pipeline {st['wall_s']:.3f} s = "
              f"{1e3 * st['wall_s'] / n:.3f} ms/utt  {keep}", flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
