#!/usr/bin/env python
"""Where the fixed cost of a SHORT command-line run goes: n files (default 192 x 8-ch x 30 s PCM16 +
float32 numpy masks in /dev/shm) through scripts/sptk/apply_adaptive_beamformer.py, `reps` times;
prints the process wall clock next to the CLI's own marks (--profile) and stage sums.
    python tools/e2e_fixed.py [--utts 192] [--reps 5] [--env KEY=VAL ...]"""
import argparse
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=192)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--env", action="append", default=[], help="KEY=VAL; several sets separated by '--env ,'")
    ap.add_argument("--extra", default="", help="extra CLI arguments")
    a = ap.parse_args()
    from setk_amd import synth
    from setk_amd.libs import wavio
    C, N = a.channels, int(a.seconds * 16000)
    T = 1 + N // 256
    d = tempfile.mkdtemp(prefix="setk_fix_", dir="/dev/shm")
    try:
        os.makedirs(f"{d}/wav")
        os.makedirs(f"{d}/mask")
        rng = np.random.default_rng(0)
        for i in range(4):
            wavio.write_pcm16(f"{d}/wav/u{i}.wav", wavio.float_to_pcm16(synth.synth_utterance(i, C, N).T), 16000)
            np.save(f"{d}/mask/u{i}.npy", rng.uniform(0.05, 0.95, size=(T, 257)).astype(np.float32))
        with open(f"{d}/wav.scp", "w") as ws, open(f"{d}/mask.scp", "w") as ms:
            for i in range(a.utts):
                if i >= 4:
                    shutil.copyfile(f"{d}/wav/u{i % 4}.wav", f"{d}/wav/u{i}.wav")
                    shutil.copyfile(f"{d}/mask/u{i % 4}.npy", f"{d}/mask/u{i}.npy")
                ws.write(f"u{i} {d}/wav/u{i}.wav\n")
                ms.write(f"u{i} {d}/mask/u{i}.npy\n")
        sets = [[]]
        for e in a.env:
            if e == ",":
                sets.append([])
            else:
                sets[-1].append(e)
        for envset in sets:
            env = dict(os.environ, **dict(e.split("=", 1) for e in envset))
            walls = []
            for r in range(a.reps + 1):
                shutil.rmtree(f"{d}/enh", ignore_errors=True)
                cmd = [sys.executable, os.path.join(ROOT, "scripts", "sptk", "apply_adaptive_beamformer.py"),
                       "--mask-format", "numpy", "--profile", f"{d}/prof.json"] + a.extra.split() + \
                      [f"{d}/wav.scp", f"{d}/mask.scp", f"{d}/enh"]
                t0 = time.perf_counter()
                rr = subprocess.run(cmd, capture_output=True, text=True, env=env)
                wall = time.perf_counter() - t0
                if rr.returncode != 0:
                    print(rr.stderr[-1500:])
                    return 1
                prof = json.load(open(f"{d}/prof.json"))
                if r == 0:
                    continue   # first touch of the fresh page-cache pages
                walls.append(wall)
                st = prof["stages"]
                print(json.dumps({"env": envset, "process_wall_s": round(wall, 3), "marks": prof.get("marks"),
                                  "cli_wall_s": round(prof["wall_s"], 3), "pipeline_wall_s": round(st["wall_s"], 3),
                                  "t_alloc": round(st["t_alloc"], 3), "t_release": round(st.get("t_release", 0), 3),
                                  "t_read": round(st["t_read"], 3), "t_slot_wait": round(st["t_slot_wait"], 3)}))
            walls.sort()
            print(f"# {envset}: process wall median {walls[len(walls) // 2]:.3f} s, min {walls[0]:.3f}, max {walls[-1]:.3f} "
                  f"({a.utts} files, {a.reps} runs)")
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
