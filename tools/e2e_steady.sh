#!/usr/bin/env bash
# Steady-state host-side scaling of the end-to-end CLI (round-3 review: the 256-utterance runs
# of tools/e2e_multi.sh were mostly start-up): P = 1, 2, 4, 8 CLI processes, each over its OWN
# table of N utterances (default 2048; 8-ch 10 s PCM16 wav + npy mask), all on the one GPU of
# this box.  The files are hard links onto 128 distinct ones (1.5 GB: beyond the CPU caches,
# within /dev/shm).  Reported per P: first-read-to-last-write clock of the slowest process, and
# the MARGINAL rate from the difference to a quarter-size run of the same P (start-up, plan,
# slab allocation and the first batch cancel).
#   bash tools/e2e_steady.sh [utts_per_process=2048] [seconds=10] [extra CLI flags]
set -u
N=${1:-2048}; SEC=${2:-10}; shift 2 || true
FLAGS="$*"
D=/dev/shm/setk_steady; rm -rf $D; mkdir -p $D/src $D/wav $D/mask gpurun_out
python - "$N" "$D" "$SEC" <<'PY'
import os, sys, numpy as np
sys.path.insert(0, ".")
from setk_amd import synth
from setk_amd.libs import wavio
n, d, sec = int(sys.argv[1]), sys.argv[2], float(sys.argv[3])
ns = int(16000 * sec); T = 1 + ns // 256
rng = np.random.default_rng(0)
K = 128
base = [synth.synth_utterance(i, 8, ns) for i in range(4)]
for i in range(K):
    mix = np.roll(base[i % 4], 997 * (i // 4), axis=1)  # distinct bytes, cheap to make
    wavio.write_pcm16(f"{d}/src/u{i}.wav", wavio.float_to_pcm16(mix.T), 16000)
    np.save(f"{d}/src/m{i}.npy", rng.uniform(0.05, 0.95, size=(T, 257)).astype(np.float32))
for p in range(8):
    for frac, tag in ((1, "full"), (4, "quarter")):
        with open(f"{d}/wav.{p}.{tag}.scp", "w") as ws, open(f"{d}/mask.{p}.{tag}.scp", "w") as ms:
            for i in range(n // frac):
                k = (i * 8 + p) % K
                w, m = f"{d}/wav/p{p}u{i}.wav", f"{d}/mask/p{p}u{i}.npy"
                if frac == 1:
                    os.link(f"{d}/src/u{k}.wav", w); os.link(f"{d}/src/m{k}.npy", m)
                ws.write(f"p{p}u{i} {w}\n"); ms.write(f"p{p}u{i} {m}\n")
print("bytes per utterance", os.path.getsize(f"{d}/src/u0.wav") + os.path.getsize(f"{d}/src/m0.npy"))
PY
OUTF=gpurun_out/e2e_steady.txt; : > $OUTF
echo "# $N utterances per process, 8-ch ${SEC} s, flags: $FLAGS" | tee -a $OUTF
run() { # P tag
  rm -rf $D/enh.*; rm -f $D/prof.*.json
  for p in $(seq 0 $(($1-1))); do
    python scripts/sptk/apply_adaptive_beamformer.py --mask-format numpy $FLAGS \
       --profile $D/prof.$p.json $D/wav.$p.$2.scp $D/mask.$p.$2.scp $D/enh.$p 2> $D/log.$p.txt &
  done
  wait
  python - "$1" "$D" <<'PY'
import sys, json
P, D = int(sys.argv[1]), sys.argv[2]
w = []
for p in range(P):
    try:
        w.append(json.load(open(f"{D}/prof.{p}.json"))["wall_s"])
    except Exception:
        w.append(float("nan"))
print(max(w))
PY
}
for P in ${PLIST:-1 2 4 8}; do
  Q=$(run $P quarter); F=$(run $P full)
  python - "$P" "$N" "$Q" "$F" "$SEC" <<'PY' | tee -a $OUTF
import sys
P, N, Q, F, sec = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5])
marg = P * (N - N // 4) / (F - Q)
print(f"P={P}: full {F:.2f} s ({P*N/F:.0f} utt/s), quarter {Q:.2f} s -> marginal {marg:.0f} utt/s aggregate, "
      f"{marg/P:.0f} per process, {marg*sec:.0f} x real time")
PY
done
# stage clocks of the last full run's process 0 (where the host time goes)
python - "$D" <<'PY' | tee -a $OUTF
import json, sys
try:
    st = json.load(open(f"{sys.argv[1]}/prof.0.json"))
    keep = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.get("stages", {}).items()}
    print("# process 0 of the last run: wall", round(st["wall_s"], 3), "stages", keep)
except Exception as e:
    print("# no stage record:", e)
PY
grep -h "launch ms" $D/log.0.txt | tail -4 | sed 's/^/# /' | tee -a $OUTF
grep -h "bound to NUMA" $D/log.0.txt | head -1 | tee -a $OUTF
rm -rf $D
