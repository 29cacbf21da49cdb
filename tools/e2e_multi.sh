#!/usr/bin/env bash
# Host-side scaling of the end-to-end CLI BEFORE an 8-GPU node shows it: P = 1, 2, 4, 8 CLI
# processes run concurrently against DISJOINT shards of PCM16 wav + npy mask files in /dev/shm,
# all on the one GPU of this box (what contends is the host: first-read page-cache rate,
# descriptor tables, pinned-slab allocation, the PCIe link -- the kernels are ~3 % of the time).
#   bash tools/e2e_multi.sh [utts_per_process=256] [extra CLI flags]
# Prints one line per P: wall clock of the slowest process, aggregate utterances/s, aggregate
# input GB/s; writes gpurun_out/e2e_multi.txt.
set -u
N=${1:-256}; shift || true
FLAGS="$*"
D=/dev/shm/setk_multi; rm -rf $D; mkdir -p $D/wav $D/mask gpurun_out
python - "$N" "$D" <<'PY'
import sys, shutil, numpy as np
sys.path.insert(0, ".")
from setk_amd import synth
from setk_amd.libs import wavio
n, d = 8 * int(sys.argv[1]), sys.argv[2]
rng = np.random.default_rng(0)
for i in range(4):
    mix = synth.synth_utterance(i, 8, 480000)
    wavio.write_pcm16(f"{d}/wav/u{i}.wav", wavio.float_to_pcm16(mix.T), 16000)
    np.save(f"{d}/mask/u{i}.npy", rng.uniform(0.05, 0.95, size=(1876, 257)).astype(np.float32))
for i in range(4, n):
    shutil.copyfile(f"{d}/wav/u{i % 4}.wav", f"{d}/wav/u{i}.wav")
    shutil.copyfile(f"{d}/mask/u{i % 4}.npy", f"{d}/mask/u{i}.npy")
for p in range(8):
    with open(f"{d}/wav.{p}.scp", "w") as ws, open(f"{d}/mask.{p}.scp", "w") as ms:
        for i in range(p * int(sys.argv[1]), (p + 1) * int(sys.argv[1])):
            ws.write(f"u{i} {d}/wav/u{i}.wav\n"); ms.write(f"u{i} {d}/mask/u{i}.npy\n")
PY
BYTES_PER_UTT=$((2*8*480000 + 4*1876*257))
: > gpurun_out/e2e_multi.txt
for P in 1 2 4 8; do
  rm -rf $D/enh.*; rm -f $D/prof.*.json
  S=$(date +%s.%N)
  for p in $(seq 0 $((P-1))); do
    python scripts/sptk/apply_adaptive_beamformer.py --mask-format numpy $FLAGS \
       --profile $D/prof.$p.json $D/wav.$p.scp $D/mask.$p.scp $D/enh.$p 2> $D/log.$p.txt &
  done
  wait
  E=$(date +%s.%N)
  python - "$P" "$N" "$S" "$E" "$BYTES_PER_UTT" "$D" <<'PY' | tee -a gpurun_out/e2e_multi.txt
import sys, json, glob
P, N, S, E, B, D = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
inner = []
for p in range(P):
    try:
        inner.append(json.load(open(f"{D}/prof.{p}.json"))["wall_s"])
    except Exception:
        inner.append(float("nan"))
wall = E - S
print(f"P={P}: {P*N} utterances, wall {wall:.2f} s (processes incl. start-up), first-read-to-last-write "
      f"max {max(inner):.2f} s -> aggregate {P*N/max(inner):.0f} utt/s, {P*N*B/max(inner)/1e9:.1f} GB/s in, "
      f"{P*N*30/max(inner):.0f} x real time")
PY
done
rm -rf $D
