#!/usr/bin/env bash
# rocprofv3 kernel + memory-copy trace of the streaming CLI on N utterances in /dev/shm:
#   bash tools/e2e_trace.sh [N]   -> gpurun_out/e2e_trace/ (stats csv files)
N=${1:-256}
export TMPDIR=/tmp
D=$(mktemp -d /dev/shm/setk_trace_XXXX)
python - "$D" "$N" <<'PY'
import os, shutil, sys, numpy as np
sys.path.insert(0, os.getcwd())
from setk_amd import synth
from setk_amd.libs import wavio
d, n = sys.argv[1], int(sys.argv[2])
C, N, T = 8, 480000, 1876
os.makedirs(d + "/wav"); os.makedirs(d + "/mask")
rng = np.random.default_rng(0)
for i in range(4):
    wavio.write_pcm16(f"{d}/wav/u{i}.wav", wavio.float_to_pcm16(synth.synth_utterance(i, C, N).T), 16000)
    np.save(f"{d}/mask/u{i}.npy", rng.uniform(0.05, 0.95, size=(T, 257)).astype(np.float32))
with open(d + "/wav.scp", "w") as ws, open(d + "/mask.scp", "w") as ms:
    for i in range(n):
        if i >= 4:
            shutil.copyfile(f"{d}/wav/u{i % 4}.wav", f"{d}/wav/u{i}.wav")
            shutil.copyfile(f"{d}/mask/u{i % 4}.npy", f"{d}/mask/u{i}.npy")
        ws.write(f"u{i} {d}/wav/u{i}.wav\n"); ms.write(f"u{i} {d}/mask/u{i}.npy\n")
PY
OUT=gpurun_out/e2e_trace; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$OUT/rp" -- \
  python scripts/sptk/apply_adaptive_beamformer.py --mask-format numpy --beamformer mvdr \
  --profile "$OUT/cli_profile.json" "$D/wav.scp" "$D/mask.scp" "$D/enh" > "$OUT/run.log" 2>&1
echo "rc=$?"; ls "$D/enh" | wc -l
rm -rf "$D"
find "$OUT" -name "*stats*.csv" | head; for f in $(find "$OUT" -name "*memory_copy_stats.csv" -o -name "*kernel_stats.csv"); do echo "== $f"; head -12 "$f"; done
# keep the merge small: the raw traces are not needed
find "$OUT" -name "*_trace.csv" -size +2M -delete
