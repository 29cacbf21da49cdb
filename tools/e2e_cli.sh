#!/usr/bin/env bash
# End-to-end CLI throughput (disk -> wav): N synthetic 8-ch 30 s utterances.
set -e
N=${1:-48}
D=/tmp/e2e; rm -rf $D; mkdir -p $D/wav $D/mask
python - "$N" "$D" <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
from setk_amd import synth
from setk_amd.libs import wavio
n, d = int(sys.argv[1]), sys.argv[2]
rng = np.random.default_rng(0)
with open(f"{d}/wav.scp", "w") as ws, open(f"{d}/mask.scp", "w") as ms:
    for i in range(n):
        mix = synth.synth_utterance(i % 4, 8, 480000)
        wavio.write_pcm16(f"{d}/wav/u{i}.wav", wavio.float_to_pcm16(mix.T), 16000)
        np.save(f"{d}/mask/u{i}.npy", rng.uniform(0.05, 0.95, size=(1876, 257)).astype(np.float32))
        ws.write(f"u{i} {d}/wav/u{i}.wav\n"); ms.write(f"u{i} {d}/mask/u{i}.npy\n")
PY
S=$(date +%s.%N)
python scripts/sptk/apply_adaptive_beamformer.py --mask-format numpy --batch-utts 48 $D/wav.scp $D/mask.scp $D/enh 2> $D/log.txt
E=$(date +%s.%N)
tail -1 $D/log.txt
python - "$N" "$S" "$E" <<'PY'
import sys
n, s, e = int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3])
print(f"end-to-end CLI: {n} x 30 s utterances in {e - s:.2f} s wall (incl. python start, torch import, "
      f"wav decode, H2D, kernels, D2H, PCM16 write) -> {n * 30 / (e - s):.0f} x real time")
PY
