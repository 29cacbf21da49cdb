#!/usr/bin/env bash
# A/B a list of experimental libraries (tools/mk_abl.sh) against the product build
# on ONE box: timing (3 repeats each, interleaved) and a parity check per variant.
#   bash tools/ab_round.sh tag lib1.so lib2.so ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
LIBS="setk_amd/libsetk_hip.so $@"
B="python bench.py --steps 20 --warmup 3 --cpu-sample 0 --e2e-utts 0 --full-batch 0 --sustain-sec 0 --other-configs 0 --pmc 0"
for rep in 1 2 3; do
  for L in $LIBS; do
    SETK_LIB=$PWD/$L $B 2>/dev/null | tail -1 > /tmp/ab.json
    python - "$L" "$rep" <<'PY' | tee -a "$OUT/ab.txt"
import json, sys
try:
    d = json.load(open("/tmp/ab.json"))
    print(sys.argv[2], sys.argv[1], d["ms_per_step"], d["stage_ms"])
except Exception as e:
    print(sys.argv[2], sys.argv[1], "FAILED", e)
PY
  done
done
for L in "$@"; do
  SETK_LIB=$PWD/$L timeout 600 python -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_enhance.py -x -q -m gpu -k "cfg1 or (cfg2_cfg3 and mvdr) or matches_oracle or ragged or edge_geom" 2>&1 | tail -2 | sed "s|^|$L: |" | tee -a "$OUT/ab.txt"
done
