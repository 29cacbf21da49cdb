#!/usr/bin/env bash
# pass-1 forms per channel count: butterflies (default) against the matrix-core transform waves
#   bash tools/ab_channels.sh [tag]
TAG=${1:-abch}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
B="--steps 20 --warmup 3 --cpu-sample 0 --e2e-utts 0 --full-batch 0 --sustain-sec 0 --other-configs 0 --pmc 0"
for rep in 1 2; do
for cfg in "1 10 2000" "2 10 1000" "4 10 500" "8 30 125"; do
  set -- $cfg
  for mc in 0 1; do
    SETK_MC_PASS1=$mc python bench.py $B --channels $1 --seconds $2 --utts $3 2>/dev/null | tail -1 > /tmp/abc.json
    python - "$1" "$mc" "$rep" <<'PY' | tee -a "$OUT/ab.txt"
import json, sys
d = json.load(open("/tmp/abc.json"))
print("rep", sys.argv[3], "C", sys.argv[1], "MC_PASS1", sys.argv[2], d["ms_per_step"], d["stage_ms"])
PY
  done
done
done
