#!/usr/bin/env bash
# Timing-only A/B of experimental libraries whose results may be wrong (ablations):
#   bash tools/ab_time.sh tag lib1.so lib2.so ...   -> gpurun_out/<tag>/ab.txt
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
B="env SETK_BENCH_NOCHECK=1 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --e2e-utts 0 --full-batch 0 --sustain-sec 0 --other-configs 0"
for rep in 1 2; do
  for L in setk_amd/libsetk_hip.so "$@"; do
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
