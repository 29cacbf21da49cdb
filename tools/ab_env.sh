#!/usr/bin/env bash
# A/B libraries with per-library environment: bash tools/ab_env.sh tag "ENV=.. lib.so" "lib2.so" ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 3 --cpu-sample 0 --e2e-utts 0 --full-batch 0 --sustain-sec 0"
for rep in 1 2 3; do
  for spec in "$@"; do
    L=${spec##* }; ENVS=${spec% *}; [ "$ENVS" = "$spec" ] && ENVS=""
    env $ENVS SETK_LIB=$PWD/$L $B 2>/dev/null | tail -1 > /tmp/ab.json
    python - "$spec" "$rep" <<'PY' | tee -a "$OUT/ab.txt"
import json, sys
try:
    d = json.load(open("/tmp/ab.json"))
    print(sys.argv[2], sys.argv[1], d["ms_per_step"], d["stage_ms"], d.get("uncached_call", {}).get("ms_per_step"))
except Exception as e:
    print(sys.argv[2], sys.argv[1], "FAILED", e)
PY
  done
done
