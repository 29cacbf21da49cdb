#!/usr/bin/env bash
# One GPU-box visit for the matrix-core transform kernels: parity subset, then an A/B of
# the step time with environment switches (legacy butterflies vs matrix cores).
#   bash tools/mc_check.sh tag "<pytest args>" "ENV1=.. ENV2=.." "ENV.." ...   ("-" = no environment)
TAG=$1; PYT=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
if [ -n "$PYT" ]; then
  timeout 1500 python -m pytest $PYT -m gpu -x -q > "$OUT/pytest.log" 2>&1
  echo "pytest rc=$?"; tail -15 "$OUT/pytest.log"
fi
B="python bench.py --steps 20 --warmup 3 --cpu-sample 1 --cpu-allcore-per-proc 0 --e2e-utts 0 --full-batch 0 --sustain-sec 0 --other-configs 0 --pmc 0"
for rep in 1 2 3; do
  for spec in "$@"; do
    ENVS=$spec; [ "$spec" = "-" ] && ENVS=""
    env $ENVS $B 2>"$OUT/bench.err" | tail -1 > /tmp/ab.json
    python - "$spec" "$rep" <<'PY' | tee -a "$OUT/ab.txt"
import json, sys
try:
    d = json.load(open("/tmp/ab.json"))
    print(sys.argv[2], sys.argv[1], d["ms_per_step"], d["stage_ms"], (d.get("cpu_baseline") or {}).get("parity_check"))
except Exception as e:
    print(sys.argv[2], sys.argv[1], "FAILED", e)
PY
  done
done
tail -3 "$OUT/bench.err"
