#!/usr/bin/env bash
# A/B of an environment toggle within one box: bash tools/ab_env.sh VAR "0 1" [bench args]
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  for rep in 1 2; do
    env $VAR=$v python bench.py --steps 10 --warmup 3 --cpu-sample 0 "$@" 2>/dev/null | tail -1 > /tmp/ab.json
    python - "$VAR=$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], d["ms_per_step"], d["stage_ms"], d["roofline"]["frac"])
PY
  done
done
