#!/usr/bin/env bash
# sweep the work-list granularity of the two streaming passes
for p1 in ${P1S:-250 500 750 1000 1500}; do
  for p2 in ${P2S:-500 1000 1500 2000}; do
    SETK_P1_ITEMS=$p1 SETK_P2_ITEMS=$p2 python bench.py --steps 8 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 > /tmp/sweep.json
    python - "$p1" "$p2" <<'PY'
import json, sys
d = json.load(open("/tmp/sweep.json"))
print(sys.argv[1], sys.argv[2], d["ms_per_step"], d["stage_ms"])
PY
  done
done
