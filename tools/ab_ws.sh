#!/usr/bin/env bash
# old (SETK_P1_WS=0) vs wave-specialised pass 1, for each library given
for L in "$@"; do for w in 0 1; do
  SETK_P1_WS=$w SETK_LIB=$PWD/$L python bench.py --steps 10 --warmup 3 --cpu-sample 0 2>/dev/null | tail -1 > /tmp/ab.json
  python - "$L ws=$w" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], d["ms_per_step"], d["stage_ms"])
PY
done; done
