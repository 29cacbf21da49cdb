#!/usr/bin/env bash
for k in mvdr gevd pmwf-0; do
  python bench.py --steps 8 --warmup 2 --cpu-sample 0 --beamformer $k 2>/dev/null | tail -1 > /tmp/bk.json
  python - "$k" <<'PY'
import json, sys
d = json.load(open("/tmp/bk.json"))
print(sys.argv[1], d["ms_per_step"], d["stage_ms"])
PY
done
