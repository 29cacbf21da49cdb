#!/usr/bin/env bash
# step time per beamformer kind (the configs[2] shard; gevd = configs[3])
B="--steps 20 --warmup 3 --cpu-sample 0 --e2e-utts 0 --full-batch 0 --sustain-sec 0 --other-configs 0 --pmc 0"
for k in mvdr gevd pmwf-0; do
  python bench.py $B --beamformer $k 2>/dev/null | tail -1 > /tmp/bk.json
  python - "$k" <<'PY'
import json, sys
d = json.load(open("/tmp/bk.json"))
print(sys.argv[1], d["ms_per_step"], d["stage_ms"])
PY
done
