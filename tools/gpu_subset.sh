#!/usr/bin/env bash
# bash tools/gpu_subset.sh tag "<pytest args>" ["bench args"]
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1200 python -m pytest $2 -m gpu -x -q -s > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -25 "$OUT/pytest.log"
if [ -n "$3" ]; then
  timeout 900 python bench.py $3 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
  python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["stage_ms"])
print(json.dumps(d.get("end_to_end"), indent=1)[:3000])
print(json.dumps(d.get("cpu_baseline", {}).get("all_cores")))
PY
  tail -3 "$OUT/bench.err"
fi
