#!/usr/bin/env bash
# A/B alternative builds of the library on one box: bash tools/ab_libs.sh lib1.so lib2.so ...
for L in "$@"; do
  SETK_BENCH_NOCHECK=1 SETK_LIB=$PWD/$L python bench.py --steps 8 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 > /tmp/ab.json
  python - "$L" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], d["ms_per_step"], d["stage_ms"])
PY
done
