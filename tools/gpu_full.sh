#!/usr/bin/env bash
# One lease of the GPU box, the driver's two commands exactly as it runs them:
#   bash tools/gpu_full.sh <tag> [bench|tests|both]
# -> gpurun_out/<tag>/pytest_gpu.log  (python -m pytest tests/ -x -q -m gpu)
#    gpurun_out/<tag>/bench_driver_form.json + .time  (python bench.py --gpus 1 --steps 20 --warmup 5)
TAG=${1:-run}; WHAT=${2:-both}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
if [ "$WHAT" != bench ]; then
  ( time timeout 3000 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider -rs --durations=15 ) > $O/pytest_gpu.log 2>&1
  tail -25 $O/pytest_gpu.log | cut -c1-300
fi
if [ "$WHAT" != tests ]; then
  ( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_form.json 2> $O/bench_driver_form.err
  tail -4 $O/bench_driver_form.err
  python - <<PY
import json
rec = json.loads([l for l in open("$O/bench_driver_form.json") if l.startswith("{")][-1])
keep = {k: v for k, v in rec.items() if not isinstance(v, (dict, list))}
print(json.dumps(keep, indent=1))
print("roofline:", json.dumps({k: v for k, v in rec["roofline"].items() if not isinstance(v, (dict, list))}))
print("cpu_baseline:", json.dumps({k: v for k, v in rec.get("cpu_baseline", {}).items() if not isinstance(v, (dict, list))}))
e = rec.get("end_to_end", {}); print("e2e:", e.get("process_rtf"), e.get("process_wall_s"), e.get("seconds_spent"), e.get("error"))
PY
fi
