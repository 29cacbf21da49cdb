#!/usr/bin/env bash
# One GPU-box visit: the GPU test-suite, then the default bench line.
#   bash tools/gpu_check.sh [tag] [pytest -k expression]
TAG=${1:-chk}; KEXPR=${2:-}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
if [ -n "$KEXPR" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -s -k "$KEXPR" > "$OUT/pytest.log" 2>&1
else
  timeout 1500 python -m pytest tests -m gpu -x -q -s > "$OUT/pytest.log" 2>&1
fi
echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
tail -5 "$OUT/pytest.log"
grep -h "^\[" "$OUT/pytest.log" | head -40
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"
tail -c 6000 "$OUT/bench.json"
tail -5 "$OUT/bench.err"
