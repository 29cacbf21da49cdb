export TMPDIR=/tmp
O=gpurun_out/round6_e; mkdir -p $O
for r in 1 2 3; do for L in default _abl/libsetk_p2f32nc.so; do
  if [ "$L" = default ]; then unset SETK_LIB; else export SETK_LIB=$PWD/$L; fi
  python bench.py --steps 100 --warmup 30 --cpu-sample 0 --full-batch 0 --e2e-utts 0 --int16-ingest 0 2>/dev/null | tail -1 > /tmp/ab.json
  python - "$L" $r <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
p2 = d["roofline"].get("pass2", {})
print(f"AB round {sys.argv[2]} {sys.argv[1]}: step {d['ms_per_step']} stages {d['stage_ms']} pass2 traffic/alg {d.get('pass2_traffic_over_algorithmic')} "
      f"frac {d['roofline']['frac']}")
PY
done; done 2>&1 | tee $O/ab_f32_carry.txt
unset SETK_LIB
for L in default _abl/libsetk_p2f32nc.so; do
  if [ "$L" = default ]; then unset SETK_LIB; else export SETK_LIB=$PWD/$L; fi
  echo "bits $L"; timeout 300 python tools/ab_bits.py 2>&1 | tail -8
done | tee $O/ab_bits.txt
unset SETK_LIB
timeout 900 python -m pytest tests/test_gpu_enhance.py tests/test_gpu_baseline_sizes.py tests/test_gpu_api.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest.txt
