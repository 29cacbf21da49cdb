set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round5g
export TMPDIR=/tmp
O=gpurun_out/round5g
(time timeout 1500 python -m pytest tests/test_gpu_enhance.py tests/test_gpu_api.py tests/test_gpu_baseline_sizes.py -q -p no:cacheprovider --maxfail=10 -k "pcm16 or streaming or cli or bench_contract or frames") > $O/pytest.log 2>&1
tail -25 $O/pytest.log | cut -c1-250
for i in 1 2; do
python bench.py --steps 100 --warmup 30 --cpu-sample 0 --pmc 0 --other-configs 0 --full-batch 0 --e2e-utts 0 2>/dev/null | tail -1 > /tmp/ab.json
python - <<'PY'
import json
d = json.load(open("/tmp/ab.json"))
i = d["int16_ingest"]
print("FLOAT", d["ms_per_step"], d["stage_ms"])
print("PLANAR enhance_only", i["enhance_only_ms"], i["stage_ms"], "with ingest", i["ms_per_step"])
print("FRAMES", i["frames_direct"]["ms_per_step"], i["frames_direct"]["stage_ms"], "identical", i["bit_identical_to_float32_path_on_pcm_over_32768"])
PY
done 2>&1 | tee $O/frames_ab.txt
