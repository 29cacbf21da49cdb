set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round5
export TMPDIR=/tmp
O=gpurun_out/round5
python tools/error_budget.py 2>&1 | grep -v "Warn\|amdgpu.ids" > $O/error_budget.txt; tail -12 $O/error_budget.txt
(time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=15) > $O/pytest_gpu.log 2>&1
tail -6 $O/pytest_gpu.log | cut -c1-200
B="--cpu-sample 0 --pmc 0 --other-configs 0 --full-batch 0 --e2e-utts 0"
rm -rf $O/kt_cfg2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg2 -- python bench.py --steps 20 --warmup 5 $B > $O/kt_cfg2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg3 -- python bench.py --steps 20 --warmup 5 $B --beamformer gevd --int16-ingest 0 > $O/kt_cfg3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg1 -- python bench.py --steps 20 --warmup 5 $B --channels 4 --seconds 10 --utts 500 --int16-ingest 0 > $O/kt_cfg1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg4 -- python tools/bench_cgmm.py --utts 125 --seconds 30 --steps 2 > $O/kt_cfg4.log 2>&1
(time timeout 900 python bench.py) > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/round5/bench.json") if l.startswith("{")][-1])
print("ms", d["ms_per_step"], d["stage_ms"], "traffic", d["roofline"].get("traffic"), d["roofline"]["pass2"]["hbm"])
i = d.get("int16_ingest", {})
print("int16", {k: i.get(k) for k in ("ms_per_step", "enhance_only_ms", "ingest_ms", "stage_ms", "bit_identical_to_float32_path_on_pcm_over_32768")})
PY
python __graft_entry__.py smoke 2>&1 | tail -3
