set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=12) > gpurun_out/r5b_pytest.log 2>&1
tail -25 gpurun_out/r5b_pytest.log
(time timeout 600 python bench.py) > gpurun_out/r5b_bench.json 2> gpurun_out/r5b_bench.err
tail -3 gpurun_out/r5b_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5b_bench.json") if l.startswith("{")][-1])
print("ms", d["ms_per_step"], d["stage_ms"], "traffic", d["roofline"].get("traffic"))
i = d.get("int16_ingest", {})
print("int16", {k: i.get(k) for k in ("ms_per_step", "enhance_only_ms", "ingest_ms", "stage_ms", "bit_identical_to_float32_path_on_pcm_over_32768")}, i.get("roofline", {}).get("pmc"))
PY
# A/B: streaming hint on pass 2's non-reused loads
for L in setk_amd/libsetk_hip.so _abl/libsetk_p2nt.so setk_amd/libsetk_hip.so _abl/libsetk_p2nt.so; do
  SETK_BENCH_NOCHECK=1 SETK_LIB=$PWD/$L python bench.py --steps 100 --warmup 30 --cpu-sample 0 --pmc 0 --other-configs 0 --full-batch 0 --e2e-utts 0 2>/dev/null | tail -1 > /tmp/ab.json
  python - "$L" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
i = d.get("int16_ingest", {})
print("AB", sys.argv[1], d["ms_per_step"], d["stage_ms"], "int16", i.get("enhance_only_ms"), i.get("stage_ms"))
PY
done 2>&1 | tee gpurun_out/r5b_nt_ab.txt
# HBM traffic of pass 2 with the hint
SETK_LIB=$PWD/_abl/libsetk_p2nt.so SETK_BENCH_NOCHECK=1 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --other-configs 0 --full-batch 0 --e2e-utts 0 --int16-ingest 0 2>/dev/null | tail -1 > gpurun_out/r5b_nt_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5b_nt_bench.json"))
print("NT pass2 hbm", d["roofline"]["pass2"]["hbm"], d["stage_ms"])
PY
bash tools/stall_table.sh r5b _abl/libsetk_p1prod.so _abl/libsetk_p1cons.so > gpurun_out/r5b_stall.log 2>&1
tail -120 gpurun_out/stall_r5b/summary.md
