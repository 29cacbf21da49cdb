set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=15) > gpurun_out/r5c_pytest.log 2>&1
tail -30 gpurun_out/r5c_pytest.log | cut -c1-250
grep -n "8ch real, gevd\|real noisy\|real 2spk" gpurun_out/r5c_pytest.log | head -20
for L in setk_amd/libsetk_hip.so _abl/libsetk_p2pf2.so _abl/libsetk_p2nont.so setk_amd/libsetk_hip.so _abl/libsetk_p2pf2.so; do
  SETK_BENCH_NOCHECK=1 SETK_LIB=$PWD/$L python bench.py --steps 100 --warmup 30 --cpu-sample 0 --pmc 0 --other-configs 0 --full-batch 0 --e2e-utts 0 2>/dev/null | tail -1 > /tmp/ab.json
  python - "$L" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
i = d.get("int16_ingest", {})
print("AB", sys.argv[1], d["ms_per_step"], d["stage_ms"], "int16", i.get("enhance_only_ms"), i.get("stage_ms"), i.get("bit_identical_to_float32_path_on_pcm_over_32768"))
PY
done 2>&1 | tee gpurun_out/r5c_pf2_ab.txt
python tools/bench_consumers.py > gpurun_out/r5c_consumers.json 2> gpurun_out/r5c_consumers.err; tail -c 3000 gpurun_out/r5c_consumers.json
