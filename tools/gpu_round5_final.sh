set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round5
export TMPDIR=/tmp
O=gpurun_out/round5
(time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=15) > $O/pytest_gpu.log 2>&1
tail -6 $O/pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
(time timeout 900 python bench.py) > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/round5/bench.json") if l.startswith("{")][-1])
print("ms", d["ms_per_step"], d["stage_ms"], "roofline", {k: d["roofline"].get(k) for k in ("bound","achieved","peak","frac","traffic")})
i = d.get("int16_ingest", {})
print("int16", {k: i.get(k) for k in ("ms_per_step", "enhance_only_ms", "stage_ms", "bit_identical_to_float32_path_on_pcm_over_32768")})
print(json.dumps(d["other_configs"]["consumers_and_unfused"].get("df_on_mask_4ch_resident")))
PY
