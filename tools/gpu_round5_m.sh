cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for L in setk_amd/libsetk_hip.so _abl/libsetk_frswap.so _abl/libsetk_frnostore.so; do
SETK_BENCH_NOCHECK=1 SETK_LIB=$PWD/$L python bench.py --steps 60 --warmup 20 --cpu-sample 0 --pmc 0 --other-configs 0 --full-batch 0 --e2e-utts 0 2>/dev/null | tail -1 > /tmp/ab.json
python - $L <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
i = d["int16_ingest"]
print(sys.argv[1], "FLOAT", d["stage_ms"]["stft_covar"], "PLANAR", i["stage_ms"]["stft_covar"], "FRAMES", i["frames_direct"]["ms_per_step"], i["frames_direct"]["stage_ms"], i["bit_identical_to_float32_path_on_pcm_over_32768"])
PY
done
