cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round5
export TMPDIR=/tmp
for rep in 1 2; do
for L in setk_amd/libsetk_hip.so _abl/libsetk_sv_default.so _abl/libsetk_sv_iterative-ilp.so _abl/libsetk_sv_iterative-minreg.so _abl/libsetk_sv_max-memory-clause.so; do
  for K in mvdr gevd; do
  SETK_BENCH_NOCHECK=1 SETK_LIB=$PWD/$L python bench.py --steps 100 --warmup 30 --cpu-sample 0 --pmc 0 --other-configs 0 --full-batch 0 --e2e-utts 0 --int16-ingest 0 --beamformer $K 2>/dev/null | tail -1 > /tmp/ab.json
  python - "$L" $K <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print("AB", sys.argv[1], sys.argv[2], d["ms_per_step"], d["stage_ms"]["reduce_solve"])
PY
  done
done
done 2>&1 | tee gpurun_out/round5/solve_sched_ab.txt
