cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp; mkdir -p gpurun_out/round5
for i in 1 2 3; do
(time python bench.py --cpu-sample 0 --full-batch 0 --other-configs 0 --pmc 0 --int16-ingest 0 --steps 20 --warmup 5) 2> gpurun_out/round5/e2e_leg_$i.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); e = d['end_to_end']
print({k: e[k] for k in e if k.startswith('marginal')}, [(r['utts'], r['wall_s_process'], r['pipeline_wall_s']) for r in e['runs']])"
grep real gpurun_out/round5/e2e_leg_$i.err
done
