set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round5
export TMPDIR=/tmp
O=gpurun_out/round5
B="--cpu-sample 0 --pmc 0 --other-configs 0 --full-batch 0 --e2e-utts 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg2 -- python bench.py --steps 20 --warmup 5 $B > $O/kt_cfg2.log 2>&1
find $O/kt_cfg2 -name "*kernel_stats.csv" | head -2
python - <<'PY'
import csv, glob
for p in glob.glob("gpurun_out/round5/kt_cfg2/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(p)))
    for r in rows[:14]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
# e2e: where does the single process spend its time, and what do batch size / threads / depth buy
for F in "" "--batch-utts 64" "--batch-utts 128" "--batch-utts 64 --read-threads 24" "--batch-utts 64 --pipeline-depth 4" ; do
  SETK_PIPE_DEBUG=1 PLIST=1 bash tools/e2e_steady.sh 2048 10 $F > /dev/null 2>&1
  cat gpurun_out/e2e_steady.txt >> $O/e2e_p1_sweep.txt
done
cat $O/e2e_p1_sweep.txt
(time timeout 900 python bench.py) > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/round5/bench.json") if l.startswith("{")][-1])
print("ms", d["ms_per_step"], d["stage_ms"], "traffic", d["roofline"].get("traffic"), d["roofline"]["pass2"]["hbm"])
i = d.get("int16_ingest", {})
print("int16", {k: i.get(k) for k in ("ms_per_step", "enhance_only_ms", "ingest_ms", "stage_ms", "bit_identical_to_float32_path_on_pcm_over_32768")})
print(json.dumps(d["other_configs"].get("consumers_and_unfused", {}))[:600])
print(json.dumps(d["end_to_end"])[:300])
PY
