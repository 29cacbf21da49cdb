#!/usr/bin/env bash
# kernel times (rocprofv3 --kernel-trace --stats) and SQ counters of the reduce + solve stage:
#   bash tools/pmc_solve.sh tag ["ENV=.."]
set -u
TAG=${1:-s}; ENVS=${2:-}
OUT=gpurun_out/pmcsolve_${TAG}
mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python bench.py --child 1 --gpus 1 --steps 3 --warmup 1"
env $ENVS rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $BENCH > "$OUT/kt.log" 2>&1
env $ENVS rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p1" -- $BENCH > "$OUT/p1.log" 2>&1
env $ENVS rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d "$OUT/p2" -- $BENCH > "$OUT/p2.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for p in glob.glob(out + "/kt/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(p)):
        print("%-70s calls %4s avg %10.1f ns  total%% %s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
d = collections.defaultdict(list); dur = collections.defaultdict(list)
for p in glob.glob(out + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "solve_kernel" not in k and "covar_finalize" not in k: continue
        k = k.split("(")[0].replace("void setk::", "")
        d[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(dur): print(k, "profiled %.4f ms" % (sum(dur[k]) / len(dur[k]) / 1e6))
for k in sorted(d): print("%-40s %-28s %.5g" % (k[0], k[1], sum(d[k]) / len(d[k])))
PY
