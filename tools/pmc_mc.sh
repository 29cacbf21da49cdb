#!/usr/bin/env bash
# SQ / GRBM counters of the streaming kernels (legacy and matrix-core forms), one rocprofv3 --pmc
# pass per counter group: bash tools/pmc_mc.sh tag ["ENV=.. ENV=.."]
set -u
TAG=${1:-q}; ENVS=${2:-}
OUT=gpurun_out/pmcmc_${TAG}
mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python bench.py --child 1 --gpus 1 --steps 3 --warmup 1"
rocprofv3 -L 2>/dev/null | grep -io "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > "$OUT/mfma_counters.txt"
pass() { # name counters...
  local n=$1; shift
  env $ENVS rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$n" -- $BENCH > "$OUT/$n.log" 2>&1 || echo "pass $n failed"
}
pass g1 GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass g2 SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT
pass g3 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH
pass g4 FETCH_SIZE
pass g5 WRITE_SIZE
pass g6 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_MFMA_F16
pass g7 SQ_WAVES_LT_64 SQ_THREAD_CYCLES_VALU SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_WAIT_IFETCH SQ_CYCLES
pass g8 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr
rocprofv3 -L > "$OUT/counters_all.txt" 2>&1
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
d = collections.defaultdict(list); dur = collections.defaultdict(list)
for p in glob.glob(out + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if not any(s in k for s in ("stft_covar", "beamform_istft")): continue
        if ", true>" in k: continue
        k = k.split("(")[0].replace("void setk::", "")
        d[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out + "/summary.txt", "w") as f:
    for k in sorted(dur):
        ms = sum(dur[k]) / len(dur[k]) / 1e6
        line = f"{k}  profiled {ms:.4f} ms"
        print(line); f.write(line + "\n")
    for k in sorted(d):
        line = f"{k[0]:44s} {k[1]:30s} {sum(d[k]) / len(d[k]):.5g}"
        print(line); f.write(line + "\n")
PY
cat "$OUT/mfma_counters.txt" | tr '\n' ' '
