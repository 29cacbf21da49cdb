#!/usr/bin/env bash
# quick SQ counter passes for the two streaming kernels: bash tools/pmc_quick.sh tag
set -u
TAG=${1:-q}
OUT=gpurun_out/pmcq_${TAG}
mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python bench.py --steps 3 --warmup 1 --cpu-sample 0"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/sq1" -- $BENCH > "$OUT/l1.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/sq2" -- $BENCH > "$OUT/l2.log" 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT --output-format csv -d "$OUT/sq3" -- $BENCH > "$OUT/l3.log" 2>&1
python - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for p in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if "stft_covar_kernel<8, false" in r["Kernel_Name"] or "beamform_istft_kernel<8" in r["Kernel_Name"]:
            d[(r["Kernel_Name"][11:22], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(d): print(k[0], k[1], "%.4g"%(sum(d[k])/len(d[k])))
PY
