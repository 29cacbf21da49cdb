#!/usr/bin/env bash
# Collect the rocprofv3 evidence behind bench.py's numbers (run on the GPU box,
# from the repo root):   bash tools/collect_profiles.sh [tag]
# Output goes to gpurun_out/prof_<tag>/ ; tools/summarize_profiles.py condenses
# it into profiles/.
# PMC passes are separate runs with --pmc only (never combined with tracing).
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python bench.py --steps 5 --warmup 1 --cpu-sample 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $BENCH > "$OUT/bench_kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/bench_write.log" 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_tcc" -- $BENCH > "$OUT/bench_tcc.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_sq1" -- $BENCH > "$OUT/bench_sq1.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/pmc_sq2" -- $BENCH > "$OUT/bench_sq2.log" 2>&1
python bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
find "$OUT" -name "*.csv" | head -40
tail -1 "$OUT/bench.json" | cut -c1-400
