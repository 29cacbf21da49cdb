#!/usr/bin/env bash
# Collect the rocprofv3 evidence behind bench.py's numbers (run on the GPU box,
# from the repo root):   bash tools/collect_profiles.sh [tag]
# Output goes to gpurun_out/prof_<tag>/<workload>/ ; tools/summarize_profiles.py
# condenses it into profiles/.  Workloads (BASELINE.json configs):
#   cfg2_mvdr8   8-ch 30 s MVDR, 125 utterances      (configs[2] shard, the bench line)
#   cfg3_gevd8   8-ch 30 s GEV,  125 utterances      (configs[3])
#   cfg1_mvdr4   4-ch 10 s MVDR, 500 utterances      (configs[1])
#   cfg4_cgmm6   6-ch 30 s CGMM (20 EM) -> MVDR, 125 utterances (configs[4], tools/bench_cgmm.py)
# PMC passes are separate runs with --pmc only (never combined with tracing).
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
B="--steps 5 --warmup 1 --cpu-sample 0 --e2e-utts 0 --full-batch 0 --sustain-sec 0 --other-configs 0 --pmc 0"
run() {  # name, command...
  local name=$1; shift
  mkdir -p "$OUT/$name"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name/kt" -- "$@" > "$OUT/$name/kt.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$name/pmc_fetch" -- "$@" > "$OUT/$name/fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$name/pmc_write" -- "$@" > "$OUT/$name/write.log" 2>&1
  grep -h '^{' "$OUT/$name/kt.log" | tail -1 > "$OUT/$name/line.json"
}
run cfg2_mvdr8 python bench.py $B
run cfg3_gevd8 python bench.py $B --beamformer gevd
run cfg1_mvdr4 python bench.py $B --channels 4 --seconds 10 --utts 500
run cfg4_cgmm6 python tools/bench_cgmm.py --utts 125 --seconds 30 --steps 2
# SQ counters of the bench workload
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/cfg2_mvdr8/pmc_sq1" -- python bench.py $B > "$OUT/cfg2_mvdr8/sq1.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/cfg2_mvdr8/pmc_sq2" -- python bench.py $B > "$OUT/cfg2_mvdr8/sq2.log" 2>&1
# un-profiled lines of the same build (never compare a profiled arm with an un-profiled one)
python bench.py $B --steps 20 --warmup 3 2>/dev/null | tail -1 > "$OUT/cfg2_mvdr8/bench_unprofiled.json"
python bench.py $B --steps 20 --warmup 3 --beamformer gevd 2>/dev/null | tail -1 > "$OUT/cfg3_gevd8/bench_unprofiled.json"
python bench.py $B --steps 20 --warmup 3 --channels 4 --seconds 10 --utts 500 2>/dev/null | tail -1 > "$OUT/cfg1_mvdr4/bench_unprofiled.json"
python tools/bench_cgmm.py --utts 125 --seconds 30 --steps 3 2>/dev/null | tail -1 > "$OUT/cfg4_cgmm6/bench_unprofiled.json"
# the full default line (CPU legs, end-to-end leg)
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
find "$OUT" -name "*kernel_stats.csv" | head; tail -c 1500 "$OUT/bench.json"
