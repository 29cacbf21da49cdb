#!/usr/bin/env bash
# Where do the non-issue cycles of the two streaming kernels go?  (round-5 review, item 3)
# rocprofv3 --att needs the thread-trace decoder library, which this image does not ship
# (only its header, /opt/rocm/include/rocprofiler-sdk/experimental/thread-trace); the SQ
# counters give the same attribution per kernel: SQ_WAVE_CYCLES splits into
#   SQ_ACTIVE_INST_ANY (a wave issuing: VALU / LDS / VMEM / SCA / MISC sub-buckets)
# + SQ_WAIT_INST_ANY   (issue stall: the instruction is ready, its pipe is not -- LDS sub-bucket)
# + SQ_WAIT_ANY        (parked: s_waitcnt on memory / LDS returns, s_barrier)
# (MI355X_MICROARCH.md, "rocprofv3 PMC slots": the three are disjoint and sum to WAVE_CYCLES).
# Run for the default build and for the single-role builds of pass 1 (transform waves only /
# covariance waves only, tools/mk_abl.sh -DSETK_ONLY_PROD / -DSETK_ONLY_CONS), counters only,
# one pass per group.   bash tools/stall_table.sh <tag> [lib ...]
set -u
TAG=${1:-r5}; shift || true
LIBS=("default" "$@")
OUT=gpurun_out/stall_${TAG}
mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python bench.py --steps 3 --warmup 1 --cpu-sample 0 --pmc 0 --int16-ingest 0 --other-configs 0 --full-batch 0 --e2e-utts 0"
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_LDS"
G3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"
for L in "${LIBS[@]}"; do
  N=$(basename "$L" .so)
  if [ "$L" = default ]; then unset SETK_LIB; else export SETK_LIB=$PWD/$L SETK_BENCH_NOCHECK=1; fi
  i=0
  for G in "$G1" "$G2" "$G3"; do
    i=$((i+1))
    rocprofv3 --pmc $G --output-format csv -d "$OUT/$N/g$i" -- $BENCH > "$OUT/$N.g$i.log" 2>&1 || echo "pass g$i of $N failed" >&2
  done
done
python tools/stall_table.py "$OUT" | tee "$OUT/summary.md"
