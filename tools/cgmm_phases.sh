#!/usr/bin/env bash
# configs[4]'s EM kernel (cgmm_bin_em_kernel<6, 256, 8, 4, 3>), attributed (round-5 review, item 5):
#   1. in-kernel phase clocks: the -DSETK_CGMM_PHASES build (SCHED=iterative-ilp bash tools/mk_abl.sh
#      cgphases cgmm_bin -DSETK_CGMM_PHASES) dumps wave 0's cycles per phase (SETK_CGMM_TIMING)
#   2. the SQ stall split of the product build (three counter passes, counters only)
#   3. HBM write traffic of the product build, three separate runs (it varied 0.97 - 1.43 GB)
# bash tools/cgmm_phases.sh <tag> [lib]      -> gpurun_out/<tag>/
set -u
TAG=${1:-cgmm}; LIB=${2:-}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
[ -n "$LIB" ] && export SETK_LIB=$PWD/$LIB
B="python tools/bench_cgmm.py --utts 125 --channels 6 --seconds 30 --iters 20"
$B --steps 3 | tee $O/bench.json
SETK_CGMM_TIMING=$PWD/$O/timing_product.txt $B --steps 1 > /dev/null
if [ -z "$LIB" ] && [ -f _abl/libsetk_cgphases.so ]; then
  SETK_LIB=$PWD/_abl/libsetk_cgphases.so SETK_CGMM_TIMING=$PWD/$O/timing_phases.txt $B --steps 1 | tee $O/bench_phases_build.json
  python tools/cgmm_phases.py $O/timing_phases.txt $O/timing_product.txt | tee $O/phases.md
fi
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_LDS"
G3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $O/stall/default/g$i -- $B --steps 1 > $O/stall.g$i.log 2>&1 || echo "pass g$i failed" >&2
done
SETK_STALL_KERNELS="cgmm_bin_em_kernel" python tools/stall_table.py $O/stall | tee $O/stall.md
for r in 1 2 3; do
  rocprofv3 --pmc WRITE_SIZE FETCH_SIZE --output-format csv -d $O/wr/r$r -- $B --steps 1 > $O/wr.$r.log 2>&1
done
python - <<PY | tee $O/write_traffic.txt
import csv, glob
for r in (1, 2, 3):
    rd = wr = n = 0
    for p in glob.glob("$O/wr/r%d/**/*counter_collection.csv" % r, recursive=True):
        for row in csv.DictReader(open(p)):
            if "cgmm_bin_em_kernel" in row["Kernel_Name"]:
                if row["Counter_Name"] == "WRITE_SIZE": wr += float(row["Counter_Value"]); n += 1
                if row["Counter_Name"] == "FETCH_SIZE": rd += float(row["Counter_Value"])
    if n: print(f"run {r}: {n} launches, written {wr * 1024 / n / 1e9:.3f} GB, read {2 * rd * 1024 / n / 1e9:.3f} GB per launch")
PY
