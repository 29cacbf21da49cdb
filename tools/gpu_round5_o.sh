cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round5
export TMPDIR=/tmp
for rep in 1 2; do
for L in setk_amd/libsetk_hip.so _abl/libsetk_cg_max-ilp.so _abl/libsetk_cg_max-memory-clause.so _abl/libsetk_cg_iterative-ilp.so _abl/libsetk_cg_iterative-minreg.so _abl/libsetk_cg_iterative-maxocc.so; do
  echo "AB $L $(SETK_LIB=$PWD/$L python tools/bench_cgmm.py --utts 125 --seconds 30 --steps 3 2>/dev/null | tail -1 | cut -c1-400)"
done
done 2>&1 | tee gpurun_out/round5/cgmm_sched_ab.txt
