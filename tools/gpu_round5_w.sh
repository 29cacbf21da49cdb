cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/round5
O=gpurun_out/round5/cgmm_four_wg_ab.txt
: > $O
SETK_CGMM_CFG=6 timeout 900 python -m pytest tests/test_gpu_cgmm.py -q -p no:cacheprovider 2>&1 | tail -3 | tee -a $O
for rep in 1 2 3; do
  for CFG in -1 6; do
    echo "## SETK_CGMM_CFG=$CFG" >> $O
    SETK_CGMM_CFG=$CFG python tools/bench_cgmm.py --utts 125 --seconds 30 --steps 5 2>&1 | grep -v amdgpu.ids >> $O
  done
done
cat $O
