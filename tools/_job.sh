export TMPDIR=/tmp
O=gpurun_out/round6_h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cgmm.py tests/test_gpu_baseline_sizes.py -q -x -m gpu -p no:cacheprovider -k "cgmm or cfg4" 2>&1 | tail -4 | tee $O/pytest_cgmm.txt
for r in 1 2 3; do for L in default _abl/libsetk_cgpk0.so; do
  if [ "$L" = default ]; then unset SETK_LIB; else export SETK_LIB=$PWD/$L; fi
  echo "CGMM round $r $L: $(timeout 300 python tools/bench_cgmm.py --utts 125 --channels 6 --seconds 30 --iters 20 --steps 5 2>&1 | tail -1)"
done; done 2>&1 | tee $O/cgmm_pk_ab.txt
for C in 4 8; do for L in default _abl/libsetk_cgpk0.so; do
  if [ "$L" = default ]; then unset SETK_LIB; else export SETK_LIB=$PWD/$L; fi
  echo "CGMM C=$C $L: $(timeout 300 python tools/bench_cgmm.py --utts 125 --channels $C --seconds 30 --iters 20 --steps 3 2>&1 | tail -1)"
done; done 2>&1 | tee -a $O/cgmm_pk_ab.txt
unset SETK_LIB
SETK_LIB=$PWD/_abl/libsetk_cgphases.so SETK_CGMM_TIMING=$PWD/$O/timing_phases.txt timeout 300 python tools/bench_cgmm.py --utts 125 --channels 6 --seconds 30 --iters 20 --steps 1 > /dev/null
SETK_CGMM_TIMING=$PWD/$O/timing_product.txt timeout 300 python tools/bench_cgmm.py --utts 125 --channels 6 --seconds 30 --iters 20 --steps 1 > /dev/null
python tools/cgmm_phases.py $O/timing_phases.txt $O/timing_product.txt | tee $O/phases.md
