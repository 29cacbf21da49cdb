export TMPDIR=/tmp
O=gpurun_out/round6_d; mkdir -p $O
for r in 1 2; do for L in default _abl/libsetk_cgr5.so; do
  if [ "$L" = default ]; then unset SETK_LIB; else export SETK_LIB=$PWD/$L; fi
  echo "CGMM round $r $L: $(timeout 300 python tools/bench_cgmm.py --utts 125 --channels 6 --seconds 30 --iters 20 --steps 5 2>&1 | tail -1)"
done; done 2>&1 | tee $O/cgmm_solve_ab.txt
unset SETK_LIB
SETK_LIB=$PWD/_abl/libsetk_cgphases.so SETK_CGMM_TIMING=$PWD/$O/timing_phases.txt timeout 300 python tools/bench_cgmm.py --utts 125 --channels 6 --seconds 30 --iters 20 --steps 1 > /dev/null
SETK_CGMM_TIMING=$PWD/$O/timing_product.txt timeout 300 python tools/bench_cgmm.py --utts 125 --channels 6 --seconds 30 --iters 20 --steps 1 > /dev/null
python tools/cgmm_phases.py $O/timing_phases.txt $O/timing_product.txt | tee $O/phases.md
timeout 600 python -m pytest tests/test_gpu_cgmm.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest_cgmm.txt
for B in 4 8 12 16; do timeout 300 python tools/e2e_fixed.py --reps 3 --extra "--batch-utts $B" 2>&1 | grep "^#" | sed "s/^/192 files, batch-utts $B: /"; done | tee $O/e2e_batch_utts.txt
for B in 4 8 12 16 32; do timeout 600 python tools/e2e_fixed.py --utts 1536 --reps 3 --extra "--batch-utts $B" 2>&1 | grep "^#" | sed "s/^/1536 files, batch-utts $B: /"; done | tee -a $O/e2e_batch_utts.txt
for B in 8 16 32 64; do timeout 600 python tools/e2e_fixed.py --utts 2048 --seconds 10 --reps 3 --extra "--batch-utts $B" 2>&1 | grep "^#" | sed "s/^/2048 files of 10 s, batch-utts $B: /"; done | tee -a $O/e2e_batch_utts.txt
