export TMPDIR=/tmp
mkdir -p gpurun_out/round6_c
timeout 900 bash tools/ab_pcm.sh 2 default _abl/libsetk_p2pcm512.so _abl/libsetk_p2pcm1024nc.so 2>&1 | tee gpurun_out/round6_c/ab_pcm.txt
timeout 300 python -m pytest tests/test_gpu_enhance.py tests/test_gpu_baseline_sizes.py -q -x -m gpu -p no:cacheprovider -k "pcm16 or streaming or frames" 2>&1 | tail -3 | tee gpurun_out/round6_c/pytest_pcm.txt
for L in default _abl/libsetk_cgr5.so; do
  if [ "$L" = default ]; then unset SETK_LIB; else export SETK_LIB=$PWD/$L; fi
  for r in 1 2; do echo "CGMM $L: $(timeout 300 python tools/bench_cgmm.py --utts 125 --channels 6 --seconds 30 --iters 20 --steps 5)"; done
done 2>&1 | tee gpurun_out/round6_c/cgmm_solve_ab.txt
unset SETK_LIB
SETK_CGMM_TIMING=$PWD/gpurun_out/round6_c/timing_product.txt timeout 300 python tools/bench_cgmm.py --utts 125 --channels 6 --seconds 30 --iters 20 --steps 1 > /dev/null
python tools/cgmm_phases.py gpurun_out/round6_c/timing_product.txt | tee gpurun_out/round6_c/phases_product.md
timeout 600 python -m pytest tests/test_gpu_cgmm.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/round6_c/pytest_cgmm.txt
for B in 8 16 32; do timeout 300 python tools/e2e_fixed.py --reps 3 --extra "--batch-utts $B" 2>&1 | grep "^#" | sed "s/^/batch-utts $B: /"; done | tee gpurun_out/round6_c/e2e_batch_utts_192.txt
for B in 16 32; do timeout 600 python tools/e2e_fixed.py --utts 1536 --reps 2 --extra "--batch-utts $B" 2>&1 | grep "^#" | sed "s/^/batch-utts $B: /"; done | tee gpurun_out/round6_c/e2e_batch_utts_1536.txt
