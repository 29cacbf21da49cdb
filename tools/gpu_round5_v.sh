cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/round5
O=gpurun_out/round5/read_small_native.txt
: > $O
timeout 600 python -m pytest tests/test_gpu_wpe.py -q -p no:cacheprovider 2>&1 | tail -3
for KB in 256 1024; do
  echo "## native, mmap threshold $KB KB, 643200-byte payloads" >> $O
  READ_SMALL_PINNED=1 READ_SMALL_MMAP_MIN_KB=$KB python tools/ubench/read_small.py 643200 4096 2>&1 | grep "native" >> $O
done
cat $O
: > gpurun_out/round5/e2e_native_small.txt
for rep in 1 2; do
for KB in 256 1024; do
  echo "## SETK_MMAP_MIN_KB=$KB" >> gpurun_out/round5/e2e_native_small.txt
  SETK_MMAP_MIN_KB=$KB PLIST="1" bash tools/e2e_steady.sh 8192 10 > /dev/null 2>&1; grep "^P=\|stage" gpurun_out/e2e_steady.txt | cut -c1-520 >> gpurun_out/round5/e2e_native_small.txt
done
done
cat gpurun_out/round5/e2e_native_small.txt
