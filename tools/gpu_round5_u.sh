cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round5
export TMPDIR=/tmp
O=gpurun_out/round5
: > $O/e2e_native_sweep.txt
for rep in 1 2; do
for FL in "" "--pipeline-depth 4" "--pipeline-depth 6" "--batch-utts 16 --pipeline-depth 6" "--batch-utts 64 --pipeline-depth 3" "--read-threads 8 --pipeline-depth 4"; do
  echo "## native $FL" >> $O/e2e_native_sweep.txt
  PLIST="1" bash tools/e2e_steady.sh 4096 30 $FL > /dev/null 2>&1; grep "^P=\|stage" gpurun_out/e2e_steady.txt | cut -c1-520 >> $O/e2e_native_sweep.txt
done
done
cat $O/e2e_native_sweep.txt
