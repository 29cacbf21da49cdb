cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5y
mkdir -p $O
for w in wpe nfft400 ch12; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -- python tools/prof_secondary.py $w > $O/$w.log 2>&1
  grep "per utterance" $O/$w.log
  f=$(find $O/$w -name "*kernel_stats.csv" | head -1)
  head -9 "$f" | cut -c1-150
done
