cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round5
export TMPDIR=/tmp
O=gpurun_out/round5
: > $O/e2e_switch_interval.txt
for rep in 1 2; do
for SI in 0.005 0.0005 0.0001 0.00002; do
  echo "## SETK_SWITCH_INTERVAL=$SI" >> $O/e2e_switch_interval.txt
  SETK_SWITCH_INTERVAL=$SI PLIST="1" bash tools/e2e_steady.sh 2048 10 > /dev/null 2>&1; grep "^P=" gpurun_out/e2e_steady.txt >> $O/e2e_switch_interval.txt
done
done
cat $O/e2e_switch_interval.txt
