cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round5
export TMPDIR=/tmp
O=gpurun_out/round5
: > $O/e2e_read_tuning.txt
for rep in 1 2; do
for CFG in "256:0" "1024:0" "1024:6" "256:6" "100000:6" "100000:12"; do
  KB=${CFG%%:*}; TH=${CFG#*:}
  FL=""; [ "$TH" != "0" ] && FL="--read-threads $TH"
  echo "## SETK_MMAP_MIN_KB=$KB $FL" >> $O/e2e_read_tuning.txt
  SETK_MMAP_MIN_KB=$KB PLIST="1" bash tools/e2e_steady.sh 2048 10 $FL > /dev/null 2>&1; grep "^P=" gpurun_out/e2e_steady.txt >> $O/e2e_read_tuning.txt
done
done
cat $O/e2e_read_tuning.txt
