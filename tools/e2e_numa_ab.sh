#!/usr/bin/env bash
# --numa auto against --numa off, one and two processes, 8192 utterances each (8-ch 10 s),
# interleaved repeats: bash tools/e2e_numa_ab.sh
for rep in 1 2; do
  for mode in auto off; do
    PLIST="1 2" bash tools/e2e_steady.sh 8192 10 --numa $mode > /dev/null 2>&1
    grep -E "^P=[12]:" gpurun_out/e2e_steady.txt | sed "s/^/rep $rep numa=$mode /"
  done
done
