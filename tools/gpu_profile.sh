#!/usr/bin/env bash
# The rocprofv3 evidence behind the bench line, one lease:  bash tools/gpu_profile.sh <tag>
#   gpurun_out/<tag>/cfg2_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the driver's bench command
#                                               (float32 and PCM16 forms of both streaming kernels, the de-interleave)
#   gpurun_out/<tag>/cfg{1,3}_kernel_stats.csv  the same command at configs[1] (4-ch 10 s x 500) and configs[3] (8-ch GEV)
#   gpurun_out/<tag>/cfg4_kernel_stats.csv      the same for configs[4] (tools/bench_cgmm.py)
#   gpurun_out/<tag>/bench_aux.json             python bench.py --aux 1 (every leg)
TAG=${1:-prof}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --gpus 1 --steps 20 --warmup 5 --pmc 0 --cpu-sample 0 --full-batch 0 --e2e-utts 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -- $B > $O/kt2.log 2>&1
cp $(find $O/kt2 -name "*kernel_stats.csv" | head -1) $O/cfg2_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt4 -- python tools/bench_cgmm.py --utts 125 --channels 6 --seconds 30 --iters 20 --steps 3 > $O/kt4.log 2>&1
cp $(find $O/kt4 -name "*kernel_stats.csv" | head -1) $O/cfg4_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -- $B --channels 4 --seconds 10 --utts 500 > $O/kt1.log 2>&1
cp $(find $O/kt1 -name "*kernel_stats.csv" | head -1) $O/cfg1_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- $B --beamformer gevd > $O/kt3.log 2>&1
cp $(find $O/kt3 -name "*kernel_stats.csv" | head -1) $O/cfg3_kernel_stats.csv
rm -rf $O/kt1 $O/kt2 $O/kt3 $O/kt4
head -8 $O/cfg2_kernel_stats.csv | cut -c1-200
( time timeout 1500 python bench.py --aux 1 ) > $O/bench_aux.json 2> $O/bench_aux.err
tail -4 $O/bench_aux.err
python - <<PY
import json
rec = json.loads([l for l in open("$O/bench_aux.json") if l.startswith("{")][-1])
print(json.dumps({k: v for k, v in rec.items() if not isinstance(v, (dict, list))}))
e = rec.get("end_to_end", {})
print("e2e:", {k: e.get(k) for k in ("process_rtf", "process_wall_s", "process_rtf_16n", "marginal_GBps_in", "marginal_ms_per_utt", "error")})
for k, v in (rec.get("other_configs") or {}).items():
    if isinstance(v, dict) and "ms_per_step" in v: print(k, v["ms_per_step"], v.get("value"))
print("power:", rec.get("power")); print("cpu all cores:", (rec.get("cpu_baseline") or {}).get("all_cores"))
PY
