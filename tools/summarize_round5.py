#!/usr/bin/env python
"""profiles/round5_*: copy this round's evidence out of gpurun_out/round5 (tools/gpu_round5_f.sh)
and condense the rocprofv3 kernel-stats tables into profiles/round5_summary.md."""
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "round5")
DST = os.path.join(ROOT, "profiles")
NAMES = {"kt_cfg2": ("cfg2_mvdr8", "BASELINE configs[2] shard: 125 x 8-ch x 30 s MVDR, `python bench.py --steps 20 --warmup 5` "
                                   "(float32 steps, then the `int16_ingest` leg: the `<..., true>` kernels)"),
         "kt_cfg3": ("cfg3_gevd8", "configs[3]: 125 x 8-ch x 30 s GEV"),
         "kt_cfg1": ("cfg1_mvdr4", "configs[1]: 500 x 4-ch x 10 s MVDR"),
         "kt_cfg4": ("cfg4_cgmm6", "configs[4]: 125 x 6-ch x 30 s CGMM (20 EM) -> MVDR, `tools/bench_cgmm.py`")}
out = ["# Round 5: rocprofv3 `--kernel-trace --stats` per workload (one MI355X)\n",
       "Made by `tools/gpu_round5_f.sh` + `tools/summarize_round5.py`.  Average durations are per launch, "
       "under the profiler (a profiled run clocks ~2 - 4 % lower than an un-profiled one: compare with "
       "`stage_ms` of `round5_bench.json`, HIP events inside un-profiled timed steps).\n"]
for d, (tag, what) in NAMES.items():
    files = glob.glob(os.path.join(SRC, d, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        continue
    shutil.copy(files[0], os.path.join(DST, f"round5_{tag}_kernel_stats.csv"))
    rows = [r for r in csv.DictReader(open(files[0])) if "setk::" in r["Name"]]
    out.append(f"\n## {tag} -- {what}\n\n| kernel | calls | average us | share of GPU time |\n|---|---|---|---|")
    for r in rows[:10]:
        name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        name = name.split("(")[0][:90]
        out.append(f"| `{name}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} % |")
line = [l for l in open(os.path.join(SRC, "bench.json")) if l.startswith("{")][-1]
open(os.path.join(DST, "round5_bench.json"), "w").write(line)
d = json.loads(line)
i = d["int16_ingest"]
out.append("\n## The default bench line of the same build (`round5_bench.json`)\n")
out.append(f"* float32 headline: {d['ms_per_step']} ms per step = {d['value']:.0f} x real time; stage_ms {d['stage_ms']}; "
           f"stage-1 HBM frac {d['roofline']['frac']}, traffic {d['roofline']['traffic'] / 1e9:.3f} GB "
           f"({d['roofline']['pass1']['hbm']['traffic_over_algorithmic']} x algorithmic), VALU-issue frac "
           f"{d['roofline']['pass1']['valu_issue']['frac']}; stage 3 traffic "
           f"{d['roofline']['pass2']['hbm']['traffic'] / 1e9:.3f} GB ({d['roofline']['pass2']['hbm']['traffic_over_algorithmic']} x)")
out.append(f"* int16_ingest: {i['ms_per_step']} ms from interleaved frames (ingest {i['ingest_ms']} ms at {i['ingest_gbs']} GB/s), "
           f"{i['enhance_only_ms']} ms with planar int16 resident; stage_ms {i['stage_ms']}; roofline frac {i['roofline']['frac']} "
           f"of 2 C N + 4 T F, traffic {i['roofline'].get('traffic')}; bit-identical to the float32 path: "
           f"{i['bit_identical_to_float32_path_on_pcm_over_32768']}")
out.append(f"* cpu_baseline: {d['cpu_baseline']['value']} x real time on {d['cpu_baseline']['cores']} core ({d['cpu_baseline']['kind']}); "
           f"parity in the run: worst {d['cpu_baseline']['parity_check']['worst_rel_rms_vs_oracle']}")
oc = d.get("other_configs", {})
for k, v in oc.items():
    if isinstance(v, dict) and "ms_per_step" in v:
        out.append(f"* {k}: {v['ms_per_step']} ms per step, stage_ms {v.get('stage_ms')}")
out.append(f"* full_batch: {d.get('full_batch')}")
e = d.get("end_to_end", {})
out.append(f"* end_to_end: marginal {e.get('marginal_ms_per_utt')} ms per utterance, {e.get('marginal_GBps_in')} GB/s of input")
cu = oc.get("consumers_and_unfused", {})
for k in ("df_on_mask_4ch_resident", "wpd_4ch_resident"):
    if k in cu:
        out.append(f"* {k}: " + str({kk: vv for kk, vv in cu[k].items() if kk != "workload"}))
open(os.path.join(DST, "round5_summary.md"), "w").write("\n".join(out) + "\n")
for src, dst in (("error_budget.txt", "round5_error_budget.txt"), ("e2e_p1_sweep.txt", "round5_e2e_p1_sweep.txt")):
    shutil.copy(os.path.join(SRC, src), os.path.join(DST, dst))
# (round5_read_small.txt was assembled by hand from three runs: numpy / page-locked destination)
shutil.copy(os.path.join(ROOT, "gpurun_out", "stall_r5b", "summary.md"), os.path.join(DST, "round5_stall_table.md"))
print(open(os.path.join(DST, "round5_summary.md")).read())
