#!/usr/bin/env python
"""Summarise tools/stall_table.sh: cycles by reason, per kernel and build.  SQ_* cycle counters
are quad-cycles summed over all waves (MI355X_MICROARCH.md); fractions are of SQ_WAVE_CYCLES."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
KERNELS = tuple(os.environ["SETK_STALL_KERNELS"].split(";")) if os.environ.get("SETK_STALL_KERNELS") else \
    ("stft_covar_kernel<8, false", "beamform_istft_mc_kernel<8", "solve_kernel<8")
for lib in sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob(os.path.join(root, lib, "g*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            k = next((k for k in KERNELS if k in r["Kernel_Name"]), None)
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                acc[k]["__ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in KERNELS:
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        if "SQ_WAVE_CYCLES" not in c:
            continue
        wc = c["SQ_WAVE_CYCLES"]
        pct = lambda n: ("%5.1f %%" % (100.0 * c[n] / wc)) if n in c else "   n/a"
        print(f"\n### {lib}: `{k}...>`  ({c['__ns'] / 1e3:.1f} us per launch under the profiler, "
              f"{c.get('SQ_WAVES', 0):.0f} waves, {c.get('SQ_INSTS_VALU', 0):.3g} VALU + "
              f"{c.get('SQ_INSTS_MFMA', 0):.3g} MFMA + {c.get('SQ_INSTS_SALU', 0):.3g} SALU + "
              f"{c.get('SQ_INSTS_LDS', 0):.3g} LDS + {c.get('SQ_INSTS_VMEM_RD', 0) + c.get('SQ_INSTS_VMEM_WR', 0):.3g} VMEM wave-instructions)\n")
        print("| share of SQ_WAVE_CYCLES (wave-resident quad-cycles) | |")
        print("|---|---|")
        print(f"| issuing (SQ_ACTIVE_INST_ANY) | {pct('SQ_ACTIVE_INST_ANY')} |")
        for n, label in (("SQ_ACTIVE_INST_VALU", "VALU (incl. MFMA issue)"), ("SQ_ACTIVE_INST_LDS", "LDS"),
                         ("SQ_ACTIVE_INST_FLAT", "global / flat memory"), ("SQ_ACTIVE_INST_SCA", "scalar"),
                         ("SQ_ACTIVE_INST_MISC", "misc (barrier, waitcnt issue, nop)")):
            print(f"| &nbsp;&nbsp;of which {label} | {pct(n)} |")
        print(f"| issue stall: instruction ready, pipe busy (SQ_WAIT_INST_ANY) | {pct('SQ_WAIT_INST_ANY')} |")
        print(f"| &nbsp;&nbsp;of which LDS pipe (SQ_WAIT_INST_LDS) | {pct('SQ_WAIT_INST_LDS')} |")
        print(f"| parked on s_waitcnt / s_barrier (SQ_WAIT_ANY) | {pct('SQ_WAIT_ANY')} |")
        tot = sum(c.get(n, 0.0) for n in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"))
        print(f"| accounted | {100.0 * tot / wc:5.1f} % |")
        if "SQ_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            print(f"| (SQ_BUSY_CYCLES {c['SQ_BUSY_CYCLES']:.3g}, GRBM_GUI_ACTIVE {c['GRBM_GUI_ACTIVE']:.3g}, "
                  f"clock {c['GRBM_GUI_ACTIVE'] / 8 / c['__ns']:.2f} GHz) | |")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            print(f"| (matrix pipe busy cycles {c['SQ_VALU_MFMA_BUSY_CYCLES']:.3g}; LDS bank conflict cycles "
                  f"{c.get('SQ_LDS_BANK_CONFLICT', 0):.3g} of {c.get('SQ_LDS_IDX_ACTIVE', 0):.3g} active) | |")
