#!/usr/bin/env python
"""
Condense gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the tracked
evidence under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary
  profiles/<tag>_pmc_summary.md     per-kernel PMC means (HBM traffic, SQ)
  profiles/<tag>_bench.json         the bench.py line of the same build
  profiles/pmc_traffic.json         HBM bytes per launch of the fused
                                    STFT+covariance kernel (read by bench.py)

HBM traffic follows MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE (KB) =
TCC_EA0_RDREQ x 64 B tallies 128-B requests at 64 B, i.e. reports exactly 1/2
of a coalesced streaming read -> read bytes = 2 x FETCH_SIZE x 1024.
WRITE_SIZE needs no correction.  Both calibrate on scale_kernel (reads and
writes exactly 4 bytes per output sample): see the calibration row.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path):
    d = collections.defaultdict(list)
    for p in glob.glob(os.path.join(path, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(p)):
            d[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


def short(name):
    name = name.replace("void ", "").replace("setk::", "")
    return name.split("(")[0]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    ks = glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))
    if ks:
        shutil.copy(ks[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    bench = None
    bj = os.path.join(src, "bench.json")
    if os.path.exists(bj):
        lines = [ln for ln in open(bj).read().splitlines() if ln.startswith("{")]
        if lines:
            bench = json.loads(lines[-1])
            json.dump(bench, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)
    allc = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_tcc", "pmc_sq1", "pmc_sq2"):
        allc.update(counters(os.path.join(src, sub)))
    kernels = sorted({k for k, _ in allc if "setk::" in k})
    names = sorted({c for _, c in allc})
    stats = {}
    if ks:
        for r in csv.DictReader(open(ks[0])):
            stats[r["Name"]] = r
    out = [f"# rocprofv3 evidence, tag `{tag}`", "",
           "Command: `python bench.py --steps 5 --warmup 1 --cpu-sample 0` (125 x 8-ch x 30 s "
           "utterances per step) under `rocprofv3 --kernel-trace --stats` and separate "
           "`--pmc` passes (tools/collect_profiles.sh).  Values are per-launch means.", ""]
    out.append("| kernel | calls | avg us | FETCH_SIZE KB (raw) | HBM read MB (2x corrected) | "
               "WRITE_SIZE KB | HBM write MB |")
    out.append("|---|---|---|---|---|---|---|")
    traffic = {}
    for k in kernels:
        fs = allc.get((k, "FETCH_SIZE"))
        ws = allc.get((k, "WRITE_SIZE"))
        st = stats.get(k, {})
        rd = 2.0 * fs * 1024 if fs is not None else None
        wr = ws * 1024 if ws is not None else None
        traffic[k] = (rd, wr)
        out.append("| {} | {} | {} | {} | {} | {} | {} |".format(
            short(k), st.get("Calls", "-"),
            f"{float(st['AverageNs']) / 1e3:.1f}" if st else "-",
            f"{fs:.0f}" if fs is not None else "-", f"{rd / 1e6:.1f}" if rd else "-",
            f"{ws:.0f}" if ws is not None else "-", f"{wr / 1e6:.1f}" if wr else "-"))
    out += ["", "Calibration: `scale_kernel` reads and writes 125 x 480000 x 4 B = 240.0 MB; "
            "raw FETCH_SIZE shows half of that, WRITE_SIZE all of it.", ""]
    out.append("| kernel | " + " | ".join(n for n in names if n not in ("FETCH_SIZE", "WRITE_SIZE"))
               + " |")
    out.append("|---|" + "---|" * len([n for n in names if n not in ("FETCH_SIZE", "WRITE_SIZE")]))
    for k in kernels:
        row = [short(k)]
        for n in names:
            if n in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            v = allc.get((k, n))
            row.append(f"{v:.4g}" if v is not None else "-")
        out.append("| " + " | ".join(row) + " |")
    open(os.path.join(dst, f"{tag}_pmc_summary.md"), "w").write("\n".join(out) + "\n")
    # traffic json for bench.py
    k1 = [k for k in kernels if "stft_covar_kernel<8, false" in k]
    if k1 and traffic[k1[0]][0] is not None and traffic[k1[0]][1] is not None:
        rd, wr = traffic[k1[0]]
        json.dump({
            "tag": tag, "kernel": short(k1[0]), "utts": 125, "channels": 8, "samples": 480000,
            "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
            "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; "
                      "read = 2 x FETCH_SIZE KB x 1024 (gfx950 half-count correction, "
                      "MI355X_MICROARCH.md HBM section), write = WRITE_SIZE KB x 1024",
        }, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    print("\n".join(out[:14]))


if __name__ == "__main__":
    main()
