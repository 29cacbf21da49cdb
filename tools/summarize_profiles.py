#!/usr/bin/env python
"""
Condense gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the tracked
evidence under profiles/:

  profiles/<tag>_<workload>_kernel_stats.csv  rocprofv3 --kernel-trace --stats summary
  profiles/<tag>_summary.md                   per-workload kernel times, HBM traffic
                                              (FETCH_SIZE / WRITE_SIZE) and SQ counters
  profiles/<tag>_bench.json                   the default bench.py line of the same build
  profiles/pmc_traffic.json                   HBM bytes per launch of the fused
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
    name = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("setk::", "")
    return name.split("(")[0]


def load_line(path):
    try:
        lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    out = [f"# rocprofv3 evidence, tag `{tag}`", "",
           "Collected by `tools/collect_profiles.sh` on one MI355X: every workload under "
           "`rocprofv3 --kernel-trace --stats` and, in separate runs, `--pmc FETCH_SIZE` / "
           "`--pmc WRITE_SIZE` (never combined with tracing).  Values are per-launch means over "
           "the profiled run (`bench.py --steps 5 --warmup 1`); `un-profiled` is the same build "
           "and workload timed without the profiler (20 steps).  HBM read = 2 x FETCH_SIZE "
           "(gfx950 half-count, MI355X_MICROARCH.md), write = WRITE_SIZE.", ""]
    for wl in sorted(os.listdir(src)):
        wdir = os.path.join(src, wl)
        if not os.path.isdir(wdir):
            continue
        ks = glob.glob(os.path.join(wdir, "kt", "*", "*_kernel_stats.csv"))
        stats = {}
        if ks:
            shutil.copy(ks[0], os.path.join(dst, f"{tag}_{wl}_kernel_stats.csv"))
            for r in csv.DictReader(open(ks[0])):
                stats[r["Name"]] = r
        allc = {}
        for sub in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
            allc.update(counters(os.path.join(wdir, sub)))
        kernels = sorted({k for k, _ in allc if "setk::" in k} | {k for k in stats if "setk::" in k})
        line = load_line(os.path.join(wdir, "bench_unprofiled.json"))
        out += [f"## {wl}", ""]
        if line:
            brief = {k: line[k] for k in ("ms_per_step", "value", "stage_ms", "ms_per_batch", "rtf",
                                          "workload") if k in line}
            if "roofline" in line:
                brief["roofline"] = {k: line["roofline"][k] for k in ("kernel", "achieved", "frac",
                                                                      "kernel_ms")}
            out += ["un-profiled: `" + json.dumps(brief) + "`", ""]
        out.append("| kernel | calls | avg us | total % | HBM read MB | HBM write MB |")
        out.append("|---|---|---|---|---|---|")
        tot_rd = tot_wr = 0.0
        for k in kernels:
            fs, ws = allc.get((k, "FETCH_SIZE")), allc.get((k, "WRITE_SIZE"))
            st = stats.get(k, {})
            rd = 2.0 * fs * 1024 if fs is not None else None
            wr = ws * 1024 if ws is not None else None
            calls = int(st["Calls"]) if st else 0
            out.append("| {} | {} | {} | {} | {} | {} |".format(
                short(k), st.get("Calls", "-"),
                f"{float(st['AverageNs']) / 1e3:.1f}" if st else "-",
                st.get("Percentage", "-"),
                f"{rd / 1e6:.1f}" if rd is not None else "-",
                f"{wr / 1e6:.1f}" if wr is not None else "-"))
            if wl == "cfg2_mvdr8" and "stft_covar_kernel<8, false" in k and rd is not None \
                    and wr is not None:
                json.dump({
                    "tag": tag, "kernel": short(k), "utts": 125, "channels": 8, "samples": 480000,
                    "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
                    "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; "
                              "read = 2 x FETCH_SIZE KB x 1024 (gfx950 half-count correction, "
                              "MI355X_MICROARCH.md HBM section), write = WRITE_SIZE KB x 1024",
                }, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
        out.append("")
        sq = sorted({c for _, c in allc if c.startswith("SQ_")})
        if sq:
            out.append("| kernel | " + " | ".join(sq) + " |")
            out.append("|---|" + "---|" * len(sq))
            for k in kernels:
                if not any((k, c) in allc for c in sq):
                    continue
                out.append("| " + short(k) + " | " +
                           " | ".join(f"{allc[(k, c)]:.4g}" if (k, c) in allc else "-" for c in sq) + " |")
            out.append("")
    bench = load_line(os.path.join(src, "bench.json"))
    if bench:
        json.dump(bench, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)
    open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out[:60]))


if __name__ == "__main__":
    main()
