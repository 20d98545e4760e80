#!/usr/bin/env python3
"""gpurun_out/<tag>/ (tools/collect_profiles.sh) -> profiles/<name>_kernel_stats.csv, <name>_summary.md and
<name>_traffic.json (HBM bytes per launch of the dominant kernel, corrected as MI355X_MICROARCH.md
prescribes: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read -> x2; WRITE_SIZE exact;
both in KiB)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    # (a tag collected more than once holds every run's files: the newest)
    fs = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


stats = one("trace/**/*_kernel_stats.csv")
shutil.copy(stats, os.path.join(dst, f"{name}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
lines = [f"# {name}: rocprofv3 --kernel-trace --stats of `python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-pair` (--no-pair: without the two-resamples-in-flight section, whose launches overlap)", "",
         "| kernel | calls | avg us | min us | max us | total ms | % |", "|---|---|---|---|---|---|---|"]
for r in rows:
    lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
                 f"{float(r['MaxNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['Percentage']):.1f} |")


def pmc(sub):
    f = one(f"{sub}/**/*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            # (one entry per kernel AND grid: the side records of the bench launch the same kernels at other sizes)
            acc[f'{r["Kernel_Name"]} @grid={r["Grid_Size"]}'][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write, l2 = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_l2")
traffic = {}
lines += ["", "## HBM traffic per launch (PMC, separate passes)", "",
          "FETCH_SIZE / WRITE_SIZE are KiB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies the 128-B",
          "requests of a wide coalesced read at 64 B, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact.", "",
          "| kernel | launches | FETCH_SIZE KiB | WRITE_SIZE KiB | corrected HBM bytes / launch | L2 hit rate |", "|---|---|---|---|---|---|"]
for k in sorted(set(fetch) | set(write)):
    if "cpm::" not in k:
        continue
    f = fetch.get(k, {}).get("FETCH_SIZE", [])
    w = write.get(k, {}).get("WRITE_SIZE", [])
    fa = sum(f) / len(f) if f else 0.0
    wa = sum(w) / len(w) if w else 0.0
    hit = l2.get(k, {}).get("TCC_HIT_sum", [])
    miss = l2.get(k, {}).get("TCC_MISS_sum", [])
    hr = (sum(hit) / (sum(hit) + sum(miss))) if hit and (sum(hit) + sum(miss)) else float("nan")
    total = 2 * fa * 1024 + wa * 1024
    traffic[k] = {"launches": len(f), "fetch_kib": fa, "write_kib": wa, "hbm_bytes_per_launch": total, "l2_hit_rate": hr}
    kn, _, grid = k.partition(" @grid=")
    lines.append(f"| `{kn[:60]}` grid {grid} | {len(f)} | {fa:.0f} | {wa:.0f} | {total:.3e} | {hr:.3f} |")
sweep = one("sweep/**/*_kernel_stats.csv")
if sweep:
    shutil.copy(sweep, os.path.join(dst, f"{name}_sweep_kernel_stats.csv"))
    lines += ["", "## Model-selection sweep (tools/sweep_bench.py: Z = 2,357 x 1,000 cars/zone, travel times on, table rebuilds per point)", "",
              "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
    for r in list(csv.DictReader(open(sweep)))[:14]:
        lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['Percentage']):.1f} |")
    log = os.path.join(src, "sweep.log")
    if os.path.exists(log):
        lines += [""] + ["    " + l.rstrip() for l in open(log) if "grid points" in l or "IVP" in l]
head = os.popen(f"git -C {root} rev-parse --short HEAD 2>/dev/null").read().strip()
dirty = os.popen(f"git -C {root} status --porcelain -- carparkingmaps_amd include bench.py tools 2>/dev/null").read().strip()
lines.insert(1, f"(tree at {head or 'unknown commit'}{' + uncommitted changes to the sources' if dirty else ''}; collected by tools/collect_profiles.sh)")
open(os.path.join(dst, f"{name}_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(os.path.join(dst, f"{name}_traffic.json"), "w"), indent=1)
print("\n".join(lines))
