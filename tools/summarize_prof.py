#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into one text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace", "*kernel_stats.csv"):
    for i, row in enumerate(csv.reader(open(f))):
        if i < 12:
            print(", ".join(row))
print()
print("== per-dispatch durations of the fused MLP kernel (kernel trace) ==")
for f in find("trace", "*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if "nerf_mlp" in r.get("Kernel_Name", ""):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            print(f'{r["Kernel_Name"][:60]:60s} {d:9.3f} ms  grid={r.get("Grid_Size","?")} wg={r.get("Workgroup_Size","?")} '
                  f'vgpr={r.get("VGPR_Count","?")} agpr={r.get("Accum_VGPR_Count","?")} sgpr={r.get("SGPR_Count","?")} lds={r.get("LDS_Block_Size","?")}')
print()
print("== PMC counters, fused MLP kernel, mean per dispatch ==")
for sub in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    acc = defaultdict(list)
    for f in find(sub, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "nerf_mlp" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{sub:10s} {k:34s} mean={sum(v)/len(v):.6g}  n={len(v)}")
