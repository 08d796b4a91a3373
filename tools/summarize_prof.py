#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into one text summary and one JSON file.

    summarize_prof.py <profile dir> [kernel-name substring ...]

The JSON (<profile dir>/pmc.json; committed as profiles/<tag>_pmc.json) holds, per kernel whose name
contains one of the substrings (default: nerf_mlp, dw_gemm, composite), the mean of every counter per
dispatch and the kernel-trace durations, plus the bench.py arguments the passes were taken with
(environment BENCH_ARGS, set by tools/profile_gpu.sh): bench.py reads `traffic` out of it and only for the
same mode / precision and the exact kernel instantiation.

Launch geometry comes from the trace's Grid_Size_X / Workgroup_Size_X columns.  Register and LDS numbers are NOT
taken from rocprofv3 (its VGPR_Count column reports 112 for a kernel whose code object says 222, and its
LDS_Block_Size leaves out dynamic LDS): tools/kernel_meta.py prints them from the code object's own metadata."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
keys = sys.argv[2:] or ["nerf_mlp", "dw_gemm", "composite"]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "")[:70]


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace", "*kernel_stats.csv"):
    for i, row in enumerate(csv.reader(open(f))):
        if i < 14:
            print(", ".join(c[:110] for c in row))
print()
summary = {"source": os.path.basename(os.path.normpath(out)), "bench_args": os.environ.get("BENCH_ARGS"), "kernels": {}}
print("== per-dispatch durations (kernel trace) ==")
dur = defaultdict(list)
meta = {}
for f in find("trace", "*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if any(k in n for k in keys):
            dur[short(n)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
            wg = int(r.get("Workgroup_Size_X") or 0) * int(r.get("Workgroup_Size_Y") or 1) * int(r.get("Workgroup_Size_Z") or 1)
            grid = int(r.get("Grid_Size_X") or 0) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1)
            meta[short(n)] = {"workgroups": grid // wg if wg else "?", "threads_per_workgroup": wg or "?"}
for n, d in dur.items():
    print(f"{n:70s} n={len(d)} mean={sum(d)/len(d):9.3f} ms min={min(d):9.3f} max={max(d):9.3f}  {meta[n]}")
    summary["kernels"].setdefault(n, {})["duration_ms_mean"] = sum(d) / len(d)
    summary["kernels"][n]["dispatches"] = len(d)
    summary["kernels"][n].update(meta[n])
print()
print("== PMC counters, mean per dispatch ==")
for sub in sorted(os.path.basename(p) for p in glob.glob(os.path.join(out, "pmc_*")) if os.path.isdir(p)):
    acc = defaultdict(lambda: defaultdict(list))
    for f in find(sub, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            n = r.get("Kernel_Name", "")
            if any(k in n for k in keys):
                acc[short(n)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, cs in acc.items():
        for k, v in cs.items():
            print(f"{sub:10s} {n:60s} {k:30s} mean={sum(v)/len(v):.6g}  n={len(v)}")
            summary["kernels"].setdefault(n, {})[k] = sum(v) / len(v)
json.dump(summary, open(os.path.join(out, "pmc.json"), "w"), indent=1)
