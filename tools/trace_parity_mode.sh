#!/bin/bash
# Kernel trace (rocprofv3) of the reference-jitter modes of tools/bench_aux.py pcie: prints start / duration / hardware
# queue of the MT19937 generator kernels and the MLP renders.  Run on the GPU box.
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_pcie
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/tools/bench_aux.py pcie > $OUT/run.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "mt19937" in r["Kernel_Name"] or "nerf_mlp" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
# print the last 44 (the final timed image-driver repetitions)
for r in rows[-46:]:
    n="RNG" if "mt19937" in r["Kernel_Name"] else "MLP"
    print(n, "start %.3f ms  dur %.3f ms  queue %s" % ((int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, r.get("Queue_Id","?")))
PY
