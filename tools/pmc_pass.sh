#!/bin/bash
# one extra rocprofv3 PMC pass over bench.py: tools/pmc_pass.sh <tag> <counter> [<counter> ...]
set -e
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/run.log 2>&1 || { echo "rocprofv3 failed"; tail -5 $OUT/run.log; exit 1; }
python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nerf_mlp" in r.get("Kernel_Name",""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(f"{k:34s} mean={sum(v)/len(v):.6g} n={len(v)}")
PY
find $OUT -name "*.csv" -size +1M -delete
