#!/bin/bash
# rocprofv3 kernel-trace stats of one tools/bench_aux.py configuration; run on the GPU box:
#   tools/profile_aux.sh <config> [tag]      e.g. tools/profile_aux.sh c4 r01e
set -e
CFG=${1:-c4}; TAG=${2:-x}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${CFG}_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/bench_aux.py $CFG > $OUT/trace.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.reader(open(f)))
print(", ".join(rows[0]))
for r in rows[1:16]:
    r[0]=r[0][:80]
    print(", ".join(r))
PY
grep "config" $OUT/trace.log || true
