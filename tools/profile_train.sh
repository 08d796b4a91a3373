#!/bin/bash
# rocprofv3 kernel-trace stats of the training step (bench_aux.py c5); run on the GPU box
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_train_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/bench_aux.py c5 > $OUT/trace.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.reader(open(f)))
print(", ".join(rows[0]))
for r in rows[1:28]:
    r[0]=r[0][:90]
    print(", ".join(r))
PY
