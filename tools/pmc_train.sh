#!/bin/bash
# PMC passes over tools/train_once.py (fused training step, 4096 x 64): HBM bytes and LDS conflicts per kernel.
# Separate passes (FETCH_SIZE / WRITE_SIZE do not fit together); run on the GPU box: tools/pmc_train.sh <tag>
set -e
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_train_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  d=$OUT/$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -- python3 $REPO/tools/train_once.py > $d.log 2>&1 || echo "pass $pass failed"
done
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("dw_gemm", "nerf_mlp_bwd", "nerf_mlp_bf16_16_kernel<true, true>", "sample_encode_bf16", "composite"):
            if key in k:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        v = v[len(v) // 2:]            # the later (warm) dispatches
        print(f"    {c:24s} mean={sum(v)/len(v):.6g}  n={len(v)}")
PY
find $OUT -name "*.csv" -size +1M -delete
