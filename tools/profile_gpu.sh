#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + PMC passes of
# the bench command; writes summaries under gpurun_out/prof_<tag>/ (copy the
# ones to keep into profiles/).  Usage: tools/profile_gpu.sh <tag> [bench args]
set -e
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
echo "trace done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $ARGS > $OUT/pmc_sq.log 2>&1 || echo "pmc_sq failed"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1 || echo "pmc_sq2 failed"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1 || echo "pmc_fetch failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > $OUT/pmc_write.log 2>&1 || echo "pmc_write failed"
# keep only the small summaries
find $OUT -name "*.csv" -size +2M -delete
python3 $REPO/tools/summarize_prof.py $OUT > $OUT/SUMMARY.txt 2>&1 || true
cat $OUT/SUMMARY.txt
