#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of the bench
# command; writes summaries under gpurun_out/prof_<tag>/ (copy the ones to keep into profiles/).
# Usage: tools/profile_gpu.sh <tag> [bench args]      e.g.  tools/profile_gpu.sh r02 --mode train
set -e
TAG=${1:-r02}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-aux $*"
run() {   # run <subdir> <rocprofv3 options...>
  local sub=$1; shift
  rocprofv3 "$@" --output-format csv -d $OUT/$sub -- python3 $REPO/bench.py $ARGS > $OUT/$sub.log 2>&1 || echo "$sub failed"
  echo "$sub done"
}
run trace --kernel-trace --stats
run pmc_sq --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU
run pmc_sq2 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
# keep only the small summaries
find $OUT -name "*.csv" -size +2M -delete
BENCH_ARGS="$*" python3 $REPO/tools/summarize_prof.py $OUT > $OUT/SUMMARY.txt 2>&1 || true
cat $OUT/SUMMARY.txt
