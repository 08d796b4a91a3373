#!/bin/bash
# Samples rocm-smi (socket power, sclk) while the headline render runs: evidence for "the kernel is power-limited".
#   tools/power_trace.sh [bench args]   -> gpurun_out/power_trace.txt        (GPU box)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/power_trace.txt
python3 $REPO/bench.py --steps 150 --no-cpu-baseline --no-aux "$@" > $REPO/gpurun_out/power_trace_bench.json 2>/dev/null &
PID=$!
: > $OUT
rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -E "Max|Power|sclk" >> $OUT   # cap and idle state, before the GPU is busy
for i in $(seq 1 40); do
  if ! kill -0 $PID 2>/dev/null; then break; fi
  echo "t=$i" >> $OUT
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" >> $OUT
  sleep 0.5
done
wait $PID
echo "bench rc=$?" >> $OUT
