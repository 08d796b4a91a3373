#!/bin/bash
# Run ON THE GPU BOX (via gpurun): timing-only builds of the 8-bit store path of the training kernels (csrc/nerf_device.h
# store_fragment_f8, -DF8_TIMING=n: 1 = no cross-lane exponent chain (wrong exponents), 2 = nothing stored, 3 = cached instead
# of non-temporal stores), each timed by rocprofv3 --kernel-trace over `bench.py --mode train --storage e4m3`.  The library is
# rebuilt in place: rebuild it without the flag afterwards (the box's copy is scratch).
# Result of round 4 (profiles/r04_f8_timing_variants.txt): forward / dX chain 305 / 309 us as shipped, 304 / 298 without the chain,
# 285 / 277 with no activation store at all, 318 / 322 with cached stores -- the kernels are bound by instruction issue, not bytes.
set -e
cd $GRAFT_REPO_ROOT
FLAGS="-O3 -std=c++20 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
run() {
  tag=$1
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/var_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --storage e4m3 --no-aux --steps 60 --warmup 20 > $GRAFT_REPO_ROOT/gpurun_out/var_$tag.log 2>&1
  cd $GRAFT_REPO_ROOT
  echo "== $tag"; cat gpurun_out/var_$tag/*/*kernel_stats.csv | grep -i "e4m3\|bf16_16\|bwd" | cut -d, -f1-4
  rm -rf gpurun_out/var_$tag
}
run shipped
for v in ${@:-1 2 3}; do
  touch nerf-simple_amd/csrc/nerf_device.h
  make -j8 -C nerf-simple_amd/csrc CXXFLAGS="$FLAGS -DF8_TIMING=$v" > /dev/null 2>&1
  run timing$v
done
