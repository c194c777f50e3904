#!/bin/bash
# Round 4: A/B of library builds on the curvilinear (CSI_METRIC_FULL) configuration at 2048^2. usage: r04_full_ab.sh <tag> <lib> ...
TAG=$1; shift
R=$GRAFT_REPO_ROOT/climaseaice.jl_amd
mkdir -p gpurun_out; : > gpurun_out/${TAG}.log
for round in 1 2; do
  for lib in "$@"; do
    ( if [ "$lib" != default ]; then export CSI_HIP_LIBRARY=$R/libcsi_hip_$lib.so; fi
      echo "$lib $(timeout 300 python scripts/bench_cases.py 2048 twelve level2 2>/dev/null | head -1)" >> gpurun_out/${TAG}.log )
  done
done
cat gpurun_out/${TAG}.log
