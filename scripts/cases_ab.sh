#!/bin/bash
# Round 4: A/B of library builds on selected bench_cases configurations. usage: cases_ab.sh <tag> "<case substrings ...>" <lib> ...
TAG=$1; CASES=$2; shift; shift
R=$GRAFT_REPO_ROOT/climaseaice.jl_amd
mkdir -p gpurun_out; : > gpurun_out/${TAG}.log
for cs in $CASES; do
  for lib in "$@"; do
    ( if [ "$lib" != default ]; then export CSI_HIP_LIBRARY=$R/libcsi_hip_$lib.so; fi
      echo "$lib $(timeout 300 python scripts/bench_cases.py 2048 $cs level2 2>/dev/null | head -1)" >> gpurun_out/${TAG}.log )
  done
done
cat gpurun_out/${TAG}.log
