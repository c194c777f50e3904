#!/bin/bash
# Run on the GPU box: interleaved A/B of library builds / env knobs at 2048^2 (3 rounds each).
# usage: scripts/ab_quick.sh <tag> "<label>:<lib>:<ENV=V ...>" ...
TAG=$1; shift
R=$GRAFT_REPO_ROOT/climaseaice.jl_amd
for round in 1 2 3; do
  for spec in "$@"; do
    label=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}; envs=${rest#*:}
    ( if [ "$lib" != default ]; then export CSI_HIP_LIBRARY=$R/libcsi_hip_$lib.so; fi
      for e in $envs; do export $e; done
      timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$label', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms']*1e3,1))" >> gpurun_out/${TAG}.log )
  done
done
cat gpurun_out/${TAG}.log
