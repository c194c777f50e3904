#!/bin/bash
# Run on the GPU box: grid-size sweep for several library builds. usage: scripts/size_sweep_libs.sh <tag> "<lib> ..." <size> ...
TAG=$1; LIBS=$2; shift; shift
R=$GRAFT_REPO_ROOT/climaseaice.jl_amd
for sz in "$@"; do
  for rep in 1 2; do
  for lib in $LIBS; do
    ( if [ "$lib" != default ]; then export CSI_HIP_LIBRARY=$R/libcsi_hip_$lib.so; fi
      timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-step --tile $sz 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$sz', '$lib', round(d['value']/1e9,2))" >> gpurun_out/${TAG}.log )
  done
  done
done
cat gpurun_out/${TAG}.log
