#!/bin/bash
# Run on the GPU box: interleaved A/B of library builds on a tile-sized grid. usage: scripts/ab_tile_libs.sh <tag> <NXxNY> <lib> ...
TAG=$1; TILE=$2; shift; shift
R=$GRAFT_REPO_ROOT/climaseaice.jl_amd
for round in 1 2 3; do
  for lib in "$@"; do
    ( if [ "$lib" != default ]; then export CSI_HIP_LIBRARY=$R/libcsi_hip_$lib.so; fi
      timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --tile $TILE 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$lib', round(d['value']/1e9,2))" >> gpurun_out/${TAG}.log )
  done
done
cat gpurun_out/${TAG}.log
