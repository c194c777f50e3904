#!/bin/bash
# interleaved A/B of library builds over rows per tile on a tile-sized grid: scripts/rows_ab.sh <tag> <NXxNY> "<rows list>" <lib> ...
TAG=$1; TILE=$2; ROWS=$3; shift; shift; shift
R=$GRAFT_REPO_ROOT/climaseaice.jl_amd
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/${TAG}.log
for round in 1 2; do
 for rows in $ROWS; do
  for lib in "$@"; do
    ( if [ "$lib" != default ]; then export CSI_HIP_LIBRARY=$R/libcsi_hip_$lib.so; fi
      CSI_PAIR_ROWS=$rows timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --tile $TILE 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('rows', $rows, '$lib', round(d['value']/1e9,2), 'launch_us', round(d['roofline']['avg_launch_ms']*1e3,2))" >> gpurun_out/${TAG}.log )
  done
 done
done
cat gpurun_out/${TAG}.log
