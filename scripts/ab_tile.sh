#!/bin/bash
# Run on the GPU box: sweep env knobs on a tile-sized grid. usage: scripts/ab_tile.sh <tag> <NXxNY> "<label>:<ENV=V ...>" ...
TAG=$1; TILE=$2; shift; shift
for round in 1 2; do
  for spec in "$@"; do
    label=${spec%%:*}; envs=${spec#*:}
    ( for e in $envs; do export $e; done
      timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --tile $TILE 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$label', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms']*1e3,1))" >> gpurun_out/${TAG}.log )
  done
done
cat gpurun_out/${TAG}.log
