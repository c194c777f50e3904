#!/bin/bash
# Run on the GPU box: cell-updates/s of the sub-cycle over grid sizes (periodic f-plane, 120 sub-steps).
TAG=$1; shift
for sz in "$@"; do
  timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-step --tile $sz 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$sz', round(d['value']/1e9,2), 'G', round(d['roofline']['avg_launch_ms']*1e3,1), 'us/launch', round(d['roofline']['achieved']), 'GB/s')" >> gpurun_out/${TAG}.log
done
cat gpurun_out/${TAG}.log
