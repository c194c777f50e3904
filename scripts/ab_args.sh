#!/bin/bash
# Run on the GPU box: interleaved A/B of bench.py argument sets (3 rounds). usage: scripts/ab_args.sh <tag> "<label>:<args>" ...
TAG=$1; shift
for round in 1 2 3; do
  for spec in "$@"; do
    label=${spec%%:*}; args=${spec#*:}
    timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$label', round(d['value']/1e9,2))" >> gpurun_out/${TAG}.log
  done
done
cat gpurun_out/${TAG}.log
