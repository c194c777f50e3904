#!/bin/bash
# Run on the GPU box: self-connected 1024x512 tile (the N = 8 strong-scaling shape) over exchange intervals / halos.
TAG=$1; shift
for spec in "$@"; do
  label=${spec%%:*}; args=${spec#*:}
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --tile 1024x512 --force-connected $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$label', round(d['value']/1e9,2), 'G', 'ms_per_subcycle', round(d['ms_per_step'],3), 'phases', r.get('all_phases_ms'), 'k1', d.get('exchange_every_substep',{}).get('value'))" >> gpurun_out/${TAG}.log
done
cat gpurun_out/${TAG}.log
