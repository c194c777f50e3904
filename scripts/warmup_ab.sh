#!/bin/bash
# does the headline depend on how long the device has been busy before the timed region?  scripts/warmup_ab.sh <tag>
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/$1.log
for round in 1 2 3; do
  for w in 5 60 200; do
    python bench.py --steps 20 --warmup $w --no-cpu-baseline --no-full-step --no-unfused 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('warmup', $w, round(d['value']/1e9,2), 'ms/step', round(d['ms_per_step'],3), 'launch_us', round(d['roofline']['avg_launch_ms']*1e3,1))" >> gpurun_out/$1.log
  done
done
cat gpurun_out/$1.log
