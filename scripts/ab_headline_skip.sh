#!/bin/bash
# GPU box: the headline line with tile activity on (default) and off, alternating
for rep in 1 2 3 4; do
  for v in 1 0; do
    CSI_TILE_SKIPPING=$v python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-full-step --no-unfused --no-structure 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('skipping $v: %.2f G  %.3f ms/step  launch %.2f us' % (d['value']/1e9, d['ms_per_step'], 1e3*d['roofline']['avg_launch_ms']))"
  done
done
