#!/bin/bash
# tuning aid: rows-per-wave sweep of the fused kernel (run through gpurun)
for r in "$@"; do
  CSI_FUSED_ROWS=$r python bench.py --no-cpu-baseline --no-full-step --steps 3 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rows', $r, 'Gcu/s', round(d['value']/1e9,2), 'subcycle ms', round(d['subcycle_ms_hip_events'],3))"
done
