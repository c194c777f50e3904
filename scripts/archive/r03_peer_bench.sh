#!/bin/bash
# self-connected tiles on one GPU: peer transport vs RCCL exchange vs untiled (scripts/r03_peer_bench.sh <tag>)
cd $GRAFT_REPO_ROOT
TAG=${1:-r03_peer}
run() { label=$1; shift
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step "$@" 2>gpurun_out/${TAG}_$label.err | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; p=d['path']
print('$label', round(d['value']/1e9,2), 'G  ms/subcycle', round(d['ms_per_step'],3), 'launch_us', round(r['avg_launch_ms']*1e3,1), p.get('halo_transport'), 'k', p['exchange_interval'], 'rccl16', (d.get('rccl_exchange') or {}).get('value'), 'k1', (d.get('exchange_every_substep') or {}).get('value'))" >> gpurun_out/${TAG}.log
}
rm -f gpurun_out/${TAG}.log
run full2048
run tile_untiled --tile 1024x512
run tile_peer --tile 1024x512 --force-connected
run tile_rccl --tile 1024x512 --force-connected --transport rccl
run t1024_untiled --tile 1024x1024
run t1024_peer --tile 1024x1024 --force-connected
run t2048x1024_untiled --tile 2048x1024
run t2048x1024_peer --tile 2048x1024 --force-connected
run full2048_peer --tile 2048x2048 --force-connected
cat gpurun_out/${TAG}.log
