#!/bin/bash
# A/B on one box: the tree under ab_base/ (an earlier commit, built) against the working tree.  scripts/r03_ab.sh <tag> "<label>:<bench args>" ...
cd $GRAFT_REPO_ROOT
TAG=$1; shift
rm -f gpurun_out/${TAG}.log
one() { dir=$1; label=$2; shift; shift
  ( cd $dir && timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step "$@" 2>/dev/null ) | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; p=d['path']
print('$label', round(d['value']/1e9,2), 'G  launch_us', round(r['avg_launch_ms']*1e3,1), p.get('halo_transport',''), 'k', p['exchange_interval'])" >> gpurun_out/${TAG}.log
}
for round in 1 2; do
  for spec in "$@"; do
    label=${spec%%:*}; args=${spec#*:}
    one ab_base base_$label $args
    one . new_$label $args
  done
done
cat gpurun_out/${TAG}.log
