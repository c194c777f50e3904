#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/r04_peer_flags.txt; : > $out
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --no-unfused "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print(round(d['value'] / 1e9, 2), 'G', 'launch_us', round(r['avg_launch_ms'] * 1e3, 2), d['path'].get('halo_transport'))"; }
for rep in 1 2; do
echo "untiled      $(run --tile 1024x512)" >> $out
echo "peer fine    $(run --tile 1024x512 --force-connected --no-compare)" >> $out
echo "peer coarse  $(CSI_PEER_FLAGS_COARSE=1 run --tile 1024x512 --force-connected --no-compare)" >> $out
echo "peer kernel only $(CSI_PEER_KERNEL=1 run --tile 1024x512)" >> $out
done
cat $out
