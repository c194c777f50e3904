cd $GRAFT_REPO_ROOT
for r in 1 2; do
for spec in "plain:0" "peerkernel:1"; do
  label=${spec%%:*}; v=${spec#*:}
  for t in 1024x512 2048x2048; do
  CSI_PEER_KERNEL=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --tile $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$label $t', round(d['value']/1e9,2), round(r['avg_launch_ms']*1e3,1))"
  done
done
done
