#!/bin/bash
# flag-protocol experiments (CSI_PEER_EXP builds, libcsi_exp<bits>.so): self-connected tiles on one GPU
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for lib in hip exp1 exp3 exp4 exp7; do
  for t in 1024x512 2048x2048; do
  CSI_HIP_LIBRARY=$GRAFT_REPO_ROOT/climaseaice.jl_amd/libcsi_$lib.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --tile $t --force-connected --halo 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$lib $t', round(d['value']/1e9,2), round(r['avg_launch_ms']*1e3,1), d['path']['halo_transport'])"
  done
done
done
