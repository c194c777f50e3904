#!/bin/bash
# GPU box: timing experiment for "forcing values through the ring on per-point metrics" (wrong results in the nocl builds): default | consumer's forcing
# loads left out at four workgroups per CU (nocl: round 5's bound) | left out with 13 KB more LDS = three per CU (nocl_pad: what a ring would run at) |
# loads kept, three per CU (pad: what the occupancy alone costs)
R=$GRAFT_REPO_ROOT/climaseaice.jl_amd
for rep in 1 2; do
  for spec in "tripolar_like 2048 on" "tripolar 2048 on" "tripolar_like 2048 off"; do
    for lib in default nocl nocl_pad pad; do
      if [ $lib = default ]; then unset CSI_HIP_LIBRARY; else export CSI_HIP_LIBRARY=$R/libcsi_hip_$lib.so; fi
      python3 scripts/run_case.py $spec 2>&1 | tail -1 | sed "s/^/$lib   /"
    done
  done
done
