#!/bin/bash
# GPU box: scripts/bench_cases.py lines (level 2) of a few configurations with the round-5 build and this build (cuts off / on), alternating
R=$GRAFT_REPO_ROOT/climaseaice.jl_amd
for rep in 1 2; do
for pat in "tripolar-like" "curvilinear channel (twelve" "north fold on uniform" "masked channel"; do
  CSI_HIP_LIBRARY=$R/libcsi_hip_r05.so python scripts/bench_cases.py 2048 "$pat" level2 2>&1 | grep level2 | head -1 | sed "s/^/r05      /"
  CSI_TILE_SKIPPING=0 CSI_ROW_CONSTANT=0 python scripts/bench_cases.py 2048 "$pat" level2 2>&1 | grep level2 | head -1 | sed "s/^/r06 off  /"
  python scripts/bench_cases.py 2048 "$pat" level2 2>&1 | grep level2 | head -1 | sed "s/^/r06 on   /"
done
done
