#!/bin/bash
# GPU box: scripts/run_case.py lines of some cases with two builds, alternating.  usage: scripts/ab_two_libs.sh <libsuffix> "<case N mode>" ...
R=$GRAFT_REPO_ROOT/climaseaice.jl_amd
SUF=$1; shift
for rep in 1 2; do
  for spec in "$@"; do
    CSI_HIP_LIBRARY=$R/libcsi_hip_$SUF.so python3 scripts/run_case.py $spec 2>&1 | tail -1 | sed "s/^/$SUF   /"
    python3 scripts/run_case.py $spec 2>&1 | tail -1 | sed "s/^/default  /"
  done
done
