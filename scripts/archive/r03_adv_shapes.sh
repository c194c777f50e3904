#!/bin/bash
# block shape of the tendency kernel (TX x TY flux tile): config 2 (512^2, one launch per stage) and 1024^2 / 2048^2
cd $GRAFT_REPO_ROOT
for lib in default 6_64 7_56 8_48 10_40 12_32 5_64 6_60; do
  L=$GRAFT_REPO_ROOT/climaseaice.jl_amd/libcsi_hip_adv_$lib.so; [ $lib = default ] && L=$GRAFT_REPO_ROOT/climaseaice.jl_amd/libcsi_hip.so
  for n in 512 1024 2048; do
    echo -n "TY_TX=$lib N=$n: "; CSI_HIP_LIBRARY=$L python3 examples/advection_only.py $n 200 2>/dev/null | head -1 | cut -c1-75
  done
done
