#!/bin/bash
# A/B builds of the pair kernel: scripts/build_variant.sh <name> "<extra hipcc flags>" -> climaseaice.jl_amd/libcsi_hip_<name>.so
# (select at run time with CSI_HIP_LIBRARY=...).  Rebuilds the default library afterwards.
set -e
cd /root/repo/climaseaice.jl_amd/csrc
touch evp_fused2.hip
make -s -j8 PROBE="$2" 2>&1 | grep -E "error" | head -5 || true
cp ../libcsi_hip.so ../libcsi_hip_$1.so
touch evp_fused2.hip
make -s -j8 2>&1 | grep -E "error" | head -5 || true
ls -la ../libcsi_hip_$1.so
