#!/bin/bash
# Register / LDS / occupancy report of the pair kernel variants: scripts/kernel_resources.sh [variant ...] (default 0)
cd /root/repo/climaseaice.jl_amd/csrc
for v in ${@:-0}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -ffp-contract=off $EXTRA -DCSI_PAIR_VARIANT=$v -c evp_fused2.hip -o /tmp/f2_$v.o -Rpass-analysis=kernel-resource-usage 2>&1 \
   | grep -E "error|Function Name|VGPRs:|SGPRs:|Spill|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* //; s/ \[-Rpass.*//' | paste - - - - - - - | sed 's/_ZN3csi5fused6k_pair//; s/EEEvPKNS_10FusedTableEiiiii//' | cut -c1-220
done
