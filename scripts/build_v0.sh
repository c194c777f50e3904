#!/bin/bash
# A/B build of the PLAIN family of the pair kernel only (variant 0: the headline's object): scripts/build_v0.sh <name> "<extra hipcc flags>"
# -> climaseaice.jl_amd/libcsi_hip_<name>.so (every other object as built; select with CSI_HIP_LIBRARY=...).  ~1.5 min instead of a full rebuild.
set -e
cd /root/repo/climaseaice.jl_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical -Wno-unused-value -I../../include \
  -ffp-contract=off $2 -DCSI_PAIR_VARIANT=0 -c evp_fused2.hip -o /tmp/evp_fused2_$1.o -Rpass-analysis=kernel-resource-usage 2> /tmp/evp_fused2_$1.res
OBJS=$(grep '^OBJS' Makefile | sed 's/OBJS *:= *//' | sed "s#evp_fused2.o#/tmp/evp_fused2_$1.o#")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libcsi_hip_$1.so $OBJS -L/opt/rocm/lib -lrccl -lrt -Wl,-rpath,/opt/rocm/lib
ls -la ../libcsi_hip_$1.so | cut -c1-90
python3 - <<PY
import re
txt=open('/tmp/evp_fused2_$1.res').read()
for b in txt.split("remark: Function Name: ")[1:]:
    name=b.split()[0]
    m=re.search(r"k_pairI(.*?)EEv",name)
    if not m: continue
    fl=re.findall(r"L[bi](\d+)E",m.group(1))
    g=lambda k:int(re.search(k+r": (\d+)",b).group(1))
    if fl[0]=='1' and fl[1]=='0' and fl[8]=='0' and fl[6]=='2': print("headline instantiation: VGPRs",g("VGPRs"),"spill",g("VGPRs Spill"),"occupancy",g(r"Occupancy \[waves/SIMD\]"),"LDS",g(r"LDS Size \[bytes/block\]"))
PY
