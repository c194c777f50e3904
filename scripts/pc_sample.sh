#!/bin/bash
# PC sampling of one bench_cases configuration (rocprofv3 beta feature): scripts/pc_sample.sh <tag> <case substring> [method] [interval]
TAG=$1; CASE=$2; METHOD=${3:-host_trap}; INT=${4:-200}
UNIT=time; [ "$METHOD" = stochastic ] && UNIT=cycles
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=/tmp/pcs_$TAG; rm -rf $OUT; mkdir -p $OUT
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $INT \
    --kernel-trace --output-format csv -d $OUT -- python3 $REPO/scripts/bench_cases.py 2048 "$CASE" level2 > $OUT/run.log 2>&1
echo "rc=$?"; tail -5 $OUT/run.log
find $OUT -type f | head -20
mkdir -p $REPO/gpurun_out/pcs_$TAG
for f in $(find $OUT -name "*pc_sampling*csv"); do ls -la $f; head -3 $f; cp $f $REPO/gpurun_out/pcs_$TAG/ 2>/dev/null; done
