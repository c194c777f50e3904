#!/bin/bash
# PC sampling of one bench_cases configuration (rocprofv3 beta feature): scripts/pc_sample.sh <tag> <case substring> [method] [interval]
# NOT available on this pool (round 4: `rocprofv3-avail list --pc-sampling` names no agent, rocprofv3 answers "configuration is not supported
# on any of the agents"); kept for a box that has it.  The in-kernel probes (pair_probe.py, pair_probe_case.py) stand in for it.
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
