#!/bin/bash
# Run on the GPU box: kernel trace of whole RK3 time steps (bench.py's model_days_per_hr leg). usage: scripts/trace_step.sh <tag>
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_step_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | cut -c1-300
