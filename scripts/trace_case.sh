#!/bin/bash
# Run on the GPU box: kernel trace of one scripts/bench_cases.py configuration. usage: scripts/trace_case.sh <tag> "<config substring>" [N]
TAG=$1; CASE=$2; N=${3:-2048}
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_case_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/bench_cases.py $N "$CASE" > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt | cut -c1-300
