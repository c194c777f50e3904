#!/bin/bash
# Run on the GPU box: kernel trace of the self-connected 1024x512 tile bench. usage: scripts/trace_conn.sh <tag> [bench args]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-step --tile 1024x512 --force-connected "$@" > $OUT/log.txt 2>&1
ls $OUT/*/ | head
