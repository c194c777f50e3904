#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + HBM traffic counters for bench.py.
# Usage: scripts/profile_bench.sh <round-tag>   -> writes gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
ARGS="$REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
