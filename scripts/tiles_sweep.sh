#!/bin/bash
# tile-count sweep (CSI_PAIR_TILES) of one bench_cases configuration: scripts/tiles_sweep.sh <tag> <case substring> "<tile counts>"
TAG=$1; CASE=$2; shift; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/${TAG}.log
for round in 1 2; do
 for t in $1; do
  if [ "$t" = default ]; then unset CSI_PAIR_TILES; else export CSI_PAIR_TILES=$t; fi
  echo "tiles=$t $(timeout 300 python scripts/bench_cases.py 2048 "$CASE" level2 2>/dev/null | head -1)" >> gpurun_out/${TAG}.log
 done
done
cat gpurun_out/${TAG}.log
