#!/bin/bash
# Round 4: tile count of the per-point-metric instantiation after the plane prefetch (2 waves per SIMD: 1024 resident tiles)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/r04_full_tiles.txt; : > $out
for t in default 512 768 896 1024 1280 1536 2048; do
  T=$t; [ $t = default ] && T=""
  echo "tiles=$t $(CSI_PAIR_TILES=$T timeout 300 python scripts/bench_cases.py 2048 twelve level2 2>/dev/null | head -1)" >> $out
done
cat $out
