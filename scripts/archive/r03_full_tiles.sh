#!/bin/bash
# tile count of the per-point-metric instantiation at 2048^2 (two waves per SIMD: 1024 resident tiles)
cd $GRAFT_REPO_ROOT
for tiles in default 768 896 960 1024 1100 1280; do
  T=$tiles; [ $tiles = default ] && T=""
  echo -n "tiles=$tiles: "; CSI_PAIR_TILES=$T python scripts/bench_cases.py 2048 "curvilinear channel (twelve" level2 2>/dev/null | grep -o "level2': [0-9.]*"
done
