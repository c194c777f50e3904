#!/bin/bash
# Round 4: does the fold band run BESIDE the pair launch when that launch leaves wave slots free?  (tile count of the pair launch)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/r04_fold_tiles.txt; : > $out
for t in default 1408 1280 1152 1024 896; do
  T=$t; [ $t = default ] && T=""
  echo "tiles=$t $(CSI_PAIR_TILES=$T timeout 300 python scripts/bench_cases.py 2048 north level2 2>/dev/null | head -1)" >> $out
done
for t in default 896 768 640; do
  T=$t; [ $t = default ] && T=""
  echo "tiles=$t $(CSI_PAIR_TILES=$T timeout 300 python scripts/bench_cases.py 2048 tripolar level2 2>/dev/null | head -1)" >> $out
done
cat $out
