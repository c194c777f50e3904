#!/bin/bash
# Closing fuzz campaign of a round: scripts/fuzz_all.sh <tag> <n> -> gpurun_out/<tag>_fuzz.txt (n seeds per generator, fresh seed ranges per tag)
TAG=${1:-r04}; N=${2:-400}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/${TAG}_fuzz.txt; : > $out
B=${3:-50000}
for f in fuzz_fused fuzz_tiles fuzz_fold fuzz_steps fuzz_local_tiles; do
  n=$N; [ $f = fuzz_steps ] && n=$((N / 4)); [ $f = fuzz_local_tiles ] && n=$((N / 2))
  echo "== $f $B .. $((B + n))" >> $out
  timeout 1500 python scripts/$f.py $B $((B + n)) 2>/dev/null | tail -6 >> $out
done
cat $out
