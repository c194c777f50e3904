#!/bin/bash
# A/B of bench_cases.py configurations: ab_base/ (an earlier commit, built) against the working tree, interleaved on one box
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for cfg in "coupled periodic" "coupled channel" "masked channel" "OMIP"; do
    for dir in ab_base .; do
      echo -n "$dir | "; ( cd $dir && python scripts/bench_cases.py 2048 "$cfg" level2 2>/dev/null | grep -v "^{" | grep "level2" | cut -c1-110 )
    done
  done
done
