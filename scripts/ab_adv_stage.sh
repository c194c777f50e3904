#!/bin/bash
# GPU box: advection-only RK3 steps above the 3 M-cell cut with one launch per stage (fused) and with separate tendency / update launches
for rep in 1 2; do
  for N in 1536 2048 3072 4096; do
    CSI_ADV_STAGE_MAX_CELLS=0 python3 scripts/adv_bench.py $N 2>&1 | grep "^$N" | sed "s/^/separate /"
    CSI_ADV_STAGE_MAX_CELLS=100000000 python3 scripts/adv_bench.py $N 2>&1 | grep "^$N" | sed "s/^/fused    /"
  done
done
