#!/bin/bash
# round-3 closing fuzz campaign on the final kernels: fused paths (with wind drag / bottom-stress arrays), decompositions with
# distinct tiles, the north fold, self-connected tiles, whole steps
cd $GRAFT_REPO_ROOT
python scripts/fuzz_fused.py 40000 41500 > gpurun_out/big_fused_fuzz.log 2>&1
grep -E "FAIL|ERR|done" gpurun_out/big_fused_fuzz.log | cut -c1-500 | head
timeout 2400 python scripts/fuzz_local_tiles.py 2100 3100 > gpurun_out/big_local_fuzz.log 2>&1
grep -E "FAIL|done" gpurun_out/big_local_fuzz.log | cut -c1-500 | head -20
python scripts/fuzz_fold.py 1100 1600 > gpurun_out/big_fold_fuzz.log 2>&1
grep -E "FAIL|done" gpurun_out/big_fold_fuzz.log | cut -c1-500 | head
python scripts/fuzz_tiles.py 1400 1900 > gpurun_out/big_tile_fuzz.log 2>&1
grep -E "FAIL|done" gpurun_out/big_tile_fuzz.log | cut -c1-500 | head
python scripts/fuzz_steps.py 1160 1400 > gpurun_out/big_steps_fuzz.log 2>&1
grep -E "FAIL|ERR|done" gpurun_out/big_steps_fuzz.log | cut -c1-500 | head
