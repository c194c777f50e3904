#!/bin/bash
# round-3 closing fuzz campaign on the final kernels: decompositions with distinct tiles, the north fold, self-connected tiles
cd $GRAFT_REPO_ROOT
timeout 2400 python scripts/fuzz_local_tiles.py 600 2100 > gpurun_out/big_local_fuzz.log 2>&1
grep -E "FAIL|done" gpurun_out/big_local_fuzz.log | cut -c1-500 | head -20
python scripts/fuzz_fold.py 300 1100 > gpurun_out/big_fold_fuzz.log 2>&1
grep -E "FAIL|done" gpurun_out/big_fold_fuzz.log | cut -c1-500 | head
python scripts/fuzz_tiles.py 600 1400 > gpurun_out/big_tile_fuzz.log 2>&1
grep -E "FAIL|done" gpurun_out/big_tile_fuzz.log | cut -c1-500 | head
