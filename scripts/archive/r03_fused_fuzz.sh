#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/fuzz_fused.py 40000 42000 > gpurun_out/big_fused_fuzz.log 2>&1
grep -E "FAIL|ERR|done" gpurun_out/big_fused_fuzz.log | cut -c1-500 | head
