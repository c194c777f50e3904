#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_local_tiles.py -m gpu -q > gpurun_out/local_tiles.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED" gpurun_out/local_tiles.log | head
timeout 2400 python scripts/fuzz_local_tiles.py 0 300 > gpurun_out/local_fuzz.log 2>&1
grep -E "FAIL|done" gpurun_out/local_fuzz.log | cut -c1-700 | head -20
