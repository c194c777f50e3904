#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_local_tiles.py -m gpu -q > gpurun_out/local_tiles.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|AssertionError: \(|Error" gpurun_out/local_tiles.log | cut -c1-300 | head -30
