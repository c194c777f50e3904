#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "decomposition" > gpurun_out/local_full.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|AssertionError|Error" gpurun_out/local_full.log | cut -c1-300 | head -20
