#!/bin/bash
# fold band: fuzz, then the whole GPU suite
cd $GRAFT_REPO_ROOT
python scripts/fuzz_fold.py 0 400 > gpurun_out/fold_fuzz.log 2>&1
grep -E "FAIL|done" gpurun_out/fold_fuzz.log | cut -c1-600 | head
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED" gpurun_out/gpu_tests.log | head
