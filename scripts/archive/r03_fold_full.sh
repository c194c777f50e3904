#!/bin/bash
# fold band: tests, fuzz (untiled and decompositions)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_evp.py tests/test_gpu_local_tiles.py tests/test_gpu_steps.py -m gpu -q -k "fold or local or tripolar or curvilinear" > gpurun_out/fold_tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED" gpurun_out/fold_tests.log | head
python scripts/fuzz_fold.py 0 300 > gpurun_out/fold_fuzz.log 2>&1
grep -E "FAIL|done" gpurun_out/fold_fuzz.log | cut -c1-500 | head
timeout 2400 python scripts/fuzz_local_tiles.py 300 600 > gpurun_out/local_fuzz.log 2>&1
grep -E "FAIL|done" gpurun_out/local_fuzz.log | cut -c1-500 | head -20
