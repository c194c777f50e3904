#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_local_tiles.py tests/test_gpu_evp.py -m gpu -q -k "local or peer" > gpurun_out/gpu_tests_dld.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED" gpurun_out/gpu_tests_dld.log | head
timeout 2400 python scripts/fuzz_local_tiles.py 0 300 > gpurun_out/local_fuzz.log 2>&1
grep -E "FAIL|done" gpurun_out/local_fuzz.log | cut -c1-700 | head -20
bash scripts/r03_ab.sh ab_dld4 "peer:--tile 1024x512 --force-connected --no-compare"
