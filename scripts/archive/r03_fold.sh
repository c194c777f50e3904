#!/bin/bash
# north fold band: dedicated tests + the fused-path tests of the fold cases
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_evp.py -m gpu -q -k "north_fold_band or (fused_kernels_bitwise and (folded or tripolar))" > gpurun_out/fold_tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^E  |^FAILED" gpurun_out/fold_tests.log | head -20
