#!/bin/bash
# GPU suite on the box: scripts/gpu_tests.sh [pytest args]; summary line on stdout, full log in gpurun_out/gpu_tests.log
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q "$@" > gpurun_out/gpu_tests.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -5
