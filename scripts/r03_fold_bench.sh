#!/bin/bash
# fold band: throughput of the fold configurations (2048^2 and 1024^2), then the whole GPU suite
cd $GRAFT_REPO_ROOT
python scripts/bench_cases.py 2048 fold > gpurun_out/fold_bench.log 2>&1
python scripts/bench_cases.py 1024 fold >> gpurun_out/fold_bench.log 2>&1
grep -v "^{" gpurun_out/fold_bench.log | grep -v "version\|Hostname\|Librccl"
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED" gpurun_out/gpu_tests.log | head
