#!/bin/bash
# fold band: the fold tests, then throughput of the fold configurations (2048^2 and 1024^2)
cd $GRAFT_REPO_ROOT
bash scripts/r03_fold.sh
python scripts/bench_cases.py 2048 fold > gpurun_out/fold_bench.log 2>&1
python scripts/bench_cases.py 1024 fold >> gpurun_out/fold_bench.log 2>&1
grep -v "^{" gpurun_out/fold_bench.log | grep -v "version\|Hostname\|Librccl\|amdgpu.ids"
