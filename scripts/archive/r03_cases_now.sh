#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/bench_cases.py 2048 2>/dev/null | grep -v "^{" | grep -v "version\|Hostname\|Librccl\|amdgpu.ids" > gpurun_out/r03_bench_cases_final.txt
cat gpurun_out/r03_bench_cases_final.txt | cut -c1-200
