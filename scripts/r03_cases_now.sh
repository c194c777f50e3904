#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/bench_cases.py 2048 "" level2 2>/dev/null | grep -v "^{" | grep -v "version\|Hostname\|Librccl\|amdgpu.ids"
