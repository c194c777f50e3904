#!/bin/bash
# flakiness soak: the threaded local-tile tests and the peer tests repeatedly, then the whole suite twice
cd $GRAFT_REPO_ROOT
fail=0
for i in $(seq 1 15); do
  python -m pytest tests/test_gpu_local_tiles.py -m gpu -q -x > gpurun_out/soak_local_$i.log 2>&1 || { fail=$((fail+1)); echo "local run $i failed"; tail -5 gpurun_out/soak_local_$i.log; }
done
echo "local-tile runs failed: $fail of 15"
for i in 1 2; do
  python -m pytest tests -m gpu -q > gpurun_out/soak_suite_$i.log 2>&1; echo "suite $i rc=$?"; grep -E "passed|failed" gpurun_out/soak_suite_$i.log | tail -1
done
