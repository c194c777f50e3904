#!/bin/bash
# one launch per RK stage (advection-only models): tests, then config 2's time per step and kernel trace
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_steps.py -m gpu -q -k "advection" > gpurun_out/adv_stage_tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|AssertionError" gpurun_out/adv_stage_tests.log | cut -c1-250 | head
bash scripts/r03_adv_trace.sh stage 512
bash scripts/r03_adv_trace.sh stage2048 2048 | head -5
