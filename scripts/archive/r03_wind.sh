#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_evp.py -m gpu -q -k "bottom_stress_arrays or wind_drag" > gpurun_out/wind_tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|AssertionError|Error" gpurun_out/wind_tests.log | cut -c1-300 | head -20
