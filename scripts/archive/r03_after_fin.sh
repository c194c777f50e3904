#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_local_tiles.py tests/test_gpu_evp.py tests/test_gpu_fullsize.py tests/test_gpu_steps.py -m gpu -q -k "local or peer or tile or decomposition or config4 or config5" > gpurun_out/fin_tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED" gpurun_out/fin_tests.log | head
for i in 1 2 3; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --tile 1024x512 --force-connected --no-compare 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('tile peer', round(d['value'] / 1e9, 2), round(d['roofline']['avg_launch_ms'] * 1e3, 1), round(d['ms_per_step'], 3))"
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --tile 1024x512 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('tile untiled', round(d['value'] / 1e9, 2), round(d['roofline']['avg_launch_ms'] * 1e3, 1), round(d['ms_per_step'], 3))"
