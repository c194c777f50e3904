#!/bin/bash
# in-process tile group: the local-tile tests, the peer / tile tests, then the headline bench (A/B of the per-side wrap flags)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_local_tiles.py -m gpu -q > gpurun_out/local_tiles.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|AssertionError: \(" gpurun_out/local_tiles.log | cut -c1-250 | head -30
python -m pytest tests/test_gpu_evp.py -m gpu -q -k "peer or tile or fused_kernels" > gpurun_out/local_peer.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED" gpurun_out/local_peer.log | head
for i in 1 2; do python bench.py --no-cpu-baseline --no-full-step 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('headline', round(d['value'] / 1e9, 2), d['ms_per_step'])"; done
python bench.py --no-cpu-baseline --no-full-step --tile 1024x512 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('tile untiled', round(d['value'] / 1e9, 2))"
python bench.py --no-cpu-baseline --no-full-step --tile 1024x512 --force-connected --no-compare 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('tile peer', round(d['value'] / 1e9, 2))"
