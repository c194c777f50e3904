#!/bin/bash
# what the driver runs at round end, in its order: the GPU suite, smoke(), the default bench line
cd $GRAFT_REPO_ROOT
python -m pytest tests/ -x -q -m gpu > gpurun_out/final_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/final_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; tail -c 600 gpurun_out/final_bench.json
