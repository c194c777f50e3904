#!/bin/bash
# round-3 start: GPU suite + the numbers the round starts from (one lease)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_tests.log
tail -3 gpurun_out/r03_tests.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r03_bench0.json 2> gpurun_out/r03_bench0.err; tail -c 600 gpurun_out/r03_bench0.json
rm -f gpurun_out/r03_tile0.log gpurun_out/r03_conn0.log
bash scripts/ab_tile.sh r03_tile0 1024x512 "base:X=1" > /dev/null
bash scripts/ab_conn.sh r03_conn0 "h32k16:--halo 32" "h16k8:--halo 16" "h4k2:--halo 4" > /dev/null
cat gpurun_out/r03_tile0.log gpurun_out/r03_conn0.log
