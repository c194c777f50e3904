#!/bin/bash
# round-3 measurement pass on one lease: GPU suite, same-lease profile of the headline, tile / transport table, other configurations, config 2
cd $GRAFT_REPO_ROOT
bash scripts/gpu_tests.sh > gpurun_out/r03_final_tests.txt 2>&1; cat gpurun_out/r03_final_tests.txt
bash scripts/same_lease_profile.sh r03b > gpurun_out/r03_final_profile.txt 2>&1; tail -2 gpurun_out/r03_final_profile.txt
bash scripts/r03_peer_bench.sh r03_peer_final > /dev/null 2>&1; cat gpurun_out/r03_peer_final.log
python scripts/bench_cases.py 2048 > gpurun_out/r03_bench_cases.txt 2>&1; grep -v "^{" gpurun_out/r03_bench_cases.txt | grep level
bash scripts/r03_adv_trace.sh final > gpurun_out/r03_adv_final.txt 2>&1; head -8 gpurun_out/r03_adv_final.txt
