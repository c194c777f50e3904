# rocprofv3 kernel trace of the default bench (sub-cycle + full RK3 steps); summary written by scripts/summarize_db.py
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/final_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/final_prof -o final -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/final_prof.log 2>&1
tail -1 gpurun_out/final_prof.log | cut -c1-300
python3 scripts/summarize_db.py gpurun_out/final_prof/final_results.db
