#!/bin/bash
# config 2 (512^2 advection only): time per RK3 step, and the kernel trace of the same program.  scripts/r03_adv_trace.sh <tag> [N]
TAG=${1:-adv}; N=${2:-512}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/adv_$TAG; rm -rf $OUT; mkdir -p $OUT
python3 examples/advection_only.py $N 400 2>/dev/null | tee $OUT/plain.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/examples/advection_only.py $N 50 > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.2f} total_ms {float(r['TotalDurationNs'])/1e6:8.3f}")
PY
