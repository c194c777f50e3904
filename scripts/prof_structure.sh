#!/bin/bash
# GPU box: rocprofv3 kernel trace (+ optional PMC passes) of one structure case in one mode.
# usage: scripts/prof_structure.sh <tag> <case> <N> <on|off> [pmc]
TAG=$1; CASE=$2; N=$3; MODE=$4; PMC=${5:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/scripts/run_case.py $CASE $N $MODE 3 > $OUT/plain.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/run_case.py $CASE $N $MODE 3 > $OUT/trace.txt 2>&1
if [ -n "$PMC" ]; then
  n=0
  for C in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    n=$((n+1))
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc$n -- python3 $R/scripts/run_case.py $CASE $N $MODE 1 > $OUT/pmc$n.txt 2>&1
  done
fi
python3 $R/scripts/summarize_structure_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/plain.txt | tail -1; cat $OUT/summary.txt | head -60
