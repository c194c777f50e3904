#!/bin/bash
# GPU box: kernel traces of the tripolar case's fold band -- beside the pair launches (product), alone (CSI_EXP_BAND_ONLY=1: wrong results,
# timing only) and on reserved CUs (CSI_BAND_CUS=4).  usage: scripts/prof_band.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/band_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in product alone cus4; do
  case $v in product) export -n CSI_EXP_BAND_ONLY CSI_BAND_CUS; unset CSI_EXP_BAND_ONLY CSI_BAND_CUS;; alone) export CSI_EXP_BAND_ONLY=1;; cus4) unset CSI_EXP_BAND_ONLY; export CSI_BAND_CUS=4;; esac
  mkdir -p $OUT/$v
  python3 $R/scripts/run_case.py tripolar_land 2048 on 5 > $OUT/$v/plain.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v/trace -- python3 $R/scripts/run_case.py tripolar_land 2048 on 3 > $OUT/$v/trace.txt 2>&1
  python3 $R/scripts/summarize_structure_profile.py $OUT/$v > $OUT/$v/summary.txt 2>&1
  echo "== $v"; tail -1 $OUT/$v/plain.txt; head -14 $OUT/$v/summary.txt
  rm -rf $OUT/$v/trace
done
