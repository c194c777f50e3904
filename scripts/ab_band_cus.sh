#!/bin/bash
# GPU box: the fold band on reserved CUs (CSI_BAND_CUS = r CUs per XCD for the band's launches, the pair launches beside them masked to
# the others; CSI_BAND_CUS_SHARE=1: the band may use every CU).  usage: scripts/ab_band_cus.sh "<case N>" ...   (cuts on, 120 sub-steps)
for rep in 1 2; do
  for spec in "$@"; do
    python3 scripts/run_case.py $spec on 5 2>&1 | tail -1 | sed "s/^/r=0        /"
    for r in 1 2 3 4 6; do
      CSI_BAND_CUS=$r python3 scripts/run_case.py $spec on 5 2>&1 | tail -1 | sed "s/^/r=$r        /"
      CSI_BAND_CUS=$r CSI_BAND_CUS_SHARE=1 python3 scripts/run_case.py $spec on 5 2>&1 | tail -1 | sed "s/^/r=$r share  /"
    done
  done
done
