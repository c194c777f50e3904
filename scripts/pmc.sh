#!/bin/bash
# Usage (through gpurun): scripts/pmc.sh <tag> "<bench args>" "<counters pass 1>" "<counters pass 2>" ...
set -u
TAG=$1; shift; BARGS=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
n=0
for C in "$@"; do
  n=$((n+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$n -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-full-step $BARGS > $OUT/p$n.log 2>&1
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(dict)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    tmp=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        tmp[(r["Kernel_Name"][:48],r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k,c),v in tmp.items(): agg[k][c]=sum(v)/len(v)
for k,v in agg.items():
    if "csi::" not in k or "fill_halo" in k or "k_init" in k: continue
    print(k)
    for c,x in sorted(v.items()): print("    %-28s %.5g"%(c,x))
PY
