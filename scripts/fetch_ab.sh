#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / duration of the k_pair launches of one bench_cases configuration, per library build:
# scripts/fetch_ab.sh <tag> <case substring> <lib> ...
TAG=$1; CASE=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
R=$REPO/climaseaice.jl_amd
: > $REPO/gpurun_out/${TAG}.txt
for lib in "$@"; do
  if [ "$lib" != default ]; then export CSI_HIP_LIBRARY=$R/libcsi_hip_$lib.so; else unset CSI_HIP_LIBRARY; fi
  OUT=/tmp/fab_$lib; rm -rf $OUT; mkdir -p $OUT
  ARGS="$REPO/scripts/bench_cases.py 2048 $CASE level2"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc -- python3 $ARGS > $OUT/tcc.log 2>&1
  python3 - >> $REPO/gpurun_out/${TAG}.txt <<PY
import csv, glob, collections
print("== $lib")
f = glob.glob("$OUT/trace/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:2]:
    print(r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3)
for d in ("fetch", "write", "tcc"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % d):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_pair" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in sorted(agg.items()):
            print(d, c, sum(v) / len(v), len(v))
PY
done
cat $REPO/gpurun_out/${TAG}.txt
