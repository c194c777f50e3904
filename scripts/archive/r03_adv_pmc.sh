#!/bin/bash
# the WENO7 tendency kernel (FAST) under rocprofv3: duration, HBM bytes, issue counters.  2048^2 (separate launches) and 512^2 (one launch per RK stage)
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
for spec in "2048 60 0" "512 200 2"; do
  set -- $spec
  OUT=$REPO/gpurun_out/adv_pmc_$1; rm -rf $OUT; mkdir -p $OUT
  ARGS="$REPO/examples/advection_only.py $1 $2 $3"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/mem -- python3 $ARGS > $OUT/mem.log 2>&1
  python3 - <<PY
import csv, glob, collections
print("## N = $1, fusion $3")
f = glob.glob("$OUT/trace/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:3]:
    print(r['Name'][:90], r['Calls'], round(float(r['AverageNs'])/1e3, 2))
for d in ("fetch", "write", "sq", "mem"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % d):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_tendencies" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in sorted(agg.items()):
            print(d, c, sum(v) / len(v), len(v))
PY
  find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -delete
done
