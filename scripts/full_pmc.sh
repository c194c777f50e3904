#!/bin/bash
# per-point-metric (CSI_METRIC_FULL) instantiation of the pair kernel: duration, HBM bytes and issue counters per launch
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/full_pmc; rm -rf $OUT; mkdir -p $OUT
ARGS="$REPO/scripts/bench_cases.py 2048 ${1:-twelve} level2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/mem -- python3 $ARGS > $OUT/mem.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:3]:
    print(r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3)
for d in ("fetch", "write", "sq", "mem"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % d):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_pair" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in sorted(agg.items()):
            print(d, c, sum(v) / len(v), len(v))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -delete
