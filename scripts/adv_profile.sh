#!/bin/bash
# Advection kernels under rocprofv3 (run on the GPU box through gpurun): kernel trace + separate PMC passes of scripts/adv_bench.py at
# 512^2 (BASELINE config 2: one launch per RK stage) and 2048^2 (separate tendency / update launches).  -> gpurun_out/prof_adv_<tag>/
# then, here: python scripts/summarize_adv_profile.py <tag>
set -u
TAG=${1:-r05}
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/prof_adv_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $REPO
python3 scripts/adv_bench.py 256 512 1024 2048 > $OUT/plain.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for N in 512 2048; do
  A="$REPO/scripts/adv_bench.py $N"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$N -- python3 $A > $OUT/trace_$N.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$N -- python3 $A > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$N -- python3 $A > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT/pmc_sq_$N -- python3 $A > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $OUT/pmc_mix_$N -- python3 $A > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, json, os
out = {}
for N in (512, 2048):
    for f in glob.glob("$OUT/pmc_*_%d/*/*_counter_collection.csv" % N):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            out.setdefault(str(N), {}).setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
        os.remove(f)
    for f in glob.glob("$OUT/trace_%d/*/*_kernel_stats.csv" % N):
        out.setdefault(str(N), {})["_stats"] = list(csv.DictReader(open(f)))
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
PY
find $OUT -name "*.db" -delete; find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*_kernel_trace.csv" -delete
du -sh $OUT
