#!/bin/bash
# ONE lease, one box: the plain bench.py line, the rocprofv3 kernel trace and the PMC passes of the same build, so that the
# roofline numbers bench.py prints and the profile under profiles/ describe the same machine.
# Usage (through gpurun): scripts/same_lease_profile.sh <tag> [extra bench args]  -> gpurun_out/prof_<tag>/...
# then, here: python scripts/summarize_profile.py <tag> <name> "note"
set -u
TAG=${1:-r03}; shift || true
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $REPO
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structure "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-step --no-structure $*"
# the kernel TRACE runs long enough to be warm (round 6, profiles/r06_profiler_vs_bench.md: the first 60 launches from an idle chip take 145 us,
# the next 60 128, the chip is at speed after ~150 -- round 5 traced 3 steps in all and read the cold start as a profiler slow-down)
TARGS="$REPO/bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-full-step --no-unfused --no-structure $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $TARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_tcc -- python3 $ARGS > $OUT/pmc_tcc.log 2>&1
# round 5: the DYNAMIC mix of the kernel's vector instructions by class (FP64 fma / mul / add, transcendental = v_rcp / v_rsq)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $OUT/pmc_mix -- python3 $ARGS > $OUT/pmc_mix.log 2>&1
# a second plain line at the end: the box did not drift while it was being profiled
cd $REPO
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-step --no-structure "$@" > $OUT/bench_after.json 2> /dev/null
# keep the merge small: the per-dispatch counter tables are large, their per-kernel means are all the summary needs
python3 - <<PY
import csv, glob, collections, json, os
out = {}
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        out.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
    os.remove(f)
json.dump(out, open("$OUT/counters_by_kernel.json", "w"), indent=1)
PY
find $OUT -name "*.db" -delete; find $OUT -name "*_agent_info.csv" -delete
du -sh $OUT
