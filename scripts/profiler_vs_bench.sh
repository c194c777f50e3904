#!/bin/bash
# ONE lease: why does the dominant kernel take ~12 % longer under rocprofv3 --kernel-trace than in the plain run (round 5: 131.5 vs 117.1 us)?
# The same long bench command plain and under the profiler, the shader clock and board power of the busy card sampled DURING both,
# and the trace's own start / end timestamps (durations AND the gaps between consecutive launches).
# usage (gpurun): scripts/profiler_vs_bench.sh <tag>   -> gpurun_out/pvb_<tag>/summary.txt
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pvb_$TAG
rm -rf $OUT; mkdir -p $OUT
sample() {
  for d in /sys/class/drm/card*/device; do
    f=$(cat $d/hwmon/hwmon*/freq1_input 2>/dev/null | head -1); p=$(cat $d/hwmon/hwmon*/power1_average 2>/dev/null | head -1)
    [ -z "$p" ] && p=$(cat $d/hwmon/hwmon*/power1_input 2>/dev/null | head -1)
    [ -n "$f" ] && echo -n "$(basename $(dirname $d)):$((f / 1000000)):$((p / 1000000)) "
  done
  echo
}
watch() {   # $1: pid, $2: file
  while kill -0 $1 2>/dev/null; do sample >> $2; sleep 0.1; done
}
ARGS="$R/bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-full-step --no-unfused --no-structure"
cd /tmp && export TMPDIR=/tmp
sample > $OUT/idle.txt
for rep in 1 2; do
  python3 $ARGS > $OUT/plain$rep.json 2> /dev/null &
  pid=$!; watch $pid $OUT/plain$rep.clock; wait $pid
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace$rep -- python3 $ARGS > $OUT/traced$rep.json 2> $OUT/traced$rep.err &
  pid=$!; watch $pid $OUT/traced$rep.clock; wait $pid
done
# ... and the SHORT command round 5's profile passes used (3 steps in all: 20 ms of GPU work from an idle chip), plain and traced
SHORT="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-step --no-unfused --no-structure"
for rep in 1 2 3; do
  python3 $SHORT > $OUT/short_plain$rep.json 2> /dev/null
  rocprofv3 --kernel-trace --output-format csv -d $OUT/short_trace$rep -- python3 $SHORT > $OUT/short_traced$rep.json 2> /dev/null
done
python3 $R/scripts/summarize_profiler_vs_bench.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*_agent_info.csv" -delete
# the traces are large: keep the first 3000 rows of one
for f in $(find $OUT -name "*kernel_trace.csv"); do head -3000 $f > $f.head; rm -f $f; done
cat $OUT/summary.txt
