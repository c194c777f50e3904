cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in 512 1024 2048; do
  rm -rf gpurun_out/gap_$n
  rocprofv3 --kernel-trace --stats -d gpurun_out/gap_$n -o gap -- python3 bench.py --size $n --no-cpu-baseline --no-full-step --steps 20 --warmup 3 > gpurun_out/gap_$n.log 2>&1
  python3 - <<PY
import json,glob,csv
line=[l for l in open("gpurun_out/gap_$n.log") if l.startswith("{")][-1]
j=json.loads(line)
print("N=$n bench avg_launch_ms", j["roofline"]["avg_launch_ms"], "value", j["value"]/1e9)
for f in glob.glob("gpurun_out/gap_$n/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_pair" in r["Name"]:
            print("   rocprof k_pair avg ns", r["AverageNs"], "calls", r["Calls"])
PY
done
