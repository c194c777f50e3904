#!/bin/bash
# kernel trace of the self-connected 1024 x 512 tile on both halo transports: scripts/r03_tile_trace.sh
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
for tr in peer rccl; do
  OUT=$REPO/gpurun_out/tile_trace_$tr; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-full-step --tile 1024x512 --force-connected --no-compare --transport $tr > $OUT/log.txt 2>&1
  python3 - <<PY
import csv, glob, json
f = glob.glob("$OUT/*/*_kernel_stats.csv")[0]
line = [l for l in open("$OUT/log.txt", errors="replace").read().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
print("## transport $tr:", round(d["value"] / 1e9, 2), "G cell-updates/s,", round(d["ms_per_step"], 3), "ms per 120 sub-steps,", d["path"])
print("| kernel | calls | avg us | total ms |\n|---|---|---|---|")
for r in list(csv.DictReader(open(f)))[:9]:
    print(f"| {r['Name'][:100]} | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['TotalDurationNs'])/1e6:.3f} |")
PY
done
