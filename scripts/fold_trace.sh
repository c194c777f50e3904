#!/bin/bash
# kernel trace of the two north-fold configurations at 2048^2 (scripts/bench_cases.py): per-kernel averages and the timeline of one pair step
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
for which in uniform tripolar; do
  OUT=$REPO/gpurun_out/fold_trace_$which; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/scripts/bench_cases.py 2048 "$which" level2 > $OUT/log.txt 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*_kernel_stats.csv")[0]
print("## $which")
print(open("$OUT/log.txt", errors="replace").read().splitlines()[-2][:300])
print("| kernel | calls | avg us | total ms |\n|---|---|---|---|")
for r in list(csv.DictReader(open(f)))[:10]:
    print(f"| {r['Name'][:90]} | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['TotalDurationNs'])/1e6:.3f} |")
t = glob.glob("$OUT/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r["Start_Timestamp"]))
# one pair step from the middle of the run
k = [i for i, r in enumerate(rows) if "k_pair" in r["Kernel_Name"]]
i0 = k[len(k) // 2]; i1 = k[len(k) // 2 + 1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    print(f"  {(int(r['Start_Timestamp']) - t0)/1e3:8.1f} .. {(int(r['End_Timestamp']) - t0)/1e3:8.1f} us  {r['Kernel_Name'][:70]}")
PY
done
