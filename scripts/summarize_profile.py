"""Summarise gpurun_out/prof_<tag> (scripts/profile_bench.sh) into profiles/<name>.md + traffic json."""
import collections, csv, glob, json, sys, os
tag, name = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
base = f"gpurun_out/prof_{tag}"
cells = 2048 * 2048
algo = {"k_stress": 96, "k_ustep": 80, "k_vstep": 80}
out = [f"# rocprofv3 summary: {name}\n\n{note}\n\n",
       "command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-step`\n"
       "(2048x2048 periodic f-plane, 120 sub-steps, FAST mode; counters in separate `--pmc` passes)\n\n## kernel stats\n\n",
       "| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n"]
ks = glob.glob(base + "/trace/*/*_kernel_stats.csv")[0]
avg = {}
for r in csv.DictReader(open(ks)):
    out.append(f"| {r['Name'][:80]} | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['Percentage']):.2f} |\n")
    for a in algo:
        if "fast::" + a in r["Name"]:
            avg[a] = float(r["AverageNs"]) / 1e3
agg = collections.defaultdict(dict)
for f in glob.glob(base + "/pmc_*/*/*_counter_collection.csv"):
    tmp = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        tmp[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in tmp.items():
        agg[k][c] = sum(v) / len(v)
out.append("\n## HBM-side traffic per launch (FETCH_SIZE, WRITE_SIZE in KB; gfx950: FETCH_SIZE reports 1/2 of the bytes read)\n\n")
out.append("| kernel | FETCH_SIZE | WRITE_SIZE | corrected traffic MB | algorithmic MB | ratio | avg us | algorithmic GB/s | traffic GB/s |\n|---|---|---|---|---|---|---|---|---|\n")
traffic = {}
for k, v in agg.items():
    kk = [a for a in algo if "fast::" + a in k]
    if not kk or "FETCH_SIZE" not in v:
        continue
    kk = kk[0]
    tr = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    al = algo[kk] * cells
    traffic[kk.replace("k_", "")] = tr
    out.append(f"| {kk} | {v['FETCH_SIZE']:.0f} | {v['WRITE_SIZE']:.0f} | {tr/1e6:.1f} | {al/1e6:.1f} | {tr/al:.2f} | {avg.get(kk, 0):.1f} | {al/avg[kk]/1e3:.0f} | {tr/avg[kk]/1e3:.0f} |\n")
out.append("\n## other counters (mean per launch)\n\n| kernel | counter | value |\n|---|---|---|\n")
for k, v in agg.items():
    if "csi::fast::k_" not in k or "k_init" in k:
        continue
    for c, x in sorted(v.items()):
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        out.append(f"| {k[:40]} | {c} | {x:.4g} |\n")
open(f"profiles/{name}.md", "w").write("".join(out))
json.dump({"source": f"profiles/{name}.md", "workload": "2048x2048 periodic f-plane, FAST", "bytes_per_launch": traffic},
          open("profiles/traffic_latest.json", "w"), indent=1)
print("".join(out))
