"""Summarise gpurun_out/prof_<tag> (scripts/profile_bench.sh) into profiles/<name>.md + profiles/counters_latest.json
(the static PMC evidence bench.py quotes, labelled with its source, in roofline.traffic / roofline.valu_frac).

usage: python scripts/summarize_profile.py <tag> <name> ["note"]
"""
import collections, csv, glob, json, sys
tag, name = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
base = f"gpurun_out/prof_{tag}"
cells = 2048 * 2048
# kernel -> (key, algorithmic bytes per cell per launch [SURVEY.md 8d], the kernel's own minimum bytes per cell per launch)
KERNELS = {"fast::k_stress": ("stress", 96, 96), "fast::k_ustep": ("ustep", 80, 80), "fast::k_vstep": ("vstep", 80, 80),
           "fused::k_substep": ("substep", 256, 120), "fused::k_pair": ("pair", 512, 120)}


def classify(kname):
    for pat, v in KERNELS.items():
        if pat in kname:
            return v
    return None


out = [f"# rocprofv3 summary: {name}\n\n{note}\n\n",
       "command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-step`\n"
       "(2048x2048 periodic f-plane, 120 sub-steps, FAST mode; counters in separate `--pmc` passes)\n\n## kernel stats\n\n",
       "| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n"]
ks = glob.glob(base + "/trace/*/*_kernel_stats.csv")[0]
avg = collections.defaultdict(list)
for r in csv.DictReader(open(ks)):
    out.append(f"| {r['Name'][:96]} | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['Percentage']):.2f} |\n")
    c = classify(r["Name"])
    if c:
        avg[c[0]].append((float(r["AverageNs"]) / 1e3, int(r["Calls"])))
avg = {k: sum(a * n for a, n in v) / sum(n for _, n in v) for k, v in avg.items()}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(base + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        c = classify(r["Kernel_Name"])
        if c:
            agg[c[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
agg = {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in agg.items()}
out.append("\n## HBM-side traffic per launch (FETCH_SIZE, WRITE_SIZE in KB; gfx950: FETCH_SIZE reports 1/2 of the bytes read)\n\n")
out.append("| kernel | FETCH_SIZE | WRITE_SIZE | corrected traffic MB | algorithmic MB | kernel minimum MB | avg us | algorithmic GB/s | traffic GB/s |\n|---|---|---|---|---|---|---|---|---|\n")
traffic = {}
byname = {v[0]: v for v in KERNELS.values()}
for k, v in agg.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v or k not in avg:
        continue
    tr = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    al, mn = byname[k][1] * cells, byname[k][2] * cells
    traffic[k] = tr
    out.append(f"| {k} | {v['FETCH_SIZE']:.0f} | {v['WRITE_SIZE']:.0f} | {tr/1e6:.1f} | {al/1e6:.1f} | {mn/1e6:.1f} | {avg[k]:.1f} | {al/avg[k]/1e3:.0f} | {tr/avg[k]/1e3:.0f} |\n")
out.append("\n## other counters (mean per launch)\n\n| kernel | counter | value |\n|---|---|---|\n")
for k, v in agg.items():
    for c, x in sorted(v.items()):
        if c not in ("FETCH_SIZE", "WRITE_SIZE"):
            out.append(f"| {k} | {c} | {x:.4g} |\n")
open(f"profiles/{name}.md", "w").write("".join(out))
dom = max(traffic, key=lambda k: avg.get(k, 0.0) * 1.0) if traffic else None
if dom:
    v = agg[dom]
    ctr = {"source": f"profiles/{name}.md", "workload": "2048x2048 periodic f-plane, FAST", "kernel": dom,
           "hbm_bytes_per_launch": traffic[dom], "avg_launch_us": avg[dom]}
    if "SQ_INSTS_VALU" in v:
        ctr["valu_insts_per_launch"] = v["SQ_INSTS_VALU"]
    if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v:
        # SQ_ACTIVE_INST_VALU: quad-cycles summed over SIMDs; GRBM_GUI_ACTIVE: cycles summed over the 8 XCDs
        ctr["valu_busy_frac"] = v["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * v["GRBM_GUI_ACTIVE"] / 8.0)
    json.dump(ctr, open("profiles/counters_latest.json", "w"), indent=1)
print("".join(out))
