"""Summarise gpurun_out/prof_<tag> (scripts/same_lease_profile.sh: the plain bench.py line, the kernel trace and the PMC
passes of ONE lease) into profiles/<name>.md + profiles/<name>_kernel_stats.csv + profiles/counters_latest.json (the PMC
evidence bench.py quotes in roofline.traffic / roofline.valu_frac -- together with the launch time of the run it was taken on,
so that bench.py only prices it against its own clock when the two agree).

usage: python scripts/summarize_profile.py <tag> <name> ["note"]
"""
import collections, csv, glob, json, os, shutil, sys
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
       "command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-full-step --no-unfused --no-structure`\n"
       "(a WARM run since round 6: the first ~150 launches from an idle chip are 5-25 % slower, profiles/r06_profiler_vs_bench.md; 2048x2048 periodic f-plane, 120 sub-steps, FAST mode;\n"
       "counters in separate `--pmc` passes of `--steps 2 --warmup 1`: counts per launch do not depend on the clock)\n\n## kernel stats\n\n",
       "| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n"]
ks = glob.glob(base + "/trace/*/*_kernel_stats.csv")[0]
avg = collections.defaultdict(list)
for r in csv.DictReader(open(ks)):
    out.append(f"| {r['Name'][:96]} | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['Percentage']):.2f} |\n")
    c = classify(r["Name"])
    if c:
        avg[c[0]].append((float(r["AverageNs"]) / 1e3, int(r["Calls"])))
avg = {k: sum(a * n for a, n in v) / sum(n for _, n in v) for k, v in avg.items()}
# The --stats average runs over EVERY launch of the traced process: the warm-up step (cold clocks: 10-15 % slower launches), the
# timed steps and bench.py's 32-sub-step per-phase profile.  What has to agree with ms_per_step is the TIMED region: take it from
# the per-dispatch trace, with the traced run's own bench.py line (steps, warm-up, sub-steps per launch) telling which launches.
traced = None
for l in open(base + "/trace.log", errors="replace").read().splitlines():
    if l.startswith("{") and '"metric"' in l:
        traced = json.loads(l)
timed_avg = {}
seg_note = ""
kt = glob.glob(base + "/trace/*/*_kernel_trace.csv")
if traced and kt:
    byk = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        c = classify(r["Kernel_Name"])
        if c:
            byk[c[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    spl = traced["roofline"]["substeps_per_launch"]
    per_step = int(round(traced["config"]["substeps"] / spl))
    w, st = traced["warmup"], traced["steps"]
    for k, v in byk.items():
        v.sort()
        if len(v) >= (w + st) * per_step:
            seg = v[w * per_step:(w + st) * per_step]
            timed_avg[k] = sum(e - b for b, e in seg) / len(seg) / 1e3
            warm = v[:w * per_step]
            span = (seg[-1][1] - seg[0][0]) / st / 1e6
            seg_note += (f"| {k} | {len(v)} | {sum(e - b for b, e in warm) / max(len(warm), 1) / 1e3:.1f} | **{timed_avg[k]:.1f}** | "
                         f"{timed_avg[k] * per_step / 1e3:.3f} | {span:.3f} | {traced['ms_per_step']:.3f} | {traced['roofline']['avg_launch_ms'] * 1e3:.1f} |\n")
if seg_note:
    out.append("\n## the traced run, launch by launch (per-dispatch trace; the `--stats` average above includes the cold warm-up step)\n\n"
               "| kernel | launches | avg us, warm-up step | avg us, TIMED steps | x launches per step = ms | first start to last end per step, ms | "
               "`ms_per_step` of the traced run | its HIP-event us per launch |\n|---|---|---|---|---|---|---|---|\n" + seg_note + "\n")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(base + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        c = classify(r["Kernel_Name"])
        if c:
            agg[c[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
if os.path.exists(base + "/counters_by_kernel.json"):          # same_lease_profile.sh reduces the tables on the box
    for kname, v in json.load(open(base + "/counters_by_kernel.json")).items():
        c = classify(kname)
        if c:
            for cn, x in v.items():
                agg[c[0]][cn].append(x["mean"])
agg = {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in agg.items()}
# the plain bench.py lines of the same lease (before and after the profiling passes)
bench = {}
for tagb in ("bench", "bench_after"):
    fb = f"{base}/{tagb}.json"
    if os.path.exists(fb):
        lines = [l for l in open(fb).read().splitlines() if l.startswith("{")]
        if lines:
            bench[tagb] = json.loads(lines[-1])
if bench:
    out.append("## the same lease, unprofiled (`python3 bench.py --steps 20 --warmup 5`, before / after the profiling passes)\n\n"
               "| run | G cell-updates/s | ms per step (120 sub-steps) | HIP-event us per launch | roofline.frac | model-days/hr |\n|---|---|---|---|---|---|\n")
    for tagb, d in bench.items():
        r = d["roofline"]
        out.append(f"| {tagb} | {d['value']/1e9:.2f} | {d['ms_per_step']:.3f} | {r['avg_launch_ms']*1e3:.1f} | {r['frac']:.3f} | {d.get('model_days_per_hr') or 0:.0f} |\n")
    out.append("\n")
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
           "hbm_bytes_per_launch": traffic[dom], "avg_launch_us": timed_avg.get(dom, avg[dom]), "stats_avg_launch_us_incl_warmup": avg[dom]}
    if traced:
        ctr["traced_run"] = {"ms_per_step": traced["ms_per_step"], "hip_event_launch_us": traced["roofline"]["avg_launch_ms"] * 1e3, "value": traced["value"]}
    if "SQ_INSTS_VALU" in v:
        ctr["valu_insts_per_launch"] = v["SQ_INSTS_VALU"]
    # round 5: the DYNAMIC mix by class (pmc_mix pass): FP64 fma / mul / add, transcendental (v_rcp_f64 / v_rsq_f64), integer; the rest
    # -- DPP moves, compares, selects, 64-bit moves -- is what is left of SQ_INSTS_VALU
    mixkeys = {"SQ_INSTS_VALU_FMA_F64": "fma_f64", "SQ_INSTS_VALU_MUL_F64": "mul_f64", "SQ_INSTS_VALU_ADD_F64": "add_f64",
               "SQ_INSTS_VALU_TRANS_F64": "trans_f64", "SQ_INSTS_VALU_INT32": "int32", "SQ_INSTS_VALU_INT64": "int64"}
    if all(k in v for k in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")):
        mix = {name_: v[k] for k, name_ in mixkeys.items() if k in v}
        mix["other"] = v["SQ_INSTS_VALU"] - sum(mix.values())
        ctr["valu_mix_per_launch"] = mix
    if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v:
        # SQ_ACTIVE_INST_VALU: quad-cycles summed over SIMDs; GRBM_GUI_ACTIVE: cycles summed over the 8 XCDs
        ctr["valu_busy_frac"] = v["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * v["GRBM_GUI_ACTIVE"] / 8.0)
    if "bench" in bench:
        b = bench["bench"]
        ctr["same_lease_bench"] = {"value": b["value"], "ms_per_step": b["ms_per_step"], "avg_launch_us": b["roofline"]["avg_launch_ms"] * 1e3,
                                   "substeps_per_launch": b["roofline"]["substeps_per_launch"]}
        n_launch = b["config"]["substeps"] / b["roofline"]["substeps_per_launch"]
        ta = timed_avg.get(dom, avg[dom])
        ctr["trace_avg_x_launches_ms"] = ta * n_launch / 1e3
        out.append(f"\nconsistency: kernel average over the timed launches of the traced run {ta:.1f} us x {n_launch:.0f} launches = {ta * n_launch / 1e3:.3f} ms "
                   f"<= its own ms_per_step {traced['ms_per_step'] if traced else float('nan'):.3f}; the unprofiled run of the same lease: ms_per_step {b['ms_per_step']:.3f}, "
                   f"HIP-event launch time {b['roofline']['avg_launch_ms']*1e3:.1f} us (a step also holds initialize_rheology!, the halo fills and finalize_rheology!).\n")
        open(f"profiles/{name}.md", "w").write("".join(out))
    # (a fourth argument "keep": a profile of something other than the headline -- a tile -- leaves bench.py's static evidence alone)
    json.dump(ctr, open("profiles/counters_latest.json" if not (len(sys.argv) > 4 and sys.argv[4] == "keep") else f"profiles/{name}_counters.json", "w"), indent=1)
shutil.copy(ks, f"profiles/{name}_kernel_stats.csv")
print("".join(out))
