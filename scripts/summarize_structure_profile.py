"""Summary of a scripts/prof_structure.sh directory: per-kernel launch statistics of the LAST traced step and PMC averages."""
import collections, csv, glob, os, sys
out = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
if rows:
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    # the last sub-cycle: from the last k_init-like launch on
    names = [r["Kernel_Name"] for r in rows]
    last_init = max(k for k, n in enumerate(names) if "k_init" in n or "fast_init" in n or "k_fast_init" in n) if any("init" in n for n in names) else 0
    cyc = rows[last_init:]
    t0, t1 = cyc[0]["s"], max(r["e"] for r in cyc)
    print(f"last sub-cycle: {len(cyc)} launches, {(t1 - t0) / 1e3:.1f} us wall")
    agg = collections.OrderedDict()
    for r in cyc:
        k = r["Kernel_Name"].split("(")[0][:110]
        a = agg.setdefault(k, [0, 0, 1 << 62, 0])
        d = r["e"] - r["s"]
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    for k, (n, tot, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:5d} x {tot / n / 1e3:9.2f} us (min {mn / 1e3:.2f}, max {mx / 1e3:.2f})  total {tot / 1e3:9.1f} us   {k}")
    # pair launches: the sequence of durations (first two and the last run every tile)
    pair = [r for r in cyc if "k_pair" in r["Kernel_Name"]]
    if pair:
        d = [(r["e"] - r["s"]) / 1e3 for r in pair]
        gaps = [(pair[k + 1]["s"] - pair[k]["e"]) / 1e3 for k in range(len(pair) - 1)]
        print("  k_pair durations (us): first", [round(x, 1) for x in d[:3]], "... middle mean", round(sum(d[2:-1]) / max(len(d[2:-1]), 1), 1), "... last", round(d[-1], 1))
        print("  gaps between consecutive k_pair launches (us): mean", round(sum(gaps) / max(len(gaps), 1), 2), "max", round(max(gaps), 2) if gaps else None)
for p in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(p):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "k_pair" not in k:
            continue
        print(os.path.basename(p), k)
        for c, x in sorted(v.items()):
            xs = sorted(x)
            print(f"    {c:28s} n {len(x):4d}  median {xs[len(xs) // 2]:.5g}  max {xs[-1]:.5g}  min {xs[0]:.5g}")
