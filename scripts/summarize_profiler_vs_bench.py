"""Summary of scripts/profiler_vs_bench.sh: launch time of k_pair plain vs under rocprofv3, clock and power during both, trace gaps."""
import csv, glob, json, os, sys
out = sys.argv[1]


def clock(path):
    """median shader clock (MHz) and the largest power (W) of the BUSY card: the one whose power is highest"""
    best = None
    cards = {}
    for ln in open(path):
        for tok in ln.split():
            c, f, p = tok.split(":")
            cards.setdefault(c, []).append((int(f), int(p)))
    for c, v in cards.items():
        pw = sorted(x[1] for x in v)
        if best is None or pw[-1] > best[1]:
            fr = sorted(x[0] for x in v if x[1] > 0.7 * pw[-1])      # samples while the card was busy
            best = (c, pw[-1], fr[len(fr) // 2] if fr else None, fr[0] if fr else None, fr[-1] if fr else None, len(fr))
    return best


for rep in (1, 2):
    for kind in ("plain", "traced"):
        f = os.path.join(out, f"{kind}{rep}.json")
        try:
            d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        except Exception as e:
            print(kind, rep, "no bench line", e)
            continue
        r = d["roofline"]
        ck = clock(os.path.join(out, f"{kind}{rep}.clock"))
        print(f"{kind:6s} run {rep}: value {d['value'] / 1e9:6.2f} G  ms_per_step {d['ms_per_step']:.3f}  HIP-event launch {1e3 * r['avg_launch_ms']:.2f} us  "
              f"launches x avg {r['launches_x_avg_ms']:.3f} ms  | busy card {ck[0]}: sclk median {ck[2]} MHz (min {ck[3]}, max {ck[4]}, {ck[5]} samples), power max {ck[1]} W")
    rows = []
    for f in glob.glob(os.path.join(out, f"trace{rep}", "**", "*kernel_trace.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "k_pair" in r["Kernel_Name"]]
    if rows:
        for r in rows:
            r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        rows.sort(key=lambda r: r["s"])
        rows = rows[len(rows) // 4:]                      # the timed steps (the warm-up comes first)
        d = sorted((r["e"] - r["s"]) / 1e3 for r in rows)
        gaps = sorted((rows[k + 1]["s"] - rows[k]["e"]) / 1e3 for k in range(len(rows) - 1) if rows[k + 1]["s"] - rows[k]["e"] < 50000)
        pitch = sorted((rows[k + 1]["s"] - rows[k]["s"]) / 1e3 for k in range(len(rows) - 1) if rows[k + 1]["s"] - rows[k]["s"] < 300000)
        print(f"  trace {rep}: {len(rows)} k_pair launches: duration median {d[len(d) // 2]:.2f} us (p10 {d[len(d) // 10]:.2f}, p90 {d[9 * len(d) // 10]:.2f}); "
              f"gap to the next launch median {gaps[len(gaps) // 2]:.2f} us (p90 {gaps[9 * len(gaps) // 10]:.2f}); start-to-start median {pitch[len(pitch) // 2]:.2f} us")

for rep in (1, 2, 3):
    for kind in ("short_plain", "short_traced"):
        f = os.path.join(out, f"{kind}{rep}.json")
        try:
            d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        except Exception:
            continue
        r = d["roofline"]
        print(f"{kind:12s} run {rep} (--steps 2 --warmup 1): value {d['value'] / 1e9:6.2f} G  ms_per_step {d['ms_per_step']:.3f}  HIP-event launch {1e3 * r['avg_launch_ms']:.2f} us")
    rows = []
    for f in glob.glob(os.path.join(out, f"short_trace{rep}", "**", "*kernel_trace.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "k_pair" in r["Kernel_Name"]]
    if rows:
        for r in rows:
            r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        rows.sort(key=lambda r: r["s"])
        d = [(r["e"] - r["s"]) / 1e3 for r in rows]
        n = len(d)
        print(f"  short trace {rep}: {n} k_pair launches: mean of launches 1-60 {sum(d[:60]) / 60:.1f} us, 61-120 {sum(d[60:120]) / max(len(d[60:120]), 1):.1f} us, "
              f"121-180 {sum(d[120:180]) / max(len(d[120:180]), 1):.1f} us, all {sum(d) / n:.1f} us")
