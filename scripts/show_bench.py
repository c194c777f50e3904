"""Print the essentials of bench.py JSON lines: python scripts/show_bench.py gpurun_out/*.log"""
import json
import sys

for path in sys.argv[1:]:
    try:
        lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
        d = json.loads(lines[-1])
    except Exception as e:
        print(f"{path}: no JSON line ({e}); tail: {open(path).read()[-400:]!r}")
        continue
    r = d["roofline"]
    extra = ""
    if d.get("exchange_every_substep"):
        extra += f" k1={d['exchange_every_substep']['value'] / 1e9:.1f}G"
    if d.get("model_days_per_hr"):
        extra += f" mdays/hr={d['model_days_per_hr']:.0f}"
    if d.get("cpu_baseline"):
        c = d["cpu_baseline"]
        extra += f" cpu={c['value'] / 1e6:.1f}M({c['cores']}thr) 1thr={c.get('one_thread_value', 0) / 1e6:.2f}M"
    print(f"{path}: {d['value'] / 1e9:6.2f} G  {d['ms_per_step']:.3f} ms/step  launch {r['avg_launch_ms'] * 1e3:.1f} us  frac {r['frac']:.3f}  "
          f"tile {d['config']['tile']} halo {d['config']['halo']} path {d['path']}{extra}")
