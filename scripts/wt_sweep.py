"""Round 4: write-through result stores of the pair kernel (CSI_WRITE_THROUGH=0 / 1) by grid size, untiled periodic f-plane, 120 sub-steps.
One process per (size, setting): the knob is read when the context is created."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = ["512x512", "1024x512", "2048x256", "1024x1024", "2048x512", "1536x1536", "2048x1024", "2048x2048", "3072x3072"]
for sz in sizes:
    row = []
    for rep in range(2):
        for wt in ("0", "1"):
            env = dict(os.environ, CSI_WRITE_THROUGH=wt)
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-full-step",
                                "--no-unfused", "--tile", sz], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            d = json.loads(p.stdout.strip().splitlines()[-1])
            row.append((wt, round(d["value"] / 1e9, 2)))
    print(sz, row, flush=True)
