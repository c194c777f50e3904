"""python scripts/summarize_adv_profile.py <tag>: gpurun_out/prof_adv_<tag>/summary.json (scripts/adv_profile.sh) ->
profiles/<tag>_advection.md (kernel times, HBM traffic with FETCH_SIZE doubled per the gfx950 rule, instruction counts and mix) and
profiles/counters_advection.json (what bench.py's `advection` record quotes)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
d = json.load(open(os.path.join(ROOT, "gpurun_out", f"prof_adv_{tag}", "summary.json")))
plain = open(os.path.join(ROOT, "gpurun_out", f"prof_adv_{tag}", "plain.txt")).read().strip().splitlines()
W = {"SQ_INSTS_VALU_FMA_F64": 2.29, "SQ_INSTS_VALU_MUL_F64": 2.29, "SQ_INSTS_VALU_ADD_F64": 2.00, "SQ_INSTS_VALU_TRANS_F64": 7.03,
     "SQ_INSTS_VALU_INT32": 1.31, "SQ_INSTS_VALU_INT64": 1.79}
out = [f"# Advection kernels under rocprofv3 ({tag}; scripts/adv_profile.sh, one MI355X)\n\n",
       "WENO(order = 7) of h and aice on a periodic grid, FAST mode, advection-only SplitRungeKutta3 steps (BASELINE config 2 at 512^2: one launch per\n"
       "RK stage, `k_tendencies<7, FAST, STEP, NT = 2>`; 2048^2: separate tendency / update launches).  FETCH_SIZE is DOUBLED (gfx950 reports half\n"
       "of the bytes read: MI355X_MICROARCH.md; round 3's file of this name had it undoubled).\n\n## plain runs (no profiler)\n\n```\n" + "\n".join(plain) + "\n```\n"]
ctr = {}
for N, kernels in d.items():
    out.append(f"\n## {N}^2\n\n| kernel | calls | avg us (trace) | FETCH_SIZE KB | WRITE_SIZE KB | HBM MB (2 x fetch + write) | VALU insts | lane-insts / cell / tracer | VALU busy | waiting | issue us / SIMD |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
    stats = {r["Name"]: r for r in kernels.get("_stats", [])}
    for k, v in kernels.items():
        if k == "_stats" or "k_tendencies" not in k and "k_tracer_step" not in k:
            continue
        m = {c: x["mean"] for c, x in v.items()}
        st = stats.get(k, {})
        avg_us = float(st.get("AverageNs", 0)) / 1e3
        hbm = (2 * m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) * 1024
        insts = m.get("SQ_INSTS_VALU", 0)
        n = int(N)
        mixed = sum(m.get(c, 0) * w for c, w in W.items())
        other = insts - sum(m.get(c, 0) for c in W)
        issue_ns = mixed + other * 2.0
        busy = m["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0) if "GRBM_GUI_ACTIVE" in m and "SQ_ACTIVE_INST_VALU" in m else float("nan")
        wait = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] else float("nan")
        short = k.split("(")[0][-60:]
        out.append(f"| `{short}` | {st.get('Calls', '?')} | {avg_us:.1f} | {m.get('FETCH_SIZE', 0):.0f} | {m.get('WRITE_SIZE', 0):.0f} | {hbm / 1e6:.1f} | {insts:.4g} | "
                   f"{insts * 64 / (2 * n * n):.0f} | {busy:.2f} | {wait:.2f} | {issue_ns / 1024 / 1e3:.1f} |\n")
        if "k_tendencies" in k:
            mix = {c[len('SQ_INSTS_VALU_'):].lower(): m.get(c, 0) for c in W}
            mix["other"] = other
            ctr[N] = {"kernel": short, "valu_insts_per_launch": insts, "issue_ns_per_launch": issue_ns, "hbm_bytes_per_launch": hbm, "trace_avg_us": avg_us,
                      "valu_busy_frac": busy, "mix": mix, "source": f"profiles/{tag}_advection.md"}
out.append("\nissue us / SIMD = the launch's vector instructions by class x the issue interval measured for the class (profiles/r01_microbenchmarks.md: fma / mul 2.29 ns,\n"
           "add 2.00, transcendental 7.03, integer 1.31 / 1.79, the rest 2.0) / 1024 SIMDs: what the launch would take were FP64 issue its only limit.\n")
open(os.path.join(ROOT, "profiles", f"{tag}_advection.md"), "w").write("".join(out))
json.dump(ctr, open(os.path.join(ROOT, "profiles", "counters_advection.json"), "w"), indent=1)
print("".join(out))
