"""Instruction mix of the headline instantiation of the pair kernel from its ISA listing (no GPU needed: hipcc cross-compiles).

python scripts/isa_mix.py [--variant 0] [--uni 1] [--auf 1] [--cf 2] [--peer 0]  ->  profiles/isa_mix_k_pair.json

Static counts over the PRODUCER wave's row loop (unrolled three times; the consumer runs the same arithmetic plus its stores).  Every class carries the issue interval measured for it on
this chip (scripts/microbench/valu_rate, profiles/r01_microbenchmarks.md, >= 2 waves per SIMD); bench.py multiplies the kernel's
DYNAMIC vector-instruction count (PMC pass) by the mix's mean interval: roofline.fp64_issue_frac_weighted."""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ISSUE_NS = {"f64_fma_mul": 2.29, "f64_add": 2.00, "f64_minmax": 1.89, "f64_cmp": 2.20, "f64_rcp_rsq": 7.03, "dpp": 2.18,
            "cndmask": 2.10, "mov64": 1.79, "other32": 1.31}


def classify(op, line):
    if "dpp" in line or "row_" in line or "wave_sh" in line:
        return "dpp"
    if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
        return "f64_rcp_rsq"
    if re.match(r"v_(fma|mul|fmac)_f64", op):
        return "f64_fma_mul"
    if re.match(r"v_add_f64", op):
        return "f64_add"
    if re.match(r"v_(max|min)_f64", op):
        return "f64_minmax"
    if re.match(r"v_cmp\w*_f64", op):
        return "f64_cmp"
    if op.startswith("v_cndmask"):
        return "cndmask"
    if re.match(r"v_(mov_b64|pk_mov|lshl_add_u64)", op):
        return "mov64"
    return "other32"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--flags", default="Lb1ELb1ELb0ELb0ELb0ELb0ELi2ELb0ELb0ELi0ELb0E", help="mangled template arguments of the instantiation")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "isa_mix_k_pair.json"))
    a = ap.parse_args()
    src = os.path.join(ROOT, "climaseaice.jl_amd", "csrc", "evp_fused2.hip")
    asm = f"/tmp/isa_mix_v{a.variant}.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "-ffp-contract=off", f"-DCSI_PAIR_VARIANT={a.variant}", "-S", "--cuda-device-only", src, "-o", asm])
    name = None
    body = []
    for ln in open(asm):
        t = ln.strip()
        m = re.match(r"(_ZN3csi5fused6k_pairI(\w+)EvPKNS_10FusedTableEiiiiiy):", t)
        if m:
            name = m.group(1) if m.group(2).startswith(a.flags) or a.flags in m.group(2) else None
            continue
        if name and t.startswith(".end_amdhsa_kernel"):
            break
        if name and t.startswith("s_endpgm") and False:
            break
        if name:
            body.append(t)
        if name and t.startswith(".section") and body and len(body) > 100:
            break
    if not body:
        sys.exit(f"instantiation {a.flags} not found in {asm}")
    # the PRODUCER's row loop: the first loop of the listing that holds three s_barrier (the body is unrolled three times).  The
    # consumer's loop runs the same arithmetic (the other order of the two velocities) plus its stores, whose general path -- halo
    # images, row bookkeeping: cold code -- sits inside the loop's address range and would distort a static count.
    labels = {}
    for i, t in enumerate(body):
        m = re.match(r"(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = i
    loop = None
    for i, t in enumerate(body):
        m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            seg = body[labels[m.group(1)]:i]
            if sum(1 for x in seg if x.startswith("s_barrier")) == 3:
                loop = seg      # (the outermost back edge with three barriers wins: keep the longest)
                if len(seg) > 600:
                    break
    if loop is None:
        sys.exit("producer loop not found")
    c = collections.Counter()
    other = collections.Counter()
    for t in loop:
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            c[classify(op, t)] += 1
            c["valu"] += 1
        elif op.startswith(("global_load", "global_store", "scratch_")):
            other["vmem"] += 1
            if op.startswith("scratch_"):
                other["scratch"] += 1
        elif op.startswith("ds_"):
            other["lds"] += 1
        elif op.startswith("s_load"):
            other["smem"] += 1
        elif op.startswith("s_waitcnt"):
            other["waitcnt"] += 1
        elif op.startswith("s_barrier"):
            other["barrier"] += 1
        elif op.startswith("s_"):
            other["salu"] += 1
    valu = c.pop("valu")
    share = {k: c.get(k, 0) / valu for k in ISSUE_NS}
    mean = sum(share[k] * ISSUE_NS[k] for k in ISSUE_NS)
    out = {"kernel": "pair", "instantiation": name, "variant": a.variant, "valu_static": valu, "counts": {k: c.get(k, 0) for k in ISSUE_NS},
           "share": share, "issue_ns": ISSUE_NS, "mean_issue_ns": mean, "non_valu_static": dict(other),
           "valu_per_stage_row": valu / 3.0,
           "source": "scripts/isa_mix.py (static counts of the ISA listing, hipcc -S --cuda-device-only); issue intervals: profiles/r01_microbenchmarks.md valu_rate",
           "note": "the producer wave's row loop (unrolled x 3): valu_static / 3 = vector instructions per stage-row; the consumer runs the same arithmetic"}
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
