"""Instruction mix between two line numbers of an ISA listing: python scripts/isa_count.py file.s start end"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()[int(sys.argv[2]) - 1:int(sys.argv[3])]
c = collections.Counter()
for l in lines:
    l = l.strip()
    if not l or l.startswith((";", ".", "//")) or l.endswith(":"):
        continue
    op = l.split()[0]
    if op.startswith("v_"):
        if re.match(r"v_(fma|mul|add|max|min|rcp|rsq|fmac)_f64", op):
            c["valu_f64"] += 1
        elif "dpp" in l:
            c["valu_dpp"] += 1
        elif op.startswith(("v_cmp", "v_cndmask")):
            c["valu_cmp_sel"] += 1
        elif op.startswith(("v_mov", "v_accvgpr")):
            c["valu_mov"] += 1
        elif op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
            c["valu_lane"] += 1
        else:
            c["valu_other"] += 1
            c["other:" + op] += 1
    elif op.startswith("s_load") or op.startswith("s_buffer"):
        c["smem"] += 1
    elif op.startswith("s_waitcnt"):
        c["waitcnt"] += 1
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        c["branch"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
    elif op.startswith("global_load") or op.startswith("scratch_load"):
        c["vmem_load"] += 1
    elif op.startswith("global_store") or op.startswith("scratch_store"):
        c["vmem_store"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
    else:
        c["misc:" + op] += 1
tot = sum(v for k, v in c.items() if ":" not in k)
print("total", tot, {k: v for k, v in sorted(c.items()) if ":" not in k})
print({k: v for k, v in sorted(c.items()) if ":" in k})
