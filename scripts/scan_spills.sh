#!/bin/bash
# Register-budget check of every instantiation of the pair kernel (no GPU needed): which ones spill to scratch memory.
# A run-time branch added to a shared instantiation can push it over its 168-VGPR budget unnoticed (round 3: wind-drag loads in
# the array-forcing variants cost model.forcing configurations a quarter of their rate until this scan showed the spills).
cd "$(dirname "$0")/../climaseaice.jl_amd/csrc"
for v in 0 1 2 3 4 5 6 7 8 9 10; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -ffp-contract=off -DCSI_PAIR_VARIANT=$v -c evp_fused2.hip -o /tmp/scan_w$v.o -Rpass-analysis=kernel-resource-usage 2>/tmp/scan_res$v.txt &
done
wait
for v in 0 1 2 3 4 5 6 7 8 9 10; do python3 - $v <<'PY'
import re, sys
v = sys.argv[1]
txt = open(f"/tmp/scan_res{v}.txt").read()
names = re.findall(r"Function Name: (\S+)", txt)
vg = re.findall(r"VGPRs: (\d+)", txt)
sc = re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", txt)
rows = []
for n, a, b in zip(names, vg, sc):
    m = re.search(r"k_pairI(.*?)EEv", n)
    flags = re.findall(r"L[bi](\d+)E", m.group(1) if m else "")
    rows.append((flags, int(a), int(b)))
spill = [r for r in rows if r[2] > 0]
print(f"variant {v}: {len(rows)} instantiations, {len(spill)} with scratch" + ("" if not spill else ": " + "; ".join(
    f"UNI{r[0][0]} AUF{r[0][1]} CF{r[0][6]} FULL{r[0][7]} PEER{r[0][8]} EXTRA{r[0][9]} DLD{r[0][10]} {r[2]} B" for r in spill)))
PY
done
