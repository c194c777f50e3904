"""Register-budget gate of the pair kernel (CPU suite: nothing runs on a GPU -- hipcc cross-compiles, and the Makefile keeps the
compiler's resource-usage remarks of every instantiation beside its object, csrc/evp_fused2*.res).

What is gated:
  * the instantiations the BASELINE configurations and scripts/bench_cases.py select on UNTILED grids -- plain (config 3 and the
    headline), walls (config 4, lat-lon), mask on uniform metrics with the default forcing (config 5) -- spill NO vector register;
  * every PEER instantiation of the plain family, the one `bench.py --gpus N` runs, spills none either (the "20 B of scratch" the
    compiler reports for them is a frame slot reserved next to their SGPR -> VGPR-lane spills: not one scratch_* instruction exists
    in those kernels, which `test_no_scratch_instruction_in_the_plain_family` checks on the ISA listing);
  * every other instantiation may spill at most what tests/golden/spill_table.json records for it (a ratchet: the table can only be
    lowered -- regenerate it with `python tests/test_spill_gate.py --write` after an improvement);
  * the occupancy each family is laid out for holds (3 waves per SIMD: plain / walls / mask on PER-ROW coefficients; 2: uniform
    coefficients -- they run 1024 tiles --, per-point metrics and ring-forced).
"""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "climaseaice.jl_amd", "csrc")
TABLE = os.path.join(ROOT, "tests", "golden", "spill_table.json")
VARIANTS = {0: "evp_fused2", 1: "evp_fused2_walls", 2: "evp_fused2_mask", 3: "evp_fused2_force", 4: "evp_fused2_mask_force",
            5: "evp_fused2_force_fd", 6: "evp_fused2_mask_force_fd", 7: "evp_fused2_force_x", 8: "evp_fused2_mask_force_x",
            9: "evp_fused2_force_w", 10: "evp_fused2_mask_force_w"}


def _rows():
    """{key: dict(vgprs, sgpr_spill, vgpr_spill, scratch, occupancy)} over every instantiation of every variant."""
    subprocess.check_call(["make", "-s", "-j8", "-C", CSRC], stdout=subprocess.DEVNULL)      # (no-op when the library is up to date)
    out = {}
    for v, stem in VARIANTS.items():
        txt = open(os.path.join(CSRC, stem + ".res")).read()
        for b in txt.split("remark: Function Name: ")[1:]:
            name = b.split()[0]
            m = re.search(r"k_pairI(.*?)EEv", name)
            if not m:
                continue
            fl = re.findall(r"L[bi](\d+)E", m.group(1))
            g = lambda k: int(re.search(k + r": (\d+)", b).group(1))       # noqa: E731
            key = f"v{v} UNI{fl[0]} AUF{fl[1]} CF{fl[6]} FULL{fl[7]} PEER{fl[8]} X{fl[9]} DLD{fl[10]}"
            out[key] = dict(vgprs=g("VGPRs"), sgpr_spill=g("SGPRs Spill"), vgpr_spill=g("VGPRs Spill"),
                            scratch=g(r"ScratchSize \[bytes/lane\]"), occupancy=g(r"Occupancy \[waves/SIMD\]"), variant=v,
                            full=fl[7] == "1", peer=fl[8] == "1", uni=fl[0] == "1", cf=int(fl[6]), auf=fl[1] == "1")
    return out


@pytest.fixture(scope="module")
def rows():
    return _rows()


def test_baseline_instantiations_spill_no_vector_register(rows):
    bad = []
    for k, r in rows.items():
        plain = r["variant"] == 0                                                       # config 3, the headline, bench.py --gpus N (PEER too)
        walls = r["variant"] == 1 and not r["peer"] and not r["full"]                   # config 4 (lat-lon, Bounded), channel / bounded cases
        mask5 = r["variant"] == 2 and not r["peer"] and not r["full"] and r["uni"] and r["cf"] >= 1      # config 5: uniform metrics, default forcing
        if (plain or walls or mask5) and r["vgpr_spill"]:
            bad.append((k, r["vgpr_spill"]))
    assert not bad, bad


def test_even_first_substep_instantiations_of_the_baseline_families(rows):
    """`AUF = 1` (the pair's first sub-step is u-first) is what csi_evp_subcycle runs when a caller that keeps the Julia sub-step loop
    starts a pair on an EVEN sub-step; csi_time_step_momentum starts at sub-step 1 and never selects it.  In the families the
    BASELINE configurations use, those instantiations are held to the same budget as their AUF = 0 twins: no spilled vector register, the
    occupancy of the family.  (Array forcing and per-point metrics: tests/golden/spill_table.json.)"""
    seen = 0
    for k, r in rows.items():
        base = r["variant"] == 0 or (r["variant"] == 1 and not r["peer"] and not r["full"]) or \
            (r["variant"] == 2 and not r["peer"] and not r["full"] and r["uni"] and r["cf"] >= 1)
        if base and r["auf"]:
            seen += 1
            assert r["vgpr_spill"] == 0, (k, r["vgpr_spill"])
            assert r["occupancy"] >= (2 if r["uni"] else 3), (k, r["occupancy"])
    assert seen >= 12 + 6 + 2


def test_spills_do_not_grow(rows):
    table = json.load(open(TABLE))
    worse = [(k, r["vgpr_spill"], table.get(k, {}).get("vgpr_spill")) for k, r in rows.items()
             if r["vgpr_spill"] > table.get(k, {"vgpr_spill": 0})["vgpr_spill"]]
    assert not worse, f"vector-register spills grew (instantiation, now, recorded): {worse}"
    assert set(table) == set(rows), "instantiation set changed: regenerate tests/golden/spill_table.json (python tests/test_spill_gate.py --write)"


def test_occupancy_targets(rows):
    bad = []
    for k, r in rows.items():
        ring_forced = r["variant"] >= 3 and not (r["variant"] == 8)       # array forcing through the LDS ring: laid out for 2 waves per SIMD
        # uniform coefficients (round 5): pair_geom gives them 1024 tiles = two waves per SIMD, and they hold the velocity phase's
        # coefficients in vector registers (Stage::hoist_uniform) -- all but variant 8 (mask + model.forcing / immersed-flux terms)
        uniform_two = r["uni"] and r["variant"] != 8
        want = 2 if (r["full"] or ring_forced or uniform_two) else 3
        if r["occupancy"] < want:
            bad.append((k, r["occupancy"], want))
    assert not bad, bad


def test_no_scratch_instruction_in_the_plain_family():
    """The plain family's PEER kernels report ScratchSize 20 B / lane and 0 spilled VGPRs: the ISA must hold no scratch_* instruction."""
    asm = "/tmp/spill_gate_v0.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "-ffp-contract=off", "-DCSI_PAIR_VARIANT=0", "-S", "--cuda-device-only", os.path.join(CSRC, "evp_fused2.hip"), "-o", asm],
                          stderr=subprocess.DEVNULL)
    n = sum(1 for ln in open(asm) if ln.strip().startswith("scratch_"))
    assert n == 0, f"{n} scratch_* instructions in the plain family"


if __name__ == "__main__" and "--write" in sys.argv:
    r = _rows()
    json.dump({k: {"vgprs": v["vgprs"], "vgpr_spill": v["vgpr_spill"], "sgpr_spill": v["sgpr_spill"], "scratch": v["scratch"], "occupancy": v["occupancy"]}
               for k, v in sorted(r.items())}, open(TABLE, "w"), indent=0)
    print(f"{len(r)} instantiations, {sum(1 for v in r.values() if v['vgpr_spill'])} with spilled vector registers -> {TABLE}")
