"""Load-batching gate of the three-kernel velocity / stress functions and the fold band's launches (CPU suite: hipcc cross-compiles
csrc/evp_fast.hip to gfx950 assembly, nothing runs on a GPU).

Round 6b (profiles/r06_band.md): a masked, array-forced velocity point of `k_ustep2` made ~45 DEPENDENT memory round trips -- a branch
and an `s_waitcnt vmcnt(0)` around every mask byte, the forcing arrays loaded behind the waits of what came before -- which is what
those kernels cost where they are latency-bound (the fold band beside a pair launch: 17.8 us per launch, 9.4 since).  The fix is a
property of the generated code, so this gate reads the generated code: in each gated kernel the FIRST `s_waitcnt vmcnt(..)` must come
after (almost) all of the point's vector loads have been issued, and a full drain `vmcnt(0)` may follow a load only a few times.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "climaseaice.jl_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# kernel-name fragment -> (loads that must be in flight before the first vector-memory wait, most `load ... vmcnt(0)` drains tolerated
# in the whole kernel: the gather's conditional loads, the rare immersed-flux term and store_with_images' scratch reloads)
GATED = {
    # (measured on the round-6b build: 24-36 loads in flight, 5-6 drains -- the rare immersed-flux term and store_with_images' reloads;
    #  the code before it: the first wait behind ~10 loads, ~40 drains)
    "8k_ustep2ILb1E": (30, 8), "8k_vstep2ILb1E": (30, 8),                        # per-point metrics, mask
    "8k_ustep2ILb0E": (24, 8), "8k_vstep2ILb0E": (24, 8),
    "7k_ustepILb1ELb1E": (20, 8), "7k_vstepILb1ELb1E": (20, 8),                  # uniform coefficients, mask
    "7k_ustepILb0ELb1E": (20, 8), "7k_vstepILb0ELb1E": (20, 8),                  # per-row coefficients, mask
    "10k_band_velILi2ELb1ELb1E": (30, 8), "10k_band_velILi2ELb1ELb0E": (30, 8),  # the band's velocity launches, per-point metrics + mask
    "13k_band_stressILi2E": (40, 1),                                              # the band's stress launch: 48 plane values + 31 field values first
}


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "evp_fast.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-ffp-contract=off",
                           "--cuda-device-only", "-S", os.path.join(CSRC, "evp_fast.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    starts = [(k, ln.split(":")[0]) for k, ln in enumerate(lines) if re.match(r"^_ZN3csi\S*: ", ln)]
    return {name: lines[k:(starts[n + 1][0] if n + 1 < len(starts) else len(lines))] for n, (k, name) in enumerate(starts)}


def sequence(body):
    """'L' per vector-memory load, 'W<n>' per s_waitcnt vmcnt(n), in program order"""
    seq = []
    for ln in body:
        t = ln.strip()
        if t.startswith(("global_load", "buffer_load", "flat_load")):
            seq.append("L")
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            seq.append("W" + re.search(r"vmcnt\((\d+)\)", t).group(1))
    return seq


@pytest.mark.parametrize("frag", list(GATED))
def test_loads_are_in_flight_before_the_first_wait(asm, frag):
    need, drains_allowed = GATED[frag]
    names = [n for n in asm if frag in n]
    assert len(names) == 1, (frag, names)
    seq = sequence(asm[names[0]])
    first_wait = next(k for k, s in enumerate(seq) if s.startswith("W"))
    in_flight = seq[:first_wait].count("L")
    assert in_flight >= need, f"{names[0]}: only {in_flight} loads issued before the first vector-memory wait (gate {need}): {''.join(seq)[:300]}"
    drains = sum(1 for a, b in zip(seq, seq[1:]) if a == "L" and b == "W0")
    assert drains <= drains_allowed, f"{names[0]}: {drains} single-load round trips (load followed by vmcnt(0); gate {drains_allowed})"
