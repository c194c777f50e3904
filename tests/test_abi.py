"""CPU tests of the C-ABI boundary: the library loads, exports every symbol include/csi.h declares,
and fails loudly (no CPU fallback) when there is no HIP device."""
import os
import re

import pytest

import climaseaice_jl_amd as csi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "csi.h")).read()
    return sorted(set(re.findall(r"\b(csi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = csi._lib.load()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), f"libcsi_hip.so does not export {s}"
    assert sorted(csi._lib.SYMBOLS) == syms, "python binding list out of sync with include/csi.h"
    assert L.csi_version() == 100


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(csi.CsiError) as e:
        csi.Context(0)
    assert "no HIP device" in str(e.value)
    g = csi.RectilinearGrid((8, 8), x=(0, 1), y=(0, 1))
    with pytest.raises(RuntimeError):
        csi.SeaIceModel(g, device="cpu")


def test_field_enum_matches_header():
    text = open(os.path.join(ROOT, "include", "csi.h")).read()
    body = text[text.index("CSI_F_U = 0"):text.index("CSI_F_COUNT")]
    names = re.findall(r"CSI_F_([A-Z0-9_]+)", body)
    assert names == csi._lib.FIELD_IDS


def test_struct_sizes_match_header_layout():
    import ctypes as C
    assert C.sizeof(csi._lib.EvpParams) == 7 * 8 + 2 * 4 + 4 * 8
    assert C.sizeof(csi._lib.Stress) == 4 * 4 + 6 * 8
    assert C.sizeof(csi._lib.Metrics) == 2 * 8 + 4 * 8 + 12 * 8 + 8
    assert C.sizeof(csi._lib.SlabParams) == 13 * 8 + 2 * 4 + 2 * 8 + 2 * 4 + 8
    assert C.sizeof(csi._lib.SnowParams) == 4 * 8 + 2 * 4


# ---- what a C compiler makes of include/csi.h (tests/abi_client.c, gcc) against the two hand-typed mirrors ---------------------------

def c_layout():
    import json
    import subprocess
    import abi_build
    exe = abi_build.build()
    return json.loads(subprocess.check_output([exe, "layout"]).decode())


def test_c_compiler_layout_matches_ctypes():
    """sizeof / offsetof of every struct as gcc lays them out == the ctypes Structures the whole Python side passes by reference."""
    import ctypes as C
    lay = c_layout()
    pairs = {"csi_metrics": csi._lib.Metrics, "csi_evp_params": csi._lib.EvpParams, "csi_stress": csi._lib.Stress,
             "csi_slab_params": csi._lib.SlabParams, "csi_snow_params": csi._lib.SnowParams}
    for cname, T in pairs.items():
        assert C.sizeof(T) == lay[cname]["size"], cname
        names = [f[0] for f in T._fields_]
        assert names == list(lay[cname]["fields"]), (cname, "field names / order")
        for n in names:
            assert getattr(T, n).offset == lay[cname]["fields"][n], (cname, n)
    e = lay["enums"]
    assert e["sizeof_enum"] == 4 and e["CSI_F_COUNT"] == len(csi._lib.FIELD_IDS) and e["CSI_F_FORCING_V"] == csi._lib.F["FORCING_V"]
    assert e["CSI_F_ZETA_C"] == csi._lib.F["ZETA_C"] and e["CSI_MODE_FAST"] == csi._lib.MODE_FAST and e["CSI_METRIC_FULL"] == csi._lib.METRIC_FULL
    assert e["CSI_STRESS_SEMI_IMPLICIT"] == csi._lib.STRESS_SEMI_IMPLICIT and e["CSI_VEL_FIELD"] == csi._lib.VEL_FIELD
    assert lay["library_version"] == e["CSI_VERSION"] == 100


def test_c_compiler_layout_matches_julia_stub():
    """The `struct Csi...` definitions of julia/ClimaSeaIceHIP.jl (never executed: no Julia here), laid out by C's rules, against gcc's."""
    lay = c_layout()
    text = open(os.path.join(ROOT, "julia", "ClimaSeaIceHIP.jl"), encoding="utf-8").read()
    size_of = {"Cdouble": (8, 8), "Float64": (8, 8), "Int32": (4, 4), "Cint": (4, 4), "Int64": (8, 8)}

    def type_layout(t):
        t = t.strip()
        if t.startswith("Ptr{"):
            return 8, 8
        m = re.match(r"NTuple\{(\d+),\s*(.*)\}$", t)
        if m:
            s, a = type_layout(m.group(2))
            return int(m.group(1)) * s, a
        return size_of[t]
    found = 0
    for jname, cname in (("CsiMetrics", "csi_metrics"), ("CsiEvpParams", "csi_evp_params"), ("CsiStress", "csi_stress"),
                         ("CsiSlabParams", "csi_slab_params"), ("CsiSnowParams", "csi_snow_params")):
        m = re.search(r"^struct\s+" + jname + r"\b[^\n]*\n(.*?)\nend", text, re.S | re.M)
        if not m:
            continue
        found += 1
        body = re.sub(r"#[^\n]*", "", m.group(1))
        fields = re.findall(r"([A-Za-z_]\w*)::((?:NTuple\{[^}]*\{[^}]*\}\})|(?:Ptr\{[^}]*\})|\w+)", body)
        off, align, offsets = 0, 1, {}
        for name, t in fields:
            s, a = type_layout(t)
            off = (off + a - 1) // a * a
            offsets[name] = off
            off += s
            align = max(align, a)
        size = (off + align - 1) // align * align
        assert size == lay[cname]["size"], (jname, size, lay[cname]["size"])
        assert list(offsets) == list(lay[cname]["fields"]), (jname, "field names / order")
        assert offsets == lay[cname]["fields"], jname
    assert found >= 3
