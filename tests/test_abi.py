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
