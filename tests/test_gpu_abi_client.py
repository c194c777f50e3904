"""The C ABI driven by a plain-C program (tests/abi_client.c, gcc, include/csi.h only) against the ctypes path, bit for bit.

What Julia's `ccall` binds is what a C compiler makes of the header (/root/reference/src/SeaIceDynamics/
split_explicit_momentum_equations.jl:103-195 is the entry the call replaces); this is the closest this image gets to executing it."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

import abi_build
import cases
import climaseaice_jl_amd as csi
from test_gpu_evp import EVP_FIELDS

pytestmark = pytest.mark.gpu

NAMES = ["u", "v", "h", "a", "s11", "s22", "s12", "un", "vn", "P", "alpha", "Delta", "zeta_f", "zeta_c"]


def put(f, name, arr):
    arr = np.ascontiguousarray(arr)
    dtype = 1 if arr.dtype == np.float64 else 0
    assert arr.dtype in (np.float64, np.int32)
    f.write(name.encode().ljust(16, b"\0") + struct.pack("<iiq", dtype, 0, arr.size) + arr.tobytes())


def get_all(path):
    out = {}
    with open(path, "rb") as f:
        while True:
            head = f.read(32)
            if len(head) < 32:
                break
            name = head[:16].rstrip(b"\0").decode()
            dtype, _, n = struct.unpack("<iiq", head[16:])
            out[name] = np.frombuffer(f.read(n * (8 if dtype else 4)), dtype=np.float64 if dtype else np.int32)
    return out


CASES = {
    "periodic_uniform": dict(Nx=96, Ny=64, substeps=20, random_uv=0.02),
    "latlon_bounded": dict(Nx=72, Ny=80, topo=("bounded", "bounded"), grid="latlon", substeps=11, random_uv=0.03, ue=0.02, ve=-0.01),
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("mode", ["fast", "strict"])
def test_c_client_equals_ctypes_path(name, mode, tmp_path):
    exe = abi_build.build()
    c = cases.make_case(**CASES[name])
    m = cases.csi_model(c, mode=mode)
    m.synchronize()
    torch.cuda.synchronize()
    g = c["g"]
    fields = {"u": m.velocities.u, "v": m.velocities.v, "h": m.ice_thickness, "a": m.ice_concentration}
    fields.update({k: EVP_FIELDS[k](m) for k in NAMES[4:]})
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    met = g.metrics()
    d, r = m.dynamics, m.dynamics.rheology
    with open(inp, "wb") as f:
        topo = [csi.model._TOPO[t] for t in g.topology]
        kind = {"uniform": csi._lib.METRIC_UNIFORM, "per_j": csi._lib.METRIC_PER_J}[met["kind"]]
        put(f, "grid", np.array([g.Nx, g.Ny, g.Hx, g.Hy, topo[0], topo[1], kind, c["substeps"],
                                 csi._lib.MODE_FAST if mode == "fast" else csi._lib.MODE_STRICT], dtype=np.int32))
        put(f, "metrics", np.array([met.get("dx", 0.0), met["dy"]], dtype=np.float64))
        if met["kind"] == "per_j":
            for k in ("dxc", "dxf", "azc", "azf"):
                put(f, k, np.asarray(met[k], dtype=np.float64))
        put(f, "extents", np.array([x for k in NAMES for x in (fields[k].ni, fields[k].nj)], dtype=np.int32))
        for k in NAMES[:4]:
            put(f, k, fields[k].numpy())                 # parents as set_ left them (halos filled)
        put(f, "evp", np.array([r.ice_compressive_strength, r.ice_compaction_hardening, r.yield_curve_eccentricity, r.minimum_plastic_stress,
                                r.min_relaxation_parameter, r.max_relaxation_parameter, r.relaxation_strength, float(getattr(d.coriolis, "f", 0.0)),
                                d.minimum_concentration, d.minimum_mass, m.sea_ice_density], dtype=np.float64))
        put(f, "evp_i", np.array([csi._lib.PRESSURE_REPLACEMENT, 0 if d.coriolis is None else 1], dtype=np.int32))
        tu, tv = c["top"]
        put(f, "stress_top_i", np.array([csi._lib.STRESS_CONST, 0, 0], dtype=np.int32))
        put(f, "stress_top", np.array([tu, tv, 0.0, 0.0, 0.0, 0.0], dtype=np.float64))
        bot = d.external_momentum_stresses.bottom
        vk = lambda v: csi._lib.VEL_ZERO if v is None else csi._lib.VEL_CONST      # noqa: E731
        put(f, "stress_bot_i", np.array([csi._lib.STRESS_SEMI_IMPLICIT, vk(bot.ue), vk(bot.ve)], dtype=np.int32))
        put(f, "stress_bot", np.array([0.0, 0.0, bot.ue or 0.0, bot.ve or 0.0, bot.rho_e, bot.Cd], dtype=np.float64))
        put(f, "dt", np.array([c["dt"]], dtype=np.float64))
    # the ctypes path
    csi.time_step_momentum(m, c["dt"])
    m.synchronize()
    # the C program (its own process, its own context and allocations)
    res = subprocess.run([exe, "run", inp, outp], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    got = get_all(outp)
    for k in ("u", "v", "s11", "s22", "s12", "alpha", "Delta", "zeta_f", "zeta_c", "P", "un", "vn"):
        mine = fields[k].numpy().ravel()
        assert np.array_equal(mine.view(np.int64), got[k].view(np.int64)), (name, mode, k, np.abs(mine - got[k]).max())
