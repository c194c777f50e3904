"""Round trip between this repository's seeded cases and bench/reference_driver.jl (the reference run in Julia).

    python tests/golden/reference_io.py export <dir>     inputs of REF_CASES -> <dir>/<case>/{case.txt, *.f64, mask.u8}
    julia --project=<ClimaSeaIce.jl> bench/reference_driver.jl <dir>
    python tests/golden/reference_io.py import <dir>     reference outputs -> tests/golden/ref_<case>.npz

tests/test_golden.py::test_oracle_matches_reference_fixture consumes tests/golden/ref_*.npz when present.  A fixture is
data only: inputs (seeded here) and the reference's outputs.  No Julia exists in the build image, so no ref_*.npz has been
produced yet -- the oracle stays "parity unpinned" until this round trip is run once.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# the golden cases (generate_golden.py) + one of each feature the hot path supports
REF_CASES = {
    "evp_periodic_32x32": dict(Nx=32, Ny=32, topo=("periodic", "periodic"), patches=True, random_uv=0.05, ue=0.03, ve=-0.01),
    "evp_bounded_latlon_48x32": dict(Nx=48, Ny=32, topo=("bounded", "bounded"), grid="latlon", patches=True, random_uv=0.05),
    "evp_channel_64x48": dict(Nx=64, Ny=48, topo=("periodic", "bounded"), patches=False, random_uv=0.02),
    "masked_channel_96x120": dict(Nx=96, Ny=120, topo=("periodic", "bounded"), patches=True, random_uv=0.05, land=0.3),
    "field_forcing_48x48": dict(Nx=48, Ny=48, topo=("periodic", "periodic"), patches=True, field_forcing=True, random_uv=0.03),
    "noslip_channel_64x40": dict(Nx=64, Ny=40, topo=("periodic", "bounded"), patches=True, random_uv=0.05, noslip=True),
    "free_drift_64x48": dict(Nx=64, Ny=48, topo=("periodic", "periodic"), patches=True, random_uv=0.05, ue=0.05, ve=-0.02,
                             top=(0.03, -0.02), free_drift=True),
    "beta_bounded_100x90": dict(Nx=100, Ny=90, topo=("bounded", "bounded"), patches=True, random_uv=0.05, beta=2e-10),
    "ice_strength_32x32": dict(Nx=32, Ny=32, topo=("periodic", "periodic"), pressure="ice_strength", coriolis=None, top=None,
                               ue=0.1, patches=False),
}
TAGS = ("momentum1", "momentum10", "step3_fe", "step3_rk3")
OUT = ("u", "v", "h", "a", "s11", "s22", "s12", "alpha", "P", "zeta_c", "zeta_f", "Delta")


def export(root, only=None):
    import cases
    os.makedirs(root, exist_ok=True)
    for name, kw in REF_CASES.items():
        if only is not None and name not in only:
            continue
        c = cases.make_case(substeps=10, **kw)
        d = os.path.join(root, name)
        os.makedirs(d, exist_ok=True)
        lines = {"Nx": c["Nx"], "Ny": c["Ny"], "H": c["H"], "topo_x": c["topo"][0], "topo_y": c["topo"][1], "grid": c["grid"],
                 "spacing": repr(float(c["spacing"])), "dt": repr(float(c["dt"])), "pressure": c["pressure"],
                 "bottom": c["bottom"] or "none", "ue": repr(float(c["ue"])), "ve": repr(float(c["ve"])),
                 "field_forcing": int(bool(c["field_forcing"])), "free_drift": int(bool(c["free_drift"])),
                 "noslip": int(bool(c["noslip"]))}
        if c["coriolis"] is not None:
            lines["coriolis"] = repr(float(c["coriolis"]))
        if c.get("beta") is not None:
            lines["beta"] = repr(float(c["beta"]))
        if c["top"] is not None and not c["field_forcing"]:
            lines["top_u"], lines["top_v"] = repr(float(c["top"][0])), repr(float(c["top"][1]))
        for k in ("h", "a", "u", "v"):
            np.ascontiguousarray(c[k], dtype="<f8").tofile(os.path.join(d, f"in_{k}.f64"))
        if c["field_forcing"]:
            for src, dst in (("top_u", "top_u"), ("top_v", "top_v"), ("ue_f", "ue"), ("ve_f", "ve")):
                np.ascontiguousarray(c[src], dtype="<f8").tofile(os.path.join(d, dst + ".f64"))
        if c.get("mask") is not None:
            lines["mask"] = 1
            np.ascontiguousarray(c["mask"], dtype=np.uint8).tofile(os.path.join(d, "mask.u8"))
        with open(os.path.join(d, "case.txt"), "w") as f:
            f.write("# written by tests/golden/reference_io.py; read by bench/reference_driver.jl\n")
            for k, v in lines.items():
                f.write(f"{k} = {v}\n")
        print("exported", name)


def import_(root, dest=HERE):
    import cases
    import climaseaice_jl_amd as csi
    for name, kw in REF_CASES.items():
        d = os.path.join(root, name)
        if not os.path.exists(os.path.join(d, "DONE")):
            print("skip (no reference output):", name)
            continue
        c = cases.make_case(substeps=10, **kw)
        g = c["g"]
        shapes = {"u": g.interior_size(csi.Face, csi.Center), "v": g.interior_size(csi.Center, csi.Face),
                  "s12": g.interior_size(csi.Face, csi.Face), "zeta_f": g.interior_size(csi.Face, csi.Face)}
        data = {"case_json": json.dumps(kw), "versions": open(os.path.join(d, "DONE")).read().strip()}
        for k in ("h", "a", "u", "v"):
            data[f"in_{k}"] = c[k]
        for tag in TAGS:
            for f in OUT:
                nx, ny = shapes.get(f, (g.Nx, g.Ny))
                a = np.fromfile(os.path.join(d, f"out_{f}_{tag}.f64"), dtype="<f8")
                data[f"{f}_{tag}"] = a.reshape(-1, nx) if a.size != nx * ny else a.reshape(ny, nx)
        np.savez_compressed(os.path.join(dest, f"ref_{name}.npz"), **data)
        print("imported", name)


if __name__ == "__main__":
    if len(sys.argv) != 3 or sys.argv[1] not in ("export", "import"):
        raise SystemExit(__doc__)
    (export if sys.argv[1] == "export" else import_)(sys.argv[2])


def weight_dtype_of(versions):
    """"f64" / "f32" from the DONE file of a reference run (bench/reference_driver.jl writes `weight_dtype ...` from the type
    parameters of WENO(order = 7): a second float type parameter is the precision of the smoothness / weight arithmetic)."""
    for ln in str(versions).splitlines():
        t = ln.split()
        if len(t) == 2 and t[0] == "weight_dtype" and t[1] in ("f64", "f32"):
            return t[1]
    return "f64"


def oracle_run(kw, tag, weight_dtype="f64"):
    """The oracle's counterpart of one reference_driver.jl run (same entry points, same order); returns interiors.
    weight_dtype: the WENO weight precision the reference run reported (weight_dtype_of)."""
    import cases
    import climaseaice_jl_amd as csi
    nsub = 1 if tag == "momentum1" else 10
    c = cases.make_case(substeps=nsub, **kw)
    p = cases.oracle_problem(c)
    p.s.weno_weights_f32 = 1 if weight_dtype == "f32" else 0
    if tag.startswith("momentum"):
        p.time_step_momentum(c["dt"])
    else:
        for it in range(3):
            if tag.endswith("fe"):
                p.time_step_fe(c["dt"], scheme=7, first_iteration=(it == 0))
            else:
                p.time_step_rk3(c["dt"], scheme=7)
    names = {"a": "aice"}
    return c, {f: p.interior(names.get(f, f)).copy() for f in OUT}
