#!/usr/bin/env python3
"""Static check of julia/ClimaSeaIceHIP.jl against the reference's sources (build container only: needs /root/reference;
there is no Julia here, so the shim cannot be executed).  Asserts, by parsing text on both sides:

  1. the type-parameter POSITION the shim dispatches on: `SeaIceModel{GR, TD, SNT, D, TS, ...}` (sea_ice_model.jl:22) --
     `HIPSeaIceModel` must constrain the parameter named D (dynamics), `HIPFESeaIceModel` / `HIPRKSeaIceModel` D and TS, at the
     positions the reference's own aliases FESeaIceModel / RKSeaIceModel use (sea_ice_fe_step.jl:9, sea_ice_rk_substep.jl:6);
  2. `SeaIceMomentumEquation{S, ...}`: the solver is the parameter `HIPMomentumEquation` constrains (the first);
  3. every function the shim extends (`function ClimaSeaIce.<...>.f(` / `ClimaSeaIce.f(...) =` / `Oceananigans.<...>.f(`)
     is defined or extended under that name by the reference, with the same number of positional arguments;
  4. every Oceananigans topology name the shim maps appears in the reference's own import list
     (SeaIceDynamics/split_explicit_momentum_equations.jl:5-16), and none of that list is left unmapped;
  5. every `csi_*` symbol the shim ccalls is declared in include/csi.h with the same number of arguments;
  6. the struct fields the shim reads from the model / dynamics / rheology exist in the reference's struct definitions.
Exit code 0 = consistent; prints one line per check."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("CSI_REFERENCE", "/root/reference")
SHIM = os.path.join(ROOT, "julia", "ClimaSeaIceHIP.jl")


def read(*p):
    return open(os.path.join(*p), encoding="utf-8").read()


def struct_params(text, name):
    m = re.search(r"struct\s+" + name + r"\{([^}]*)\}", text)
    assert m, f"struct {name} not found"
    return [x.strip() for x in m.group(1).split(",")]


def struct_fields(text, name):
    m = re.search(r"struct\s+" + name + r"\b[^\n]*\n(.*?)\nend", text, re.S)
    assert m, f"struct {name} not found"
    return re.findall(r"^\s*([A-Za-z_ -￿][\w -￿]*)\s*::", m.group(1), re.M)


def alias_constraints(text, alias):
    """positions (0-based) of the parameters an alias `const X = SeaIceModel{<:Any, ..., <:T}` constrains, with their bounds"""
    m = re.search(r"const\s+" + alias + r"\s*=\s*(\w+)\{([^}]*)\}", text)
    assert m, f"alias {alias} not found"
    parts = [x.strip() for x in m.group(2).split(",")]
    return m.group(1), {k: p[2:].strip() for k, p in enumerate(parts) if p != "<:Any"}


def nargs(sig):
    """number of positional arguments of a Julia signature string `a, b::T, c = 1; kw`"""
    sig = sig.split(";")[0]
    depth, n, cur = 0, 0, ""
    for ch in sig:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            n += 1 if cur.strip() else 0
            cur = ""
        else:
            cur += ch
    return n + (1 if cur.strip() else 0)


def main():
    if not os.path.isdir(REF):
        print(f"check_julia_shim: {REF} not present -- nothing to check against")
        return 77
    shim = read(SHIM)
    src = {}
    for d, _, fs in os.walk(os.path.join(REF, "src")):
        for f in fs:
            if f.endswith(".jl"):
                src[os.path.relpath(os.path.join(d, f), REF)] = read(d, f)
    alltext = "\n".join(src.values())
    ok = True

    def check(cond, msg):
        nonlocal ok
        print(("ok   " if cond else "FAIL ") + msg)
        ok = ok and bool(cond)

    # 1. SeaIceModel's parameter positions
    params = struct_params(src["src/sea_ice_model.jl"], "SeaIceModel")
    fields = re.search(r"struct SeaIceModel\{.*?\nend", src["src/sea_ice_model.jl"], re.S).group(0)
    dyn_param = re.search(r"^\s*dynamics\s*::\s*(\w+)", fields, re.M).group(1)
    ts_param = re.search(r"^\s*timestepper\s*::\s*(\w+)", fields, re.M).group(1)
    d_pos, ts_pos = params.index(dyn_param), params.index(ts_param)
    base, c = alias_constraints(shim, "HIPSeaIceModel")
    check(base == "SeaIceModel" and c == {d_pos: "HIPMomentumEquation"},
          f"HIPSeaIceModel constrains parameter {d_pos + 1} ({dyn_param} = dynamics) of SeaIceModel{{{', '.join(params[:6])}, ...}}: {c}")
    for alias, ref_alias, ref_file in (("HIPFESeaIceModel", "FESeaIceModel", "src/sea_ice_fe_step.jl"),
                                       ("HIPRKSeaIceModel", "RKSeaIceModel", "src/sea_ice_rk_substep.jl")):
        _, rc = alias_constraints(src[ref_file], ref_alias)
        _, sc = alias_constraints(shim, alias)
        check(set(rc) == {ts_pos} and set(sc) == {d_pos, ts_pos} and sc[d_pos] == "HIPMomentumEquation" and
              sc[ts_pos].split(".")[-1] == rc[ts_pos].split(".")[-1],
              f"{alias} = {ref_alias} (parameter {ts_pos + 1}: {rc.get(ts_pos)}) ∩ HIPSeaIceModel: {sc}")
    # 2. the solver parameter of SeaIceMomentumEquation
    mp = struct_params(src["src/SeaIceDynamics/sea_ice_momentum_equations.jl"], "SeaIceMomentumEquation")
    mf = re.search(r"struct SeaIceMomentumEquation\{.*?\nend", src["src/SeaIceDynamics/sea_ice_momentum_equations.jl"], re.S).group(0)
    s_param = re.search(r"^\s*solver\s*::\s*(\w+)", mf, re.M).group(1)
    base, c = alias_constraints(shim, "HIPMomentumEquation")
    check(base == "SeaIceMomentumEquation" and c == {mp.index(s_param): "HIPSplitExplicitSolver"},
          f"HIPMomentumEquation constrains parameter {mp.index(s_param) + 1} ({s_param} = solver) of SeaIceMomentumEquation: {c}")
    # 3. extended functions exist with the same arity
    ext = re.findall(r"^(?:function\s+)?((?:ClimaSeaIce|Oceananigans)(?:\.\w+)*)\.([\w!]+)\(([^\n]*?)\)\s*(?:=|$)", shim, re.M)
    check(len(ext) >= 5, f"{len(ext)} extended methods found in the shim")
    for mod, fn, sig in ext:
        pat = re.compile(r"^(?:@inline\s+)?(?:function\s+)?(?:[\w.]+\.)?" + re.escape(fn) + r"\(([^\n]*?)\)(?:\s+where[^\n=]*)?\s*(?:=[^=]|$)", re.M)
        sigs = [m.group(1) for m in pat.finditer(alltext)]
        ar = sorted({nargs(x) for x in sigs})
        # optional arguments (x = default) give several arities
        mine = nargs(sig)
        opt = sig.count("=")
        check(bool(sigs) and any(a in ar for a in range(mine - opt, mine + 1)),
              f"{mod}.{fn}: {mine} positional argument(s) in the shim, reference defines arities {ar}")
    # 4. topologies
    imp = re.search(r"using Oceananigans\.Grids:(.*?)\nusing", src["src/SeaIceDynamics/split_explicit_momentum_equations.jl"], re.S).group(1)
    ref_topos = set(re.findall(r"\b((?:Left|Right|Fully)\w*(?:Connected|Folded))\b", imp))
    shim_topos = set(re.findall(r"OG\.((?:Left|Right|Fully)\w*(?:Connected|Folded))\b", shim)) | \
        set(re.findall(r":((?:Left|Right|Fully)\w*(?:Connected|Folded))\b", shim))
    check(ref_topos <= shim_topos, f"every connected / folded topology the reference imports is mapped or refused by the shim: missing {sorted(ref_topos - shim_topos)}")
    check(shim_topos <= ref_topos, f"the shim names no topology the reference does not import: extra {sorted(shim_topos - ref_topos)}")
    # 5. ccall'ed symbols vs include/csi.h
    hdr = re.sub(r"/\*.*?\*/", "", read(ROOT, "include", "csi.h"), flags=re.S)
    decl = {m.group(1): nargs(m.group(2)) for m in re.finditer(r"\b(csi_\w+)\s*\(([^;{]*?)\)\s*;", hdr)}
    for m in re.finditer(r"ccall\(\(:(csi_\w+),\s*libcsi\),\s*\w+,\s*\(([^)]*)\)", shim):
        name, types = m.group(1), m.group(2)
        n = nargs(types.rstrip(", "))
        want = decl.get(name)
        if want is not None and decl[name] == 1 and re.search(name + r"\s*\(\s*void\s*\)", hdr):
            want = 0
        check(want == n, f"ccall {name}: {n} argument type(s), include/csi.h declares {want}")
    # 6. struct fields the shim reads
    model_fields = set(struct_fields(src["src/sea_ice_model.jl"], "SeaIceModel"))
    used = set(re.findall(r"\bmodel\.(\w+)", shim))
    check(used <= model_fields, f"model.<field> reads exist in SeaIceModel: unknown {sorted(used - model_fields)}")
    dyn_fields = set(struct_fields(src["src/SeaIceDynamics/sea_ice_momentum_equations.jl"], "SeaIceMomentumEquation"))
    used = set(re.findall(r"\bdyn\.(\w+)", shim))
    check(used <= dyn_fields, f"dyn.<field> reads exist in SeaIceMomentumEquation: unknown {sorted(used - dyn_fields)}")
    rh_fields = set(struct_fields(src["src/Rheologies/elasto_visco_plastic_rheology.jl"], "ElastoViscoPlasticRheology"))
    used = set(re.findall(r"\br\.(\w+)", shim))
    check(used <= rh_fields, f"r.<field> reads exist in ElastoViscoPlasticRheology: unknown {sorted(used - rh_fields)}")
    print("check_julia_shim:", "consistent with the reference" if ok else "INCONSISTENT")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
