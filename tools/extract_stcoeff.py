#!/usr/bin/env python3
"""Cuts the accommodation ("sticking") coefficients of st_coeff_a / st_coeff_t out of the reference source (kpp.f90:857-1038, 664-851;
SURVEY.md §8 f3: liq_parm recomputes them for every layer in every time step, kpp.f90:614, 634; fast_k_mt_x reads them) ->
mistra_amd/mech/<mech>.stcoeff (what the library loads) and <mech>.stcoeff.json (readable; tests, oracle/rates_py.py).

The routine is, inside `do k=2,nf`, four local factors of the layer's temperature (tcorr, RT, CoRT, zexp2) and one assignment per species,

    alpha(ind_X,k) = <number> | <expression of t(k), the local factors, exp, CoR, R> | alpha(ind_Y,k) | a_n2o5(k,1)

some of them under `if (lpJoyce14bc)` / `if (.not.lpBuxmann15alph)` (namelist switches, module config), behind `alpha(:,:) = 0.1_dp` and
followed by `alpha(j,k) = min(1.d0,alpha(j,k))`.  Every assignment becomes a postfix program in the format of the rate-constant tables
(tools/extract_rates.py: the device evaluator of mistra_amd/csrc/rates.hip runs both) over the input vector

    env = [ t(k), cw(1,k), cm(1,k), sion1(13,1,k), sion1(14,1,k) ]          (the last four: what a_n2o5(k,1) reads, kpp.f90:8377-8425)

with the local factors substituted where they are used (the same operations on the same operands: the same values), literals folded in the
kind Fortran gives them, one table per setting of the two switches (variant = lpJoyce14bc + 2*lpBuxmann15alph).  The extractor refuses a
statement of another shape, so a change of the reference cannot slip through unnoticed.

    python tools/extract_stcoeff.py            (output committed; tests/test_rates.py checks that it is up to date)
"""
import json
import os
import re
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from extract_liq import body  # noqa: E402
from extract_rates import OP, OUT, REF, Parser, emit, fold, parameters  # noqa: E402

ROUTINES = {"aer": "st_coeff_a", "tot": "st_coeff_t"}
ENV = ["t", "cw1", "cm1", "sion1_13", "sion1_14"]
FUNC_ID = {"exp": (26, 1), "a_n2o5": (27, 4), "min": (28, 2)}      # ids behind the rate laws' (mistra_amd/csrc/rates.hip)
SWITCHES = ("lpjoyce14bc", "lpbuxmann15alph")


def module_constant(name):
    """a real(kind=dp) parameter of module constants"""
    text = open(os.path.join(REF, "constants.f90"), errors="replace").read()
    m = re.search(r"parameter\s*::\s*%s\s*=\s*([\d.deDE+-]+)_dp" % name, text, re.I)
    return float(m.group(1).lower().replace("d", "e"))


def subst(node, env):
    """replace variables / array references by their trees"""
    if node[0] == "var":
        if node[1] in env:
            return env[node[1]]
        raise ValueError("unknown name %r" % node[1])
    if node[0] == "neg":
        return ("neg", subst(node[1], env))
    if node[0] == "bin":
        return ("bin", node[1], subst(node[2], env), subst(node[3], env))
    if node[0] == "ref":
        name, args = node[1], node[2]
        if name == "t" and args == [("var", "k")]:
            return ("var", "t")
        if name == "alpha" and len(args) == 2 and args[1] == ("var", "k") and args[0][0] == "num":
            j = int(args[0][2])
            if j not in env["__alpha__"]:
                raise ValueError("alpha(%d,k) is read before it is set" % j)
            return env["__alpha__"][j]
        if name == "a_n2o5":
            if args != [("var", "k"), ("num", "int", 1)]:
                raise ValueError("a_n2o5 is expected as a_n2o5(k,1)")
            return ("ref", "a_n2o5", [("var", v) for v in ENV[1:]])
        if name == "exp" and len(args) == 1:
            return ("ref", "exp", [subst(args[0], env)])
        raise ValueError("reference of a shape this extractor does not know: %r" % (node,))
    return node


def extract(mech):
    params = parameters(os.path.join(REF, "%s_Parameters.h" % mech))
    name = ROUTINES[mech]
    cal, R = module_constant("cal15"), module_constant("gas_const")      # USE constants, ONLY : cal => cal15, R => gas_const
    lines = body(name)
    variants = []
    for variant in range(4):
        flag = {SWITCHES[0]: bool(variant & 1), SWITCHES[1]: bool(variant & 2)}
        names = {"cal": ("num", "double", cal), "r": ("num", "double", R), "cor": ("num", "double", cal / R), "__alpha__": {}}
        conds, in_loop, seen = [], False, {"default": False, "clamp": False, "cor": False}
        for no, l in lines:
            s = l.strip()
            low = s.lower()
            if re.match(r"real\s*\(kind=dp\)\s*,\s*parameter\s*::\s*cor\s*=\s*cal/r$", low):
                seen["cor"] = True
                continue
            if re.match(r"alpha\(:,:\)\s*=\s*0\.1_dp$", low):
                seen["default"] = True
                continue
            if re.match(r"do\s+k\s*=\s*2\s*,\s*nf$", low):
                in_loop = not seen["clamp"] and not names["__alpha__"]      # the first such loop holds the assignments, the second the clamp
                continue
            if re.match(r"alpha\(j,k\)\s*=\s*min\(1\.d0,alpha\(j,k\)\)$", low):
                seen["clamp"] = True
                continue
            m = re.match(r"if\s*\(\s*(\.not\.)?\s*(\w+)\s*\)\s*then$", low)
            if m:
                if m.group(2) not in flag:
                    raise ValueError("%s line %d: condition on %s" % (name, no, m.group(2)))
                conds.append(flag[m.group(2)] != bool(m.group(1)))
                continue
            if low == "else":
                conds[-1] = not conds[-1]
                continue
            if re.match(r"end\s*if$", low):
                conds.pop()
                continue
            m = re.match(r"(tcorr|rt|cort|zexp2)\s*=\s*(.*)$", low)
            if m and in_loop:
                names[m.group(1)] = subst(Parser(m.group(2), params).expr(), names)
                continue
            m = re.match(r"alpha\(\s*(ind_\w+)\s*,\s*k\s*\)\s*=\s*(.*)$", low)
            if m:
                if not in_loop:
                    raise ValueError("%s line %d: assignment outside the layer loop" % (name, no))
                if all(conds):
                    names["__alpha__"][params[m.group(1)]] = fold(subst(Parser(m.group(2), params).expr(), names))
                continue
            if "=" in s and re.search(r"\balpha\s*\(", low) and not re.match(r"(common|real)", low):
                raise ValueError("%s line %d: statement on alpha of a shape this extractor does not know: %s" % (name, no, s))
        assert all(seen.values()) and not conds, (name, seen, conds)
        progs = []
        for j in range(1, params["nspec"] + 1):
            prog, funcs = [["num", 1.0]], set()      # min(1.d0, alpha(j,k))
            emit(names["__alpha__"].get(j, ("num", "double", 0.1)), prog, funcs)
            prog.append(["call", "min", 2])
            progs.append(prog)
        variants.append({"lpJoyce14bc": flag[SWITCHES[0]], "lpBuxmann15alph": flag[SWITCHES[1]], "set": sorted(names["__alpha__"]), "programs": progs})
    return {"mech": mech, "nspec": params["nspec"], "source": "kpp.f90: %s (a_n2o5: kpp.f90:8377)" % name, "env": ENV, "first_layer": 2,
            "variants": variants}


def binary(table):
    """four tables back to back, each in the format of <mech>.rates ('KRAT' v2; mistra_amd/csrc/capi.cpp: RatesTable::load):
    int32 {magic, 2, nout, nenv, nconst, nwords, nfslot = 0, variant} | double consts | int32 offs[nout+1] | int32 words"""
    slot = {n: i for i, n in enumerate(table["env"])}
    out = b""
    for v, var in enumerate(table["variants"]):
        consts, words, offs = [], [], [0]
        for prog in var["programs"]:
            for t in prog:
                if t[0] == "num":
                    consts.append(float(t[1]))
                    words.append(OP["const"] | ((len(consts) - 1) << 8))
                elif t[0] == "var":
                    words.append(OP["env"] | (slot[t[1]] << 8))
                elif t[0] == "call":
                    fid, nargs = FUNC_ID[t[1]]
                    assert nargs == t[2], t
                    words.append(OP["call"] | (fid << 8))
                else:
                    words.append(OP[t[0]])
            offs.append(len(words))
        out += struct.pack("<8i", 0x5441524B, 2, table["nspec"], len(table["env"]), len(consts), len(words), 0, v)
        out += np.asarray(consts, np.float64).tobytes() + np.asarray(offs, np.int32).tobytes() + np.asarray(words, np.int32).tobytes()
    return out


def main():
    check = "--check" in sys.argv
    ok = True
    for mech in ("aer", "tot"):
        table = extract(mech)
        js, bn = json.dumps(table, separators=(",", ":")), binary(table)
        pj, pb = os.path.join(OUT, mech + ".stcoeff.json"), os.path.join(OUT, mech + ".stcoeff")
        if check:
            same = os.path.exists(pj) and open(pj).read() == js and os.path.exists(pb) and open(pb, "rb").read() == bn
            ok = ok and same
            print(mech, "up to date" if same else "DIFFERS from what the extractor writes")
        else:
            open(pj, "w").write(js)
            open(pb, "wb").write(bn)
            print(mech, [len(v["set"]) for v in table["variants"]], "species set per variant of", table["nspec"], "->", pb, len(bn), "bytes")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
