#!/usr/bin/env python3
"""Cuts the tables of henry_a / henry_t and equil_co_a / equil_co_t out of the reference source (kpp.f90:1676-2145, 2954-3363; SURVEY.md
§8 f3) -> mistra_amd/mech/<mech>.liq (what the library loads) and <mech>.liq.json (readable; tests, oracle/liq_py.py).

Both routines are lists of one-line assignments per species inside a loop over the layers (equil_co_x: and over the bins):

    henry(ind_X,k) = <number> | func3(a0,b0)                 func3(a0,b0) = a0*exp(b0*Tfact), Tfact = 1/tt(k) - 3.3540d-3
    ... then, every species:  henry > 0  ->  1 / (henry * FCT),  FCT = 0.0820577 * tt(k)
    xkef(ind_X,kc,k) = f1 * f2 * ...        a factor is a number, funa(a0,b0,k) = a0*exp(b0*(1/tt(k)-3.354d-3)), cv2 (= conv2(kc,k)) or
    xkeb(ind_X,kc,k) = f1 * f2 * ...        xgamma(i,kc,k); products are formed left to right as Fortran does; the whole list applies where
                                            cv2 > 0, elsewhere every entry of the bin is set to zero

The extractor refuses any assignment of another shape, so a change of the reference cannot slip through unnoticed.  Numbers are
folded in the kind Fortran gives them (tools/extract_rates.py: Parser, fold).

    python tools/extract_liq.py            (output committed; tests/test_pack.py checks that it is up to date)
"""
import json
import os
import re
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from extract_rates import OUT, REF, Parser, fold, parameters  # noqa: E402

ROUTINES = {"aer": ("henry_a", "equil_co_a"), "tot": ("henry_t", "equil_co_t")}


def body(name):
    """code lines of one subroutine of kpp.f90, comments stripped"""
    lines = open(os.path.join(REF, "kpp.f90"), errors="replace").read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"\s*subroutine\s+%s\b" % name, l, re.I))
    out = []
    for i in range(start, len(lines)):
        l = lines[i].split("!")[0].rstrip()
        if l.strip():
            out.append((i + 1, l))
        if re.match(r"\s*end\s+subroutine\s+%s\b" % name, lines[i], re.I):
            return out
    raise ValueError(name)


def const(node, what):
    node = fold(node)
    if node[0] != "num":
        raise ValueError("%s is not a constant: %r" % (what, node))
    return float(node[2])


def factors(node, out, where):
    """left-to-right factor list of a product tree"""
    if node[0] == "bin" and node[1] == "*":
        factors(node[2], out, where)
        right = node[3]
        if right[0] == "bin":
            raise ValueError("%s: a parenthesised product on the right of '*' is not a left-to-right chain" % where)
        factors(right, out, where)
    elif node[0] == "num":
        out.append(["num", float(node[2])])
    elif node[0] == "var" and node[1] == "cv2":
        out.append(["cv2"])
    elif node[0] == "ref" and node[1] == "funa" and len(node[2]) == 3 and node[2][2] == ("var", "k"):
        out.append(["funa", const(node[2][0], where), const(node[2][1], where)])
    elif node[0] == "ref" and node[1] == "xgamma" and len(node[2]) == 3 and node[2][1] == ("var", "kc") and node[2][2] == ("var", "k"):
        i = fold(node[2][0])
        if i[0] != "num" or i[1] != "int":
            raise ValueError("%s: xgamma index" % where)
        out.append(["xg", int(i[2])])
    else:
        raise ValueError("%s: factor of a shape this extractor does not know: %r" % (where, node))


def extract(mech):
    params = parameters(os.path.join(REF, "%s_Parameters.h" % mech))
    hname, ename = ROUTINES[mech]
    # ---- henry_x
    henry, tref, fct, seen_zero, seen_inv = {}, None, None, False, False
    for no, l in body(hname):
        s = l.strip()
        m = re.match(r"henry\(\s*(ind_\w+)\s*,\s*k\s*\)\s*=\s*(.*)$", s, re.I)
        if m:
            j = params[m.group(1).lower()]
            node = Parser(m.group(2), params).expr()
            if node[0] == "ref" and node[1] == "func3" and len(node[2]) == 2:
                entry = [j, const(node[2][0], s), const(node[2][1], s)]
            else:
                entry = [j, const(node, s), None]
            henry[j] = entry      # (a later assignment to the same species overrides an earlier one, as in the reference)
            continue
        m = re.match(r"tfact\s*=\s*1\.d0\s*/\s*tt\(k\)\s*-\s*([\d.deDE+-]+)$", s, re.I)
        if m:
            tref = const(Parser(m.group(1), params).expr(), s)
            continue
        m = re.match(r"fct\s*=\s*([\d.deDE_dp+-]+)\s*\*\s*tt\(k\)$", s, re.I)
        if m:
            fct = const(Parser(m.group(1), params).expr(), s)
            continue
        if re.match(r"henry\(:,:\)\s*=\s*0\._dp$", s, re.I):
            seen_zero = True
        elif re.match(r"henry\(j,k\)\s*=\s*1\._dp\s*/\s*\(henry\(j,k\)\*FCT\)$", s, re.I):
            seen_inv = True
        elif re.match(r"func3\(a0,b0\)\s*=\s*a0\*exp\(b0\*Tfact\)$", s, re.I):
            pass
        elif "=" in s and re.search(r"henry\s*\(", s, re.I) and not re.match(r"(common|real|if\s*\(henry\(j,k\)\.gt\.0\._dp\)\s*then)", s, re.I) and "xkmt" not in s:
            raise ValueError("%s line %d: statement on henry of a shape this extractor does not know: %s" % (hname, no, s))
    assert tref is not None and fct is not None and seen_zero and seen_inv, (hname, tref, fct, seen_zero, seen_inv)
    # ---- equil_co_x
    ef, eb, etref, nkc_eq, gate = {}, {}, None, None, False
    for no, l in body(ename):
        s = l.strip()
        m = re.match(r"(xkef|xkeb)\(\s*(ind_\w+)\s*,\s*kc\s*,\s*k\s*\)\s*=\s*(.*)$", s, re.I)
        if m:
            j = params[m.group(2).lower()]
            fl = []
            factors(fold(Parser(m.group(3), params).expr()), fl, "%s line %d" % (ename, no))
            (ef if m.group(1).lower() == "xkef" else eb)[j] = fl
            continue
        m = re.match(r"funa\(a0,b0,k\)\s*=\s*a0\*exp\(b0\*\(1/tt\(k\)-([\d.deDE+-]+)\)\)$", s, re.I)
        if m:
            etref = const(Parser(m.group(1), params).expr(), s)
            continue
        m = re.match(r"do\s+kc\s*=\s*1\s*,\s*(\w+)$", s, re.I)
        if m:
            nkc_eq = 4 if m.group(1).lower() == "nkc" else int(m.group(1))
            continue
        if re.match(r"if\s*\(cv2\.gt\.0\._dp\)\s*then$", s, re.I):
            gate = True
        elif re.match(r"xke[fb]\(:,kc,k\)\s*=\s*0\._dp$", s, re.I) or re.match(r"cv2\s*=\s*conv2\(kc,k\)$", s, re.I):
            pass
        elif "=" in s and re.search(r"xke[fb]\s*\(", s, re.I) and not re.match(r"(common|real)", s, re.I) and "xkmt" not in s:
            raise ValueError("%s line %d: statement on xkef / xkeb of a shape this extractor does not know: %s" % (ename, no, s))
    assert etref is not None and nkc_eq is not None and gate and sorted(ef) == sorted(eb), (ename, etref, nkc_eq, gate)
    table = {"mech": mech, "nspec": params["nspec"], "source": "kpp.f90: %s, %s" % (hname, ename),
             "henry": {"tref": tref, "fct": fct, "entries": [henry[j] for j in sorted(henry)]},
             "equil": {"tref": etref, "nkc": nkc_eq, "entries": [[j, ef[j], eb[j]] for j in sorted(ef)]}}
    return table


KIND = {"num": 0, "funa": 1, "cv2": 2, "xg": 3}


def binary(table):
    """int32 header {magic 'LIQT', version 1, nspec, n_henry, n_equil, nkc_eq, nfac, 0} | doubles {henry tref, henry fct, equil tref}
    | henry: int32 j[nh], int32 kind[nh] (0 number, 1 func3), double a0[nh], b0[nh]
    | equil: int32 j[ne], int32 foff[ne+1], int32 boff[ne+1] (into the factor arrays), int32 fkind[nfac], int32 farg[nfac], double fa[nfac], fb[nfac]"""
    h, e = table["henry"]["entries"], table["equil"]["entries"]
    fk, fa_i, fa, fb, foff, boff = [], [], [], [], [], []
    for _, fprog, bprog in e:
        for offs, prog in ((foff, fprog), (boff, bprog)):
            offs.append(len(fk))
            for f in prog:
                fk.append(KIND[f[0]])
                fa_i.append(f[1] if f[0] == "xg" else 0)
                fa.append(f[1] if f[0] in ("num", "funa") else 0.0)
                fb.append(f[2] if f[0] == "funa" else 0.0)
    # (offsets interleave: the f program of entry i ends where its b program starts, the b program ends where entry i+1's f program starts)
    foff.append(len(fk))
    boff.append(len(fk))
    out = struct.pack("<8i", 0x5451494C, 1, table["nspec"], len(h), len(e), table["equil"]["nkc"], len(fk), 0)
    out += struct.pack("<3d", table["henry"]["tref"], table["henry"]["fct"], table["equil"]["tref"])
    out += np.array([x[0] for x in h], "<i4").tobytes() + np.array([0 if x[2] is None else 1 for x in h], "<i4").tobytes()
    out += np.array([x[1] for x in h], "<f8").tobytes() + np.array([0.0 if x[2] is None else x[2] for x in h], "<f8").tobytes()
    out += np.array([x[0] for x in e], "<i4").tobytes() + np.array(foff, "<i4").tobytes() + np.array(boff, "<i4").tobytes()
    out += np.array(fk, "<i4").tobytes() + np.array(fa_i, "<i4").tobytes() + np.array(fa, "<f8").tobytes() + np.array(fb, "<f8").tobytes()
    return out


def main():
    check = "--check" in sys.argv
    out_dir = OUT
    ok = True
    for mech in ("aer", "tot"):
        table = extract(mech)
        js = json.dumps(table, separators=(",", ":"))
        bn = binary(table)
        pj, pb = os.path.join(out_dir, mech + ".liq.json"), os.path.join(out_dir, mech + ".liq")
        if check:
            same = os.path.exists(pj) and open(pj).read() == js and os.path.exists(pb) and open(pb, "rb").read() == bn
            ok = ok and same
            print(mech, "up to date" if same else "DIFFERS from what the extractor writes")
        else:
            open(pj, "w").write(js)
            open(pb, "wb").write(bn)
            nf = sum(len(f) + len(b) for _, f, b in table["equil"]["entries"])
            print(mech, len(table["henry"]["entries"]), "Henry constants (%d with a temperature law)," % sum(1 for x in table["henry"]["entries"] if x[2] is not None),
                  len(table["equil"]["entries"]), "equilibria,", nf, "factors, bins 1..%d ->" % table["equil"]["nkc"], pb, len(bn), "bytes")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
