#!/usr/bin/env python3
"""Cuts the table of v_mean_a / v_mean_t out of the reference source (kpp.f90:1472-1670, 1268-1465; SURVEY.md §8 f3: the mean molecular
speeds liq_parm recomputes for every layer in every time step, kpp.f90:612, 632) -> mistra_amd/mech/<mech>.vmean (what the library loads)
and <mech>.vmean.json (readable; tests, oracle/liq_py.py).

The routine is a list of one-line assignments per species inside a loop over the layers 1..nmaxf,

    vmean(ind_X,k) = func(<molar mass in kg/mol>,k)          func(a,k) = sqrt(tt(k)/a)*4.60138

behind `vmean(:,:) = 0._dp`.  The extractor refuses any statement on vmean of another shape, so a change of the reference cannot slip
through unnoticed; a species assigned twice keeps the LAST value, as in the reference.  Numbers are folded in the kind Fortran gives them
(tools/extract_rates.py: Parser, fold): the factor 4.60138 is a default-real literal, i.e. the double nearest to its float32 value.

    python tools/extract_vmean.py            (output committed; tests/test_pack.py checks that it is up to date)
"""
import json
import os
import re
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from extract_liq import body, const  # noqa: E402
from extract_rates import OUT, REF, Parser, parameters  # noqa: E402

ROUTINES = {"aer": "v_mean_a", "tot": "v_mean_t"}


def extract(mech):
    params = parameters(os.path.join(REF, "%s_Parameters.h" % mech))
    name = ROUTINES[mech]
    mass, coef, seen_zero, seen_loop = {}, None, False, False
    for no, l in body(name):
        s = l.strip()
        m = re.match(r"vmean\(\s*(ind_\w+)\s*,\s*k\s*\)\s*=\s*func\(\s*([^,]+),\s*k\s*\)$", s, re.I)
        if m:
            mass[params[m.group(1).lower()]] = const(Parser(m.group(2), params).expr(), s)
            continue
        m = re.match(r"func\(a,k\)\s*=\s*sqrt\(tt\(k\)/a\)\*([\d.deDE_dp+-]+)$", s, re.I)
        if m:
            coef = const(Parser(m.group(1), params).expr(), s)
            continue
        if re.match(r"vmean\(:,:\)\s*=\s*0\._dp$", s, re.I):
            seen_zero = True
        elif re.match(r"do\s+k\s*=\s*1\s*,\s*nmaxf$", s, re.I):
            seen_loop = True
        elif "=" in s and re.search(r"vmean\s*\(", s, re.I) and not re.match(r"(common|real)", s, re.I):
            raise ValueError("%s line %d: statement on vmean of a shape this extractor does not know: %s" % (name, no, s))
    assert coef is not None and seen_zero and seen_loop and mass, (name, coef, seen_zero, seen_loop, len(mass))
    return {"mech": mech, "nspec": params["nspec"], "source": "kpp.f90: %s" % name, "coef": coef, "entries": [[j, mass[j]] for j in sorted(mass)]}


def binary(table):
    """int32 header {magic 'VMNT', version 1, nspec, n, 0, 0, 0, 0} | double coef | int32 j[n] (1-based species) | double mass[n]"""
    e = table["entries"]
    out = struct.pack("<8i", 0x544E4D56, 1, table["nspec"], len(e), 0, 0, 0, 0) + struct.pack("<d", table["coef"])
    return out + np.array([x[0] for x in e], "<i4").tobytes() + np.array([x[1] for x in e], "<f8").tobytes()


def main():
    check = "--check" in sys.argv
    ok = True
    for mech in ("aer", "tot"):
        table = extract(mech)
        js, bn = json.dumps(table, separators=(",", ":")), binary(table)
        pj, pb = os.path.join(OUT, mech + ".vmean.json"), os.path.join(OUT, mech + ".vmean")
        if check:
            same = os.path.exists(pj) and open(pj).read() == js and os.path.exists(pb) and open(pb, "rb").read() == bn
            ok = ok and same
            print(mech, "up to date" if same else "DIFFERS from what the extractor writes")
        else:
            open(pj, "w").write(js)
            open(pb, "wb").write(bn)
            print(mech, len(table["entries"]), "mean molecular speeds of", table["nspec"], "species, factor %r ->" % table["coef"], pb, len(bn), "bytes")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
