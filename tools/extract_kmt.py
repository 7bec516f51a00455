#!/usr/bin/env python3
"""Extracts the species list of the reference's mass-transfer-coefficient routines fast_k_mt_a / fast_k_mt_t (kpp.f90:2683-2947 |
2421-2676) into data: mistra_amd/mech/<mech>.kmt.json = {"nx": 50, "lex": [C index (1-based) of each exchanged
species, in the order of the routine's DATA statement], "names": [...], "nka": 70, "nkt": 70, "nkc": 4}.  Needs the reference tree."""
import json
import os
import re
import sys

REF = os.environ.get("MISTRA_REFERENCE_SRC", "/root/reference/src")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mistra_amd", "mech")


def main():
    text = open(os.path.join(REF, "kpp.f90"), errors="replace").read()
    gp = open(os.path.join(REF, "global_params.f90"), errors="replace").read()
    dims = {k: int(re.search(r"integer,\s*parameter\s*::\s*%s\s*=\s*(\d+)" % k, gp).group(1)) for k in ("nka", "nkt", "nkc")}
    for mech, sub in (("aer", "fast_k_mt_a"), ("tot", "fast_k_mt_t")):
        body = text[text.index("subroutine %s " % sub):text.index("end subroutine %s" % sub)]
        m = re.search(r"data\s+lex\s*/(.*?)/", body, re.S)
        names = [x for x in re.sub(r"[&\s]", "", re.sub(r"!.*", "", m.group(1))).split(",") if x]
        nx = int(re.search(r"integer,\s*parameter\s*::\s*nx\s*=\s*(\d+)", body).group(1))
        assert len(names) == nx, (len(names), nx)
        par = {}
        for pm in re.finditer(r"PARAMETER\s*\(\s*(\w+)\s*=\s*(\d+)\s*\)", open(os.path.join(REF, mech + "_Parameters.h"), errors="replace").read()):
            par[pm.group(1).lower()] = int(pm.group(2))
        lex = [par[nm.lower()] for nm in names]
        tab = dict(mech=mech, routine=sub, nx=nx, lex=lex, names=names, **dims)
        json.dump(tab, open(os.path.join(OUT, mech + ".kmt.json"), "w"), separators=(",", ":"))
        with open(os.path.join(OUT, mech + ".kmt"), "w") as f:      # what the library reads: nx nka nkt nkc, then the indices
            f.write("%d %d %d %d\n%s\n" % (nx, dims["nka"], dims["nkt"], dims["nkc"], " ".join(str(i) for i in lex)))
        print(mech, nx, "species", lex[:6], "...")


if __name__ == "__main__":
    main()
