#!/usr/bin/env python3
"""tests/golden/stcoeff_<mech>.npz from oracle/_ref/capture_liq_{BTZ96,Joyce2014}.bin: st_coeff_a / st_coeff_t calls of the RUNNING reference
model (oracle/capture_liq_wrap.f90 around liq_parm's calls; namelist.BTZ96 with chem=T: both switches off, droplet chemistry on;
namelist.Joyce2014_basecase: lpJoyce14bc = T, i.e. alpha(N2O5) = a_n2o5(k,1), on its dry aerosol (a_n2o5 = 0); namelist.BTZ96 with the one line
`lpJoyce14bc = T` added to the scratch copy — MISTRA_NAMELIST_SED of oracle/capture_run.sh —, so that a_n2o5 runs on wet, nitrate-bearing
aerosol): per recorded layer what the routine reads — t(k), cw(1,k),
cm(1,k), sion1(13:14,1,k), the two switches — and the alpha(:,k) it leaves.  Data only."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "..", "oracle", "_ref")
WHAT = ("BTZ96: as tests/golden/make_liq_golden.py; Joyce2014: MISTRA_RUN_TAG=_liqj MISTRA_COLUMN_MINUTES=3 oracle/capture_run.sh Joyce2014_basecase 1 "
        "MISTRA_CAPTURE_LIQ_FILE=... MISTRA_CAPTURE_LIQ_SKIP=2 _EVERY=5 _MAX=2 _LAYERS=8; BTZ96_joyce: MISTRA_RUN_TAG=_liqs MISTRA_COLUMN_MINUTES=3 "
        "MISTRA_NAMELIST_SED='s/^&mistra_cfg/\\&mistra_cfg\\n lpJoyce14bc = T/' oracle/capture_run.sh BTZ96 1 ... (the same capture settings)")


def records(path):
    raw = open(path, "rb").read()
    off = 0
    while off < len(raw):
        h = np.frombuffer(raw, np.int32, 6, off); off += 24
        assert h[0] == 0x4C495143
        routine, k, nspec, nkc, j6 = (int(x) for x in h[1:])
        n = 7 + nspec if routine >= 7 else 1 + nspec if routine <= 2 or routine >= 5 else 1 + nkc + j6 * nkc + 4 * nspec * nkc
        d = np.frombuffer(raw, np.float64, n, off).copy(); off += 8 * n
        if routine >= 7:
            yield routine, k, d


def main():
    per = {7: [], 8: []}
    for case in ("BTZ96", "Joyce2014", "BTZ96_joyce"):
        for routine, k, d in records(os.path.join(REF, "capture_liq_%s.bin" % case)):
            per[routine].append((case, k, d))
    info = open(os.path.join(REF, "BUILD_INFO")).read().replace("\n", "; ")
    for mech, r in (("aer", 7), ("tot", 8)):
        rs = per[r]
        out = dict(case=np.array([c for c, _, _ in rs]), k=np.array([k for _, k, _ in rs], np.int32), env=np.stack([d[:5] for _, _, d in rs]),
                   lp_joyce14bc=np.array([int(d[5]) for _, _, d in rs], np.int32), lp_buxmann15alph=np.array([int(d[6]) for _, _, d in rs], np.int32),
                   alpha=np.stack([d[7:] for _, _, d in rs]), provenance=np.array(WHAT + "; " + info))
        path = os.path.join(HERE, "stcoeff_%s.npz" % mech)
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), "bytes;", len(rs), "layers; cases", sorted(set(out["case"].tolist())), "lpJoyce14bc", out["lp_joyce14bc"].tolist(),
              "T %.1f..%.1f" % (out["env"][:, 0].min(), out["env"][:, 0].max()), "cw1 > 0 in", int((out["env"][:, 1] > 0).sum()))


if __name__ == "__main__":
    main()
