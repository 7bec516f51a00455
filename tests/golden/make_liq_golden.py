#!/usr/bin/env python3
"""tests/golden/liq_<mech>.npz from oracle/_ref/capture_liq_BTZ96.bin: henry_x / equil_co_x / v_mean_x calls of the RUNNING reference model
(oracle/capture_liq_wrap.f90 around liq_parm's calls, namelist.BTZ96 with chem=T): per recorded layer what the routine reads and what
it leaves in henry(:,k), vmean(:,k), xkef(:,:,k), xkeb(:,:,k) (the latter two also before the call).  Data only."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "..", "oracle", "_ref")
WHAT = ("reference namelist.BTZ96 (chem=F -> T, netcdf=F), model minutes 1-7; MISTRA_RUN_TAG=_liq MISTRA_COLUMN_MINUTES=7 oracle/capture_run.sh BTZ96 1 "
        "MISTRA_CAPTURE_LIQ_FILE=... MISTRA_CAPTURE_LIQ_SKIP=2 _EVERY=20 _MAX=2 _LAYERS=8")


def main():
    raw = open(os.path.join(REF, "capture_liq_BTZ96.bin"), "rb").read()
    off, per = 0, {1: [], 2: [], 3: [], 4: [], 5: [], 6: []}
    while off < len(raw):
        h = np.frombuffer(raw, np.int32, 6, off); off += 24
        assert h[0] == 0x4C495143
        routine, k, nspec, nkc, j6 = (int(x) for x in h[1:])
        n = 7 + nspec if routine >= 7 else 1 + nspec if routine <= 2 or routine >= 5 else 1 + nkc + j6 * nkc + 4 * nspec * nkc
        d = np.frombuffer(raw, np.float64, n, off).copy(); off += 8 * n
        if routine >= 7:
            continue      # st_coeff_x: tests/golden/make_stcoeff_golden.py
        if routine <= 2 or routine >= 5:
            per[routine].append(dict(k=k, tt=d[0], henry=d[1:]))
        else:
            p = 1 + nkc
            xg = d[p:p + j6 * nkc].reshape(nkc, j6); p += j6 * nkc
            blocks = [d[p + i * nspec * nkc:p + (i + 1) * nspec * nkc].reshape(nkc, nspec) for i in range(4)]
            per[routine].append(dict(k=k, tt=d[0], conv2=d[1:1 + nkc], xgamma=xg, xkef_before=blocks[0], xkeb_before=blocks[1], xkef=blocks[2], xkeb=blocks[3]))
    info = open(os.path.join(REF, "BUILD_INFO")).read().replace("\n", "; ")
    for mech, rh, re_, rv in (("aer", 1, 3, 5), ("tot", 2, 4, 6)):
        hs, es, vs = per[rh], per[re_], per[rv]
        out = dict(henry_k=np.array([r["k"] for r in hs], np.int32), henry_tt=np.array([r["tt"] for r in hs]), henry=np.stack([r["henry"] for r in hs]),
                   equil_k=np.array([r["k"] for r in es], np.int32), equil_tt=np.array([r["tt"] for r in es]),
                   vmean_k=np.array([r["k"] for r in vs], np.int32), vmean_tt=np.array([r["tt"] for r in vs]), vmean=np.stack([r["henry"] for r in vs]),
                   provenance=np.array(WHAT + "; " + info))
        for key in ("conv2", "xgamma", "xkef_before", "xkeb_before", "xkef", "xkeb"):
            out[key] = np.stack([r[key] for r in es])
        path = os.path.join(HERE, "liq_%s.npz" % mech)
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), "bytes; henry layers", out["henry_k"].tolist(), "T %.1f..%.1f" % (out["henry_tt"].min(), out["henry_tt"].max()),
              "; v_mean layers", out["vmean_k"].tolist(), "species set", (out["vmean"] != 0).sum(axis=1).tolist(), "; equil layers", out["equil_k"].tolist(), "bins with conv2 > 0 per layer", (out["conv2"] > 0).sum(axis=1).tolist(),
              "entries changed", ((out["xkef"] != out["xkef_before"]) | (out["xkeb"] != out["xkeb_before"])).sum(axis=(1, 2)).tolist())


if __name__ == "__main__":
    main()
