#!/usr/bin/env python3
"""tests/golden/kmt_<mech>.npz from oracle/_ref/capture_kmt_BTZ96.bin: fast_k_mt_a / fast_k_mt_t calls of the RUNNING reference model
(oracle/capture_kmt_wrap.f90 around liq_parm's calls, namelist.BTZ96 with chem=T), per recorded layer what the routine reads and
xkmt(:,:,k), vt(:,k) before and after.  Data only."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "..", "oracle", "_ref")
WHAT = ("reference namelist.BTZ96 (chem=F -> T, netcdf=F), model minutes 1-7; MISTRA_RUN_TAG=_kmt MISTRA_COLUMN_MINUTES=7 oracle/capture_run.sh BTZ96 1 "
        "MISTRA_CAPTURE_KMT_FILE=... MISTRA_CAPTURE_KMT_SKIP_a=1 _EVERY_a=2 _MAX_a=2 _SKIP_t=1 _EVERY_t=2 _MAX_t=2 MISTRA_CAPTURE_KMT_LAYERS=4")


def main():
    raw = open(os.path.join(REF, "capture_kmt_BTZ96.bin"), "rb").read()
    off, per = 0, {1: [], 2: []}
    while off < len(raw):
        h = np.frombuffer(raw, np.int32, 10, off); off += 40
        assert h[0] == 0x4B4D5443
        variant, k, nspec, nka, nkt, nkc, ka, ifeed, nkc_l = (int(x) for x in h[1:])
        kw = np.frombuffer(raw, np.int32, nka, off).copy(); off += 4 * nka
        n = 2 * nkt * nka + 2 * nkc + 1 + 2 * nspec + 2 * nspec * nkc + 2 + 2 * nkc
        d = np.frombuffer(raw, np.float64, n, off); off += 8 * n
        p = 0
        def take(m):
            nonlocal p
            v = d[p:p + m].copy(); p += m
            return v
        r = dict(k=k, ka=ka, ifeed=ifeed, nkc_l=nkc_l, kw=kw, rq=take(nkt * nka).reshape(nka, nkt), ff=take(nkt * nka).reshape(nka, nkt), cw=take(nkc), cm=take(nkc),
                 freep=take(1)[0], alpha=take(nspec), vmean=take(nspec), xkmt_before=take(nspec * nkc).reshape(nkc, nspec), xkmt_after=take(nspec * nkc).reshape(nkc, nspec),
                 t=take(1)[0], p=take(1)[0], vt_before=take(nkc), vt_after=take(nkc))
        per[variant].append(r)
    info = open(os.path.join(REF, "BUILD_INFO")).read().replace("\n", "; ")
    for variant, mech in ((1, "aer"), (2, "tot")):
        rs = per[variant]
        if not rs:
            continue
        assert all(np.array_equal(r["rq"], rs[0]["rq"]) and np.array_equal(r["kw"], rs[0]["kw"]) and r["ka"] == rs[0]["ka"] for r in rs)
        out = {key: np.stack([r[key] for r in rs]) for key in ("ff", "cw", "cm", "alpha", "vmean", "xkmt_before", "xkmt_after", "vt_before", "vt_after")}
        out.update(freep=np.array([r["freep"] for r in rs]), t=np.array([r["t"] for r in rs]), p=np.array([r["p"] for r in rs]), k=np.array([r["k"] for r in rs], np.int32), rq=rs[0]["rq"], kw=rs[0]["kw"], ka=np.int32(rs[0]["ka"]),
                   ifeed=np.int32(rs[0]["ifeed"]), nkc_l=np.int32(rs[0]["nkc_l"]), provenance=np.array(WHAT + "; " + info))
        path = os.path.join(HERE, "kmt_%s.npz" % mech)
        np.savez_compressed(path, **out)
        changed = (out["xkmt_after"] != out["xkmt_before"]).sum(axis=(1, 2))
        print(path, os.path.getsize(path), "bytes;", len(rs), "layers", out["k"].tolist(), "coefficients rewritten per layer", changed.tolist(),
              "vt rewritten per layer", (out["vt_after"] != out["vt_before"]).sum(axis=1).tolist(), "dry bins with vt", int(((out["cm"] <= 0) & (out["cw"] > 0)).sum()))


if __name__ == "__main__":
    main()
