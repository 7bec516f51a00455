#!/usr/bin/env python3
"""tests/golden/cwrc.npz from oracle/_ref/capture_cwrc_BTZ96.bin: cw_rc and dry_cw_rc calls of the RUNNING reference model (oracle/capture_cwrc_wrap.f90
around liq_parm's calls, namelist.BTZ96 with chem=T): per recorded layer what the routines read — the layer's particle spectrum ff, relative humidity,
the cloud flags; per call the grid rq, e, kw, ka and the crystallisation / deliquescence humidities — and the rc, cw, cm, conv2 they leave.  Data only."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "..", "oracle", "_ref")
WHAT = ("reference namelist.BTZ96 (chem=F -> T, netcdf=F); MISTRA_RUN_TAG=_cwrc MISTRA_COLUMN_MINUTES=4 oracle/capture_run.sh BTZ96 1 "
        "MISTRA_CAPTURE_CWRC_FILE=... MISTRA_CAPTURE_CWRC_SKIP=3 _EVERY=12 _MAX=2 _LAYERS=6")


def main():
    raw = open(os.path.join(REF, "capture_cwrc_BTZ96.bin"), "rb").read()
    off, calls = 0, {1: [], 2: []}
    while off < len(raw):
        h = np.frombuffer(raw, np.int32, 10, off); off += 40
        assert h[0] == 0x43525743
        routine, layers, nkt, nka, nkc, ka, ifeed, kinv = (int(x) for x in h[1:9])
        g = nkt * nka
        d = np.frombuffer(raw, np.float64, g + nkt + nka + 4, off).copy(); off += 8 * d.size
        call = dict(ka=ka, ifeed=ifeed, kinv=kinv, rq=d[:g].reshape(nka, nkt), e=d[g:g + nkt], kw=d[g + nkt:g + nkt + nka].astype(np.int32), crys4=d[-4:], layers=[])
        for _ in range(layers):
            n = 2 + nkc + g + 4 * nkc
            d = np.frombuffer(raw, np.float64, n, off).copy(); off += 8 * n
            p = 2 + nkc
            call["layers"].append(dict(k=int(d[0]), feu=d[1], cloud=d[2:p].astype(np.int32), ff=d[p:p + g].reshape(nka, nkt), rc=d[p + g:p + g + nkc],
                                       cw=d[p + g + nkc:p + g + 2 * nkc], cm=d[p + g + 2 * nkc:p + g + 3 * nkc], conv2=d[p + g + 3 * nkc:]))
        calls[routine].append(call)
    info = open(os.path.join(REF, "BUILD_INFO")).read().replace("\n", "; ")
    out = dict(provenance=np.array(WHAT + "; " + info))
    c0 = calls[1][0]
    for key in ("rq", "e", "kw", "crys4"):      # the grid does not change between calls (checked)
        for c in calls[1] + calls[2]:
            assert np.array_equal(c[key], c0[key])
        out[key] = c0[key]
    out["ka"], out["ifeed"], out["kinv"] = np.int32(c0["ka"]), np.int32(c0["ifeed"]), np.int32(c0["kinv"])
    for name, r in (("wet", 1), ("dry", 2)):
        ls = [l for c in calls[r] for l in c["layers"]]
        for key in ("k", "feu", "cloud", "ff", "rc", "cw", "cm", "conv2"):
            out["%s_%s" % (name, key)] = np.stack([np.asarray(l[key]) for l in ls])
    path = os.path.join(HERE, "cwrc.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes; cw_rc layers", out["wet_k"].tolist(), "bins switched on", (out["wet_conv2"] > 0).sum(axis=1).tolist(),
          "; dry_cw_rc layers", out["dry_k"].tolist(), "ka", int(out["ka"]), "ifeed", int(out["ifeed"]))


if __name__ == "__main__":
    main()
