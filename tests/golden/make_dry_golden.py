#!/usr/bin/env python3
"""tests/golden/dryrates.npz from oracle/_ref/capture_dry_BTZ96.bin: dry_rates_g / dry_rates_a / dry_rates_t calls of the RUNNING reference model
(oracle/capture_dry_wrap.f90 around liq_parm's calls, namelist.BTZ96 with chem=T): per recorded layer what the routines read — temperature, mean free
path, the dry radii rcd(1:2), the mean molecular speeds (a, t) or the Henry entries before the call (g) of the four species of their idr list — and the
xkmtd, xeq(HNO3) and Henry entries they leave.  Data only."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "..", "oracle", "_ref")
WHAT = ("reference namelist.BTZ96 (chem=F -> T, netcdf=F); MISTRA_RUN_TAG=_dry MISTRA_COLUMN_MINUTES=4 oracle/capture_run.sh BTZ96 1 "
        "MISTRA_CAPTURE_DRY_FILE=... MISTRA_CAPTURE_DRY_SKIP=3 _EVERY=12 _MAX=2 _LAYERS=8")


def main():
    raw = open(os.path.join(REF, "capture_dry_BTZ96.bin"), "rb").read()
    off, per = 0, {1: [], 2: [], 3: []}
    while off < len(raw):
        h = np.frombuffer(raw, np.int32, 4, off); off += 16
        assert h[0] == 0x52595244
        d = np.frombuffer(raw, np.float64, 25, off).copy(); off += 200
        per[int(h[1])].append((int(h[2]), d))
    info = open(os.path.join(REF, "BUILD_INFO")).read().replace("\n", "; ")
    out = dict(provenance=np.array(WHAT + "; " + info))
    for name, r in (("gas", 1), ("aer", 2), ("tot", 3)):
        rs = per[r]
        out[name + "_k"] = np.array([k for k, _ in rs], np.int32)
        for key, sl in (("tt", slice(0, 1)), ("freep", slice(1, 2)), ("rcd", slice(2, 4)), ("vmean4", slice(4, 8)), ("henry4_before", slice(8, 12)),
                        ("xkmtd", slice(12, 20)), ("xeq", slice(20, 21)), ("henry4", slice(21, 25))):
            a = np.stack([d[sl] for _, d in rs])
            out[name + "_" + key] = a.reshape(len(rs), 2, 4) if key == "xkmtd" else a[:, 0] if a.shape[1] == 1 else a
    path = os.path.join(HERE, "dryrates.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", {n: out[n + "_k"].tolist() for n in ("gas", "aer", "tot")}, "rcd > 0:", {n: int((out[n + "_rcd"] > 0).sum()) for n in ("gas", "aer", "tot")})


if __name__ == "__main__":
    main()
