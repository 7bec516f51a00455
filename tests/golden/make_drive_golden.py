#!/usr/bin/env python3
"""Generates tests/golden/drive_<mech>.npz from oracle/_ref/capture_drive_<case>.bin: whole gas_drive / aer_drive / tot_drive calls of
the RUNNING reference model, recorded by oracle/capture_drive_wrap.f90 (linked into the capture build with -Wl,--wrap=x_drive_).

Per kept call: the driver's arguments, the layer's s1 / s3 / sl1 / sion1 before and after, the species index maps of module
gas_common, C as the driver handed it to INTEGRATE_x (= the result of its pack half), RCONST, C after the integration, bgs(:,:,k) and
— for budget levels — bg(:,:,kl) before and after.  Data only; the capture command is in each file's `provenance`."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "..", "oracle", "_ref")
CASES = {
    "BTZ96": ("capture_drive_BTZ96.bin",
              "reference namelist.BTZ96 (chem=F -> T, netcdf=F), model minutes 1-12 (night, stratus: gas, aer and tot layers); "
              "MISTRA_RUN_TAG=_drive MISTRA_COLUMN_MINUTES=12 oracle/capture_run.sh BTZ96 1 MISTRA_CAPTURE_DRIVE_FILE=... "
              "MISTRA_CAPTURE_DRIVE_SKIP_g=400 _EVERY_g=139 _MAX_g=24 _SKIP_a=300 _EVERY_a=67 _MAX_a=24 _SKIP_t=400 _EVERY_t=89 _MAX_t=24"),
}
J2, J6, NKC, NBGS = 121, 55, 4, 122      # global_params.f90:96-103; common /budgs/ bgs(2,122,n)


class Reader:
    def __init__(self, raw):
        self.raw, self.off = raw, 0

    def i32(self, n):
        v = np.frombuffer(self.raw, np.int32, n, self.off).copy()
        self.off += 4 * n
        return v

    def f64(self, n):
        v = np.frombuffer(self.raw, np.float64, n, self.off).copy()
        self.off += 8 * n
        return v


def read_records(path):
    rd = Reader(open(path, "rb").read())
    recs = []
    while rd.off < len(rd.raw):
        h = rd.i32(9)
        assert h[0] == 0x44524956, hex(h[0])
        mech, k, j1, j5, nvar, nfix, nreact, nargs = (int(x) for x in h[1:])
        r = dict(mech=("gas", "aer", "tot")[mech], k=k, j1=j1, j5=j5, args=rd.f64(nargs))
        r["gas_m2k"] = rd.i32(2 * j1).reshape(j1, 2)      # Fortran (2, j1) column-major = [j][2]
        r["gas_k2m"] = rd.i32(j1)
        r["rad_m2k"] = rd.i32(2 * j5).reshape(j5, 2)
        r["rad_k2m"] = rd.i32(j5)

        def layer(tag):
            r["s1_" + tag], r["s3_" + tag] = rd.f64(j1), rd.f64(j5)
            r["sl1_" + tag] = rd.f64(J2 * NKC)            # sl1(j2, nkc) slab, column-major: [kc][i]
            r["sion1_" + tag] = rd.f64(J6 * NKC)

        def budgets(tag):
            r["bgs_" + tag] = rd.f64(2 * NBGS)            # bgs(2, 122): [slot][2]
            r["level"] = int(rd.i32(1)[0])
            r["bg_" + tag] = rd.f64(2 * nreact) if r["level"] > 0 else np.zeros(2 * nreact)

        layer("in")
        budgets("in")
        r["c_in"], r["rconst"], r["c_out"] = rd.f64(nvar + nfix), rd.f64(nreact), rd.f64(nvar + nfix)
        r["env"] = rd.f64({"gas": 74, "aer": 330, "tot": 544}[r["mech"]])      # Update_RCONST_x's inputs as MISTRA_RATES_ENV_x packed them
        layer("out")
        budgets("out")
        recs.append(r)
    return recs


def main():
    info = open(os.path.join(REF, "BUILD_INFO")).read().replace("\n", "; ")
    per = {"gas": [], "aer": [], "tot": []}
    prov = []
    for name, (fname, what) in CASES.items():
        path = os.path.join(REF, fname)
        if not os.path.exists(path):
            print("no capture", path, "- skipped")
            continue
        prov.append(name + ": " + what)
        for r in read_records(path):
            per[r["mech"]].append(r)
    for mech, rs in per.items():
        if not rs:
            continue
        for key in ("gas_m2k", "gas_k2m", "rad_m2k", "rad_k2m"):      # the maps are fixed for a run
            assert all(np.array_equal(r[key], rs[0][key]) for r in rs)
        out = {k: np.stack([r[k] for r in rs]) for k in rs[0] if isinstance(rs[0][k], np.ndarray) and not k.endswith(("m2k", "k2m"))}
        for key in ("gas_m2k", "gas_k2m", "rad_m2k", "rad_k2m"):
            out[key] = rs[0][key]
        out["k"] = np.array([r["k"] for r in rs], np.int32)
        out["level"] = np.array([r["level"] for r in rs], np.int32)
        out["provenance"] = np.array(" | ".join(prov) + "; " + info)
        path = os.path.join(HERE, "drive_%s.npz" % mech)
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), "bytes;", len(rs), "calls; budget levels among them:", int((out["level"] > 0).sum()))


if __name__ == "__main__":
    main()
