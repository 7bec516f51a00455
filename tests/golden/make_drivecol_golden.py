#!/usr/bin/env python3
"""tests/golden/drivecol_<case>.npz: ONE whole 10-s step of the column as kpp_driver's layer loop ran it in the reference model — every
gas_drive / aer_drive / tot_drive call of the step in model order (oracle/capture_drive_wrap.f90, window MISTRA_CAPTURE_DRIVE_SEQ_FROM/TO),
joined with the INTEGRATE_x records of the same calls (oracle/capture_wrap.c: /Statistics/, exit time).  Per layer: mechanism, k, the
driver's scalars (air, h2o, cvv1..4, dt), the rate evaluator's input vector as MISTRA_RATES_ENV_x packed it inside the model, the layer's
rows of s1 / s3 / sl1 / sion1 / bgs before and after, C as handed to INTEGRATE_x and after it; bg of the budget levels.  The species maps of
the mechanisms that ran.  Both models ran with MISTRA_RESET_DUMMIES=1 (KPP's dummy products start from 0 in every layer, as a batched driver
gives them).  Data only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from make_drive_golden import read_records      # noqa: E402
from oracle.oracle import read_capture          # noqa: E402

REF = os.path.join(HERE, "..", "..", "oracle", "_ref")
CASES = {
    "BTZ96": ("capture_drivecol_BTZ96.bin", "capture_BTZ96_dcol.bin",
              "reference namelist.BTZ96 (chem=F -> T, netcdf=F), the 12th 10-s step (model minute 2; stratus: gas, aer and tot layers)"),
    "Joyce2014": ("capture_drivecol_Joyce2014.bin", "capture_Joyce2014_basecase_dcol.bin",
                  "reference namelist.Joyce2014_basecase as shipped (netcdf=F), the 12th 10-s step (148 gas layers)"),
    "base1": ("capture_drivecol_base1.bin", "capture_base1_dcol.bin", "reference namelist.base1 (netcdf=F), the 36th 10-s step (gas and aer layers)"),
}
HOW = ("MISTRA_RESET_DUMMIES=1 MISTRA_RUN_TAG=_dcol MISTRA_COLUMN_MINUTES=<m> oracle/capture_run.sh <case> 1 MISTRA_CAPTURE_DRIVE_FILE=... "
       "MISTRA_CAPTURE_DRIVE_SEQ_FROM=148*(s-1) _SEQ_TO=148*s MISTRA_CAPTURE_SEQ_FROM=148*(s-1) MISTRA_CAPTURE_SEQ_TO=148*s")
MID = {"gas": 0, "aer": 1, "tot": 2}
NENV = 544


def scalars(mech, a):
    if mech == "gas":
        return a[1], [a[6], a[7], 0, 0, 0, 0]
    if mech == "aer":
        return a[1], [a[10], a[11], a[2], a[3], 0, 0]
    return a[1], [a[14], a[15], a[2], a[3], a[4], a[5]]


def main():
    info = open(os.path.join(REF, "BUILD_INFO")).read().replace("\n", "; ")
    for case, (fdrive, fint, what) in CASES.items():
        pd, pi = os.path.join(REF, fdrive), os.path.join(REF, fint)
        if not (os.path.exists(pd) and os.path.exists(pi)):
            print("no capture for", case, "- skipped")
            continue
        rs, ints = read_records(pd), read_capture(pi)
        assert len(rs) == len(ints) == 148, (len(rs), len(ints))
        out = {}
        n = len(rs)
        out["mech"] = np.array([MID[r["mech"]] for r in rs], np.int32)
        out["k"] = np.array([r["k"] for r in rs], np.int32)
        out["level"] = np.array([r["level"] for r in rs], np.int32)
        sc = np.zeros((n, 6))
        env = np.zeros((n, NENV))
        nv = max(len(r["c_in"]) for r in rs)
        c_in, c_out = np.zeros((n, nv)), np.zeros((n, nv))
        for i, (r, q) in enumerate(zip(rs, ints)):
            assert q["mech"] == r["mech"] and np.array_equal(q["var_in"], r["c_in"][:len(q["var_in"])])      # the same call in both records
            dt, sc[i] = scalars(r["mech"], r["args"])
            assert dt == 10.0
            env[i, :len(r["env"])] = r["env"]
            c_in[i, :len(r["c_in"])] = r["c_in"]
            c_out[i, :len(r["c_out"])] = r["c_out"]
        out.update(scal=sc, env=env, c_in=c_in, c_out=c_out)
        for key in ("s1_in", "s3_in", "sl1_in", "sion1_in", "bgs_in", "s1_out", "s3_out", "sl1_out", "sion1_out", "bgs_out"):
            out[key] = np.stack([r[key] for r in rs])
        lev = [i for i, r in enumerate(rs) if r["level"] > 0]
        nb = max(len(rs[i]["bg_in"]) for i in lev) if lev else 0
        bgi, bgo = np.zeros((len(lev), nb)), np.zeros((len(lev), nb))
        for j, i in enumerate(lev):
            bgi[j, :len(rs[i]["bg_in"])] = rs[i]["bg_in"]
            bgo[j, :len(rs[i]["bg_out"])] = rs[i]["bg_out"]
        out.update(bg_layers=np.array(lev, np.int32), bg_in=bgi, bg_out=bgo)
        out["stats"] = np.stack([q["stats"] for q in ints])
        out["tin_out"] = np.array([q["tin_out"] for q in ints])
        for mech in ("gas", "aer", "tot"):
            mine = [r for r in rs if r["mech"] == mech]
            if mine:
                for key in ("gas_m2k", "gas_k2m", "rad_m2k", "rad_k2m"):
                    assert all(np.array_equal(r[key], mine[0][key]) for r in mine)
                    out["%s_%s" % (mech, key)] = mine[0][key]
        out["provenance"] = np.array(what + "; " + HOW + "; " + info)
        path = os.path.join(HERE, "drivecol_%s.npz" % case)
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), "bytes; layers per mechanism", np.bincount(out["mech"], minlength=3).tolist(), "budget levels", len(lev))


if __name__ == "__main__":
    main()
