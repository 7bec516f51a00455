#!/usr/bin/env python3
"""Generate tests/golden/integrate_<mech>.npz from a capture of the running reference model.

Provenance (all steps happen in the build container, where /root/reference exists):
  1. oracle/build_ref.sh all          compiles the reference's own Fortran sources with flang -O2 -ffp-contract=off
  2. oracle/capture_run.sh BTZ96 <hours> ...  runs the reference's shipped stratus case (namelists/namelist.BTZ96 with chem=T,
                                       netcdf=F, 1 model hour) through oracle/column_driver.f90 and records real
                                       INTEGRATE_g/a/t calls (oracle/capture_wrap.c): /GDATA_x/ before and after, /Statistics/
  3. this script                        converts the records into the small fixtures committed here

The fixtures are DATA: inputs (VAR, FIX, RCONST, TIN, TOUT) and the reference's outputs (VAR, statistics, exit time,
last step).  The exact capture command is stored in each file's `provenance` field.
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.oracle import read_capture  # noqa: E402

REF = os.path.join(HERE, "..", "..", "oracle", "_ref")
# name suffix -> (capture file, what was run)
SETS = {
    "": (os.path.join(REF, "capture_BTZ96.bin"),
         "reference namelist.BTZ96 (chem=T, netcdf=F, lstmax=1: first model hour, night); "
         "oracle/capture_run.sh BTZ96 1 MISTRA_CAPTURE_SKIP_t=1500 MISTRA_CAPTURE_EVERY_t=241 MISTRA_CAPTURE_MAX_t=64 "
         "MISTRA_CAPTURE_SKIP_a=1000 MISTRA_CAPTURE_EVERY_a=331 MISTRA_CAPTURE_MAX_a=32 "
         "MISTRA_CAPTURE_SKIP_g=2000 MISTRA_CAPTURE_EVERY_g=701 MISTRA_CAPTURE_MAX_g=32"),
    "_day": (os.path.join(REF, "capture_BTZ96_day.bin"),
             "reference namelist.BTZ96 (chem=T, netcdf=F, lstmax=8: calls from model hours 6.5-8, after sunrise, photolysis on); "
             "oracle/capture_run.sh BTZ96 8 MISTRA_CAPTURE_SKIP_t=110000 MISTRA_CAPTURE_EVERY_t=1201 MISTRA_CAPTURE_MAX_t=24 "
             "MISTRA_CAPTURE_SKIP_a=70000 MISTRA_CAPTURE_EVERY_a=1301 MISTRA_CAPTURE_MAX_a=16 "
             "MISTRA_CAPTURE_SKIP_g=150000 MISTRA_CAPTURE_EVERY_g=2203 MISTRA_CAPTURE_MAX_g=16"),
    "_base1": (os.path.join(REF, "capture_base1.bin"),
               "reference namelist.base1 (chem=T as shipped, netcdf=F, lstmax=2: cloud-free marine boundary layer, gas and aerosol "
               "chemistry); oracle/capture_run.sh base1 2 MISTRA_CAPTURE_SKIP_t=200 MISTRA_CAPTURE_EVERY_t=997 MISTRA_CAPTURE_MAX_t=12 "
               "MISTRA_CAPTURE_SKIP_a=500 MISTRA_CAPTURE_EVERY_a=2003 MISTRA_CAPTURE_MAX_a=16 "
               "MISTRA_CAPTURE_SKIP_g=500 MISTRA_CAPTURE_EVERY_g=3001 MISTRA_CAPTURE_MAX_g=12"),
    # BASELINE.json configs[0], the reference's own CPU-runnable case: a box-model run (one level, aerosol mechanism, 360
    # INTEGRATE_a calls per model hour), sequenced by oracle/column_driver.f90's box branch
    "_buys13": (os.path.join(REF, "capture_Buys13_0D.bin"),
                "reference namelist.Buys13_0D (box=T as shipped, netcdf=F, lstmax=1: first model hour of the box run, 32 steps per call); "
                "oracle/capture_run.sh Buys13_0D 1 MISTRA_CAPTURE_SKIP_a=20 MISTRA_CAPTURE_EVERY_a=11 MISTRA_CAPTURE_MAX_a=32"),
}


# whole column steps: EVERY INTEGRATE_x call of kpp_driver's layer loop (kpp.f90:4310-4470) for one or two 10-s steps, in
# the model's call order, whatever mechanism each layer ran (BASELINE.json configs[4]: 148 cells per step)
COLUMN_SETS = {
    "Joyce2014": (os.path.join(REF, "capture_Joyce2014_basecase_nuc.bin"),
                  "reference namelist.Joyce2014_basecase as shipped (nuc=T, Napari and Lovejoy nucleation; only netcdf=F and lstmax=1), "
                  "column steps 180 and 181 of the first model hour (all 148 layers run the gas mechanism); "
                  "MISTRA_RUN_TAG=_nuc MISTRA_COLUMN_MINUTES=31 oracle/capture_run.sh Joyce2014_basecase 1 "
                  "MISTRA_CAPTURE_SEQ_FROM=26640 MISTRA_CAPTURE_SEQ_TO=26936"),
    "base1": (os.path.join(REF, "capture_base1_col.bin"),
              "reference namelist.base1 (netcdf=F, lstmax=1), column step 300 of the first model hour (79 layers run gas, 69 aer); "
              "MISTRA_RUN_TAG=_col oracle/capture_run.sh base1 1 MISTRA_CAPTURE_SEQ_FROM=44400 MISTRA_CAPTURE_SEQ_TO=44548"),
    "BTZ96": (os.path.join(REF, "capture_BTZ96_col.bin"),
              "reference namelist.BTZ96 (chem=F -> T, netcdf=F, lstmax=1), column step 150 of the first model hour (stratus: gas, aer "
              "and tot layers); MISTRA_RUN_TAG=_col oracle/capture_run.sh BTZ96 1 MISTRA_CAPTURE_SEQ_FROM=22200 MISTRA_CAPTURE_SEQ_TO=22348"),
}


def main():
    only = set(sys.argv[1:])          # e.g. `make_golden.py _buys13 BTZ96`: just these sets (default: all whose capture exists)
    for suffix, (capture, cmd) in SETS.items():
        if only and suffix not in only:
            continue
        if os.path.exists(capture):
            convert(suffix, capture, cmd)
        else:
            print("no capture", capture, "- skipped")
    for name, (capture, cmd) in COLUMN_SETS.items():
        if only and name not in only:
            continue
        if os.path.exists(capture) and os.path.getsize(capture):
            convert_column(name, capture, cmd)
        else:
            print("no capture", capture, "- skipped")


def convert_column(name, capture, cmd):
    """tests/golden/column_<name>.npz: per mechanism m the arrays m_var_in, m_fix, m_rconst, m_var_out, m_stats, m_tin_out,
    m_stepmin_out and m_seq (position of the call in the model's sequence of INTEGRATE_x calls; 148 consecutive positions
    = one column step)."""
    recs = read_capture(capture)
    info = open(os.path.join(HERE, "..", "..", "oracle", "_ref", "BUILD_INFO")).read()
    out = dict(provenance=np.array(cmd + "; " + info.replace("\n", "; ")), cells_per_step=np.int32(148))
    for mech in ("gas", "aer", "tot"):
        rs = [r for r in recs if r["mech"] == mech]
        if not rs:
            continue
        out[mech + "_var_in"] = np.stack([r["var_in"] for r in rs])
        out[mech + "_fix"] = np.stack([r["fix"] for r in rs])
        out[mech + "_rconst"] = np.stack([r["rconst"] for r in rs])
        out[mech + "_var_out"] = np.stack([r["var_out"] for r in rs])
        out[mech + "_stats"] = np.stack([r["stats"] for r in rs]).astype(np.int32)
        out[mech + "_tin_out"] = np.array([r["tin_out"] for r in rs])
        out[mech + "_stepmin_out"] = np.array([r["stepmin_out"] for r in rs])
        out[mech + "_seq"] = np.array([r["callno"] for r in rs], np.int32)
        assert all(r["tin"] == 0.0 and r["tout"] == rs[0]["tout"] for r in rs)
        out["tout"] = np.float64(rs[0]["tout"])
    path = os.path.join(HERE, "column_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("column", name, {m: len([r for r in recs if r["mech"] == m]) for m in ("gas", "aer", "tot")}, "->", path, os.path.getsize(path), "bytes")


def convert(suffix, capture, cmd):
    recs = read_capture(capture)
    info = open(os.path.join(HERE, "..", "..", "oracle", "_ref", "BUILD_INFO")).read()
    for mech in ("gas", "aer", "tot"):
        rs = [r for r in recs if r["mech"] == mech]
        if not rs:
            print(mech, suffix, "no records")
            continue
        out = dict(
            var_in=np.stack([r["var_in"] for r in rs]), fix=np.stack([r["fix"] for r in rs]),
            rconst=np.stack([r["rconst"] for r in rs]), var_out=np.stack([r["var_out"] for r in rs]),
            stats=np.stack([r["stats"] for r in rs]).astype(np.int32),
            tin=np.array([r["tin"] for r in rs]), tout=np.array([r["tout"] for r in rs]),
            tin_out=np.array([r["tin_out"] for r in rs]), stepmin_out=np.array([r["stepmin_out"] for r in rs]),
            callno=np.array([r["callno"] for r in rs], np.int32),
            provenance=np.array(cmd + "; " + info.replace("\n", "; ")))
        path = os.path.join(HERE, "integrate_%s%s.npz" % (mech, suffix))
        np.savez_compressed(path, **out)
        if suffix == "":      # the benchmark workload's base states (mistra_amd/workload.py): the inputs only, inside the package
            np.savez_compressed(os.path.join(HERE, "..", "..", "mistra_amd", "data", "base_%s.npz" % mech), var=out["var_in"],
                                fix=out["fix"], rconst=out["rconst"], provenance=out["provenance"])
        print(mech, len(rs), "records ->", path, os.path.getsize(path), "bytes; steps", out["stats"][:, 2].min(), "..", out["stats"][:, 2].max())


if __name__ == "__main__":
    main()
