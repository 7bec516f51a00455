#!/usr/bin/env python3
"""tests/golden/endstate_<case>.npz: the chemical state the UNPATCHED reference model (oracle/_ref/mistra_capture, build_ref.sh `model`: the reference's own
routines, flang -O2 -ffp-contract=off, sequenced by oracle/column_driver.f90) ends in after a number of model minutes — what the same model with its
chemistry on the GPU (oracle/_ref/mistra_gpu, build_gpu_model.sh) has to arrive at (tests/test_gpu_model.py).  Data only: s1 [n, j1], s3 [n, j5] of module
gas_common, sl1 [n, nkc*j2], sion1 [n, nkc*j6] of /blck17/, the temperature profile t [n].

    oracle/build_gpu_model.sh            (stages the model's run-time data under oracle/_ref/model_inputs)
    python tests/golden/make_endstate_golden.py"""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.join(HERE, "..", "..")
REF = os.path.join(REPO, "oracle", "_ref")
CASES = (("Joyce2014_basecase", 5), ("BTZ96", 10), ("Buys13_0D", 10), ("Bott2020", 5))


def load_dump(path):
    raw = open(path, "rb").read()
    j1, j5, nsl, nsi, n = (int(x) for x in np.frombuffer(raw, np.int32, 5))
    d, o, out = np.frombuffer(raw, np.float64, offset=20), 0, {}
    for key, width in (("s1", j1), ("s3", j5), ("sl1", nsl), ("sion1", nsi)):
        out[key] = d[o:o + width * n].reshape(n, width).copy(); o += width * n
    out["t"] = d[o:o + n].copy(); o += n
    assert o == d.size
    return out


def main():
    info = open(os.path.join(REF, "BUILD_INFO")).read().replace("\n", "; ")
    for case, minutes in CASES:
        run, dump = os.path.join(REF, "run_model_%s_cpu" % case), os.path.join(REF, "endstate_%s_cpu.bin" % case)
        r = subprocess.run([os.path.join(REPO, "oracle", "model_run.sh"), os.path.join(REF, "mistra_capture"), case, str(minutes), run, "MISTRA_COLUMN_DUMP=" + dump],
                           check=True, capture_output=True, text=True)
        line = [x for x in open(os.path.join(run, "stderr.log"), errors="replace").read().splitlines() if "chemistry stem" in x][-1].strip()
        out = load_dump(dump)
        out["minutes"] = np.array(minutes)
        out["provenance"] = np.array("reference namelist.%s (netcdf=F, chem=T), %d model minutes of the unpatched model; %s; %s" % (case, minutes, line, info))
        path = os.path.join(HERE, "endstate_%s.npz" % case)
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), "bytes;", line)


if __name__ == "__main__":
    main()
