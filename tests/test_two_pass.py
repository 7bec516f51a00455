"""The two-pass kpp_driver patch of INTEGRATION.md §4 (shim/kpp_two_pass.patch), validated where it can run: in the build
container, on the CPU.  oracle/build_two_pass.sh links the reference model with the patched kpp.f90 / x_drive, the UNMODIFIED
Fortran shim and batch module of shim/, and — there being no GPU here — oracle/two_pass_standin.c in the place of
libmistra_chem.so (the batched calls are served by the reference's own integrator).  The patched model and the unpatched one
run the same case; every INTEGRATE_x call is recorded in both (oracle/capture_wrap.c) and the records must be IDENTICAL, bit
for bit: same inputs per layer, same results, same /Statistics/ — i.e. the pack / record / batch / hand-back / budget
plumbing of the two passes changes nothing.

One thing in the reference cannot survive batching and is switched off in BOTH models for the comparison
(MISTRA_RESET_DUMMIES): x_drive never initialises KPP's dummy product species (DUMM1, DUMM2), so serially they carry the
previous LAYER's leftovers in COMMON /GDATA_x/ into the next layer's error norm.  See INTEGRATION.md §4."""
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO

REF = os.path.join(REPO, "oracle", "_ref")
needs_models = pytest.mark.skipif(not (os.path.isdir("/root/reference/namelists") and os.path.exists(os.path.join(REF, "mistra_capture"))
                                       and os.path.exists(os.path.join(REF, "mistra_two_pass"))),
                                  reason="needs the reference tree and the two model builds (oracle/build_ref.sh model, oracle/build_two_pass.sh)")


def _run(tag, model, minutes, extra):
    env = dict(os.environ, MISTRA_RUN_TAG=tag, MISTRA_MODEL_BIN=os.path.join(REF, model), MISTRA_COLUMN_MINUTES=str(minutes),
               MISTRA_RESET_DUMMIES="1")
    args = [os.path.join(REPO, "oracle", "capture_run.sh"), "base1", "1"] + ["%s=%s" % kv for kv in extra.items()]
    subprocess.run(args, env=env, check=True, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    from oracle.oracle import read_capture
    return read_capture(os.path.join(REF, "capture_base1%s.bin" % tag))


@needs_models
def test_two_pass_driver_reproduces_the_serial_model_bit_for_bit():
    minutes = 6          # 36 column steps of 148 layers (gas and aer), ~5 300 INTEGRATE_x calls per model
    window = dict(MISTRA_CAPTURE_SEQ_FROM=148 * 30, MISTRA_CAPTURE_SEQ_TO=148 * 36)      # the last six steps, every layer
    a = _run("_tp_serial", "mistra_capture", minutes, window)
    b = _run("_tp_batched", "mistra_two_pass", minutes, window)
    assert len(a) == len(b) == 148 * 6
    # the batched model integrates a step's gas layers, then its aer layers; per mechanism the layer order is the model's
    for mech in ("gas", "aer"):
        ra, rb = [r for r in a if r["mech"] == mech], [r for r in b if r["mech"] == mech]
        assert len(ra) == len(rb) and len(ra) > 100
        for x, y in zip(ra, rb):
            for k in ("var_in", "fix", "rconst", "var_out", "stats"):
                assert np.array_equal(x[k], y[k]), (mech, k)
            assert x["tin_out"] == y["tin_out"] and x["stepmin_out"] == y["stepmin_out"]


def test_patch_in_the_repo_is_what_the_generator_writes(tmp_path):
    """shim/kpp_two_pass.patch is generated (oracle/two_pass_patch.py), not hand-edited."""
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("no reference tree here")
    out = tmp_path / "p.patch"
    subprocess.run(["python3", os.path.join(REPO, "oracle", "two_pass_patch.py"), "/root/reference/src", str(tmp_path / "src"), str(out)],
                   check=True, stdout=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(REPO, "shim", "kpp_two_pass.patch")).read()
