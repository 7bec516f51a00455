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


def _run(tag, model, minutes, extra, case="base1"):
    env = dict(os.environ, MISTRA_RUN_TAG=tag, MISTRA_MODEL_BIN=os.path.join(REF, model), MISTRA_COLUMN_MINUTES=str(minutes),
               MISTRA_RESET_DUMMIES="1")
    args = [os.path.join(REPO, "oracle", "capture_run.sh"), case, "1"] + ["%s=%s" % kv for kv in extra.items()]
    subprocess.run(args, env=env, check=True, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    from oracle.oracle import read_capture
    return read_capture(os.path.join(REF, "capture_%s%s.bin" % (case, tag)))


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


needs_drive_model = pytest.mark.skipif(not (os.path.isdir("/root/reference/namelists") and os.path.exists(os.path.join(REF, "mistra_capture"))
                                            and os.path.exists(os.path.join(REF, "mistra_drive"))),
                                       reason="needs the reference tree and the two model builds (oracle/build_ref.sh model, oracle/build_drive.sh)")


@needs_drive_model
@pytest.mark.parametrize("case,minutes,mechs", [("base1", 6, ("gas", "aer")), ("BTZ96", 2, ("gas", "aer", "tot"))])
def test_single_pass_batched_driver_reproduces_the_serial_model_bit_for_bit(case, minutes, mechs):
    """shim/kpp_drive.patch: kpp_driver's loop runs ONCE, x_drive stages its layer behind its /kpp_rate_x/ prologue
    (KPP_DRIVE_STAGE_x) and ONE mistra_chem_drive call per mechanism does the rest.  oracle/build_drive.sh links the patched model with the
    unmodified shim/ files and oracle/drive_standin.f90 in the library's place: every staged layer is served by the reference's own x_drive
    called with an argument list REBUILT from what crossed the C boundary — layer number, scal, the rate evaluator's input vector.  The
    records of every INTEGRATE_x call of the last six column steps (inputs as the driver packed them, RCONST, results, /Statistics/) are
    identical to the unpatched model's, bit for bit: deferring the layers changes nothing, and (layer, scal, env) is a complete
    description of a driver call.  base1: gas and aer layers; BTZ96 with chem=T: all three mechanisms."""
    window = dict(MISTRA_CAPTURE_SEQ_FROM=148 * (6 * minutes - 6), MISTRA_CAPTURE_SEQ_TO=148 * 6 * minutes)
    a = _run("_dr_serial", "mistra_capture", minutes, window, case)
    b = _run("_dr_batched", "mistra_drive", minutes, window, case)
    assert len(a) == len(b) == 148 * 6
    for mech in mechs:
        ra, rb = [r for r in a if r["mech"] == mech], [r for r in b if r["mech"] == mech]
        assert len(ra) == len(rb) and len(ra) > 100
        for x, y in zip(ra, rb):
            for k in ("var_in", "fix", "rconst", "var_out", "stats"):
                assert np.array_equal(x[k], y[k]), (mech, k)
            assert x["tin_out"] == y["tin_out"] and x["stepmin_out"] == y["stepmin_out"]


@pytest.mark.parametrize("mode,name", [("two-pass", "kpp_two_pass.patch"), ("drive", "kpp_drive.patch"), ("liq", "kpp_liq.patch")])
def test_patches_in_the_repo_are_what_the_generator_writes(tmp_path, mode, name):
    """shim/kpp_two_pass.patch, shim/kpp_drive.patch and shim/kpp_liq.patch are generated (oracle/two_pass_patch.py), not hand-edited."""
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("no reference tree here")
    out = tmp_path / "p.patch"
    subprocess.run(["python3", os.path.join(REPO, "oracle", "two_pass_patch.py"), "--mode", mode, "/root/reference/src", str(tmp_path / "src"), str(out)],
                   check=True, stdout=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(REPO, "shim", name)).read()


def test_the_gpu_model_build_stages_data_only():
    """oracle/build_gpu_model.sh stages what the end-to-end GPU run (tests/test_gpu_model.py) reads at run time under oracle/_ref/model_inputs, because the
    reference tree does not exist on the GPU box: initial profiles, radiation / photolysis tables, species lists, namelists — DATA.  No source file of the
    reference may be among them, and the scratch copies of the four patched files must be gone."""
    root = os.path.join(REPO, "oracle", "_ref", "model_inputs")
    if not os.path.isdir(root):
        pytest.skip("oracle/build_gpu_model.sh has not run here (no reference tree)")
    seen = []
    for d, _, files in os.walk(root):
        for f in files:
            seen.append(os.path.relpath(os.path.join(d, f), root))
            ext = os.path.splitext(f)[1].lower()
            assert ext in (".dat", ".csv", "") or f.startswith("namelist."), "not a data file: " + seen[-1]
            assert ext not in (".f", ".f90", ".h", ".sc", ".eqn", ".def", ".spc", ".k", ".bud")
            head = open(os.path.join(d, f), "rb").read(4096).lower()
            assert b"subroutine" not in head and b"end module" not in head, "source text in " + seen[-1]
    assert any(s.startswith("namelists/namelist.") for s in seen) and any(s.startswith("mech/") and s.endswith(".csv") for s in seen)
    assert not os.path.exists(os.path.join(REPO, "oracle", "_ref", "gpu_model", "src")), "patched scratch copies of reference files left behind"
