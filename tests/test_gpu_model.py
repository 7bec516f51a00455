"""The reference model END TO END with its chemistry on the MI355X (-m gpu; BASELINE.json configs[4] = SURVEY.md §8 config 5, and a cloudy case).
oracle/_ref/mistra_gpu (oracle/build_gpu_model.sh) is the reference's own compiled routines — dynamics, microphysics, radiation, photolysis, chemistry stem —
with both shipped patches applied (shim/kpp_drive.patch: one device call per mechanism and 10-s step; shim/kpp_liq.patch: liq_parm's kernel calls), the
unmodified Fortran shim and the product library; it reads the model's run-time DATA staged under oracle/_ref/model_inputs (nothing from /root/reference) and
leaves its chemical end state in a file.  Expected: the end state of the UNPATCHED model on the CPU (tests/golden/endstate_<case>.npz,
make_endstate_golden.py).  Tolerances: the gas-only Joyce2014 column 1e-12 (measured 2e-15: every layer sits at the integrator's 7-step floor, where the
kernel is bit-identical to the reference in all but the last place); the stratus column BTZ96 — 34 tot, 46 aer, 68 gas layers, 60 steps that feed back through
cloud water and sedimentation — 1e-4 of an entry above 1e-3 of its species' column maximum (measured 1.3e-6; the integrator's own RTOL is 1e-3).  Also the
box case namelist.Buys13_0D (BASELINE configs[0]: one aer cell per step; 1e-5, measured 1e-7 after 180 steps) and namelist.Bott2020 with chem = T (1e-4,
measured 9e-6)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

REF = os.path.join(REPO, "oracle", "_ref")
have_model = os.path.exists(os.path.join(REF, "mistra_gpu")) and os.path.isdir(os.path.join(REF, "model_inputs", "input"))


def _load_dump(path):
    raw = open(path, "rb").read()
    j1, j5, nsl, nsi, n = (int(x) for x in np.frombuffer(raw, np.int32, 5))
    d, o, out = np.frombuffer(raw, np.float64, offset=20), 0, {}
    for key, width in (("s1", j1), ("s3", j5), ("sl1", nsl), ("sion1", nsi)):
        out[key] = d[o:o + width * n].reshape(n, width); o += width * n
    out["t"] = d[o:o + n]
    return out


@pytest.mark.skipif(not have_model, reason="oracle/_ref/mistra_gpu is built where the reference tree is (oracle/build_gpu_model.sh)")
@pytest.mark.parametrize("case,tol", [("Joyce2014_basecase", 1e-12), ("BTZ96", 1e-4), ("Buys13_0D", 1e-5), ("Bott2020", 1e-4)])
def test_reference_model_with_its_chemistry_on_the_gpu(case, tol, tmp_path):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    g = np.load(os.path.join(REPO, "tests", "golden", "endstate_%s.npz" % case))
    dump = tmp_path / "end.bin"
    r = subprocess.run([os.path.join(REPO, "oracle", "model_run.sh"), os.path.join(REF, "mistra_gpu"), case, str(int(g["minutes"])), str(tmp_path / "run"),
                        "MISTRA_COLUMN_DUMP=%s" % dump], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    log = open(tmp_path / "run" / "stderr.log", errors="replace").read() + open(tmp_path / "run" / "stdout.log", errors="replace").read()
    assert "not registered" not in log, "the model's COMMON arrays were refused by mistra_chem_pin_host"
    got = _load_dump(dump)
    worst = 0.0
    for key in ("s1", "s3", "sl1", "sion1"):
        want, have = g[key], got[key]
        assert have.shape == want.shape
        scale = np.abs(want).max(axis=0, keepdims=True)
        major = np.abs(want) > 1e-3 * scale
        rel = np.abs(have - want)[major] / np.abs(want[major])
        assert rel.max() <= tol, "%s: %s differs from the reference model's end state by %.2e" % (case, key, rel.max())
        assert np.all(np.abs(have - want)[~major] <= tol * np.broadcast_to(scale, want.shape)[~major] + 1e-300)
        worst = max(worst, float(rel.max()))
    assert np.abs(got["t"] - g["t"]).max() <= 1e-9
    line = [x for x in log.splitlines() if "chemistry stem" in x][-1].strip()
    print("\n  %s, %d model minutes, chemistry on the GPU: end state within %.1e of the reference model's; %s\n  reference model on the build container's CPU: %s"
          % (case, int(g["minutes"]), worst, line, str(g["provenance"]).split(";")[1].strip()))
