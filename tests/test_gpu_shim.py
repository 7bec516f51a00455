"""Drop-in proof for the Fortran surface (-m gpu): a Fortran program fills COMMON /GDATA_x/ and calls
INTEGRATE_x(TIN, TOUT) — the reference's own signature (gas.f:710 | aer.f:1408 | tot.f:2812), here provided by
shim/mistra_kpp_shim.f90 over the C ABI — and gets the captured reference results back."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import MECHS, REPO, rel_diff

pytestmark = pytest.mark.gpu
FLANG = "/opt/rocm/lib/llvm/bin/flang"


@pytest.mark.skipif(not os.path.exists(FLANG), reason="no Fortran compiler on this box")
@pytest.mark.parametrize("mech", MECHS)
def test_fortran_integrate_x_through_shim(mech, golden, tmp_path):
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "shim")], check=True)
    g = golden[mech]
    n = 4
    rec = np.concatenate([g["var_in"][:n], g["fix"][:n], g["rconst"][:n]], axis=1)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(np.array([float(n)]).tobytes())
        f.write(np.ascontiguousarray(rec).tobytes())
    subprocess.run([os.path.join(REPO, "shim", "shim_driver"), mech[0], str(fin), str(fout)], check=True, timeout=300)
    nvar = g["var_in"].shape[1]
    out = np.fromfile(fout, np.float64).reshape(n, nvar + 2)
    assert rel_diff(out[:, :nvar], g["var_out"][:n]).max() <= 2e-5
    assert np.allclose(out[:, nvar], g["tin_out"][:n], rtol=1e-12)              # TIN <- exit time
    # STEPMIN <- last step size: it follows the error estimate of the most sensitive trace species, so it carries the
    # same round-off-level spread as the concentrations (tests/test_gpu_parity.py), not more
    assert np.allclose(out[:, nvar + 1], g["stepmin_out"][:n], rtol=2e-5)
