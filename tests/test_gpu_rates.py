"""Update_RCONST_g/a/t on the device (-m gpu): the rate constants the HIP evaluator (mistra_amd/csrc/rates.hip)
makes of seeded input vectors against what the COMPILED REFERENCE made of them (tests/golden/rates_<mech>.npz: update_rconst_x_
and the rate laws of kpp.f90 through oracle/_ref/libmistra_ref.so, recorded by tests/golden/make_rates_golden.py).

Tolerance: the table and the evaluation order are the reference's (tests/test_rates.py reproduces it bit for bit with the
host libm); the device's exp / pow / log10 differ from the host's in the last place, and a rate law chains up to five of
them, so: exact where no transcendental is involved (switches, literals, photolysis rates), 1e-13 relative elsewhere."""
import os

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def chem():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mistra_amd import chem as c
    c.init(0)
    return c


@pytest.mark.parametrize("mech", ["gas", "aer", "tot"])
def test_device_rate_constants_match_the_reference(chem, mech):
    g = np.load(os.path.join(REPO, "tests", "golden", "rates_%s.npz" % mech))
    env, want = g["env"], g["rconst"]
    got = chem.update_rconst(mech, env)
    assert got.shape == want.shape
    assert np.array_equal(got == 0.0, want == 0.0)
    nz = want != 0.0
    rel = np.abs(got[nz] - want[nz]) / np.abs(want[nz])
    exact = float((got[nz] == want[nz]).mean())
    print("%s RCONST on the device vs compiled reference: %d values, %.1f %% bit-identical, max rel diff %.2e" % (mech, int(nz.sum()), 100 * exact, rel.max()))
    assert rel.max() <= 1e-13
    # reactions without a rate-law call are products of inputs and literals: no library function, no tolerance
    import json
    table = json.load(open(os.path.join(REPO, "mistra_amd", "mech", mech + ".rates.json")))
    plain = np.array([not any(t[0] == "call" for t in p) for p in table["programs"]])
    assert plain.sum() > 100 and np.array_equal(got[:, plain], want[:, plain])


@pytest.mark.parametrize("mech", ["gas", "aer", "tot"])
def test_model_captured_calls_on_the_device(chem, mech):
    """The env vectors the product's Fortran routine MISTRA_RATES_ENV_x packed INSIDE the running reference model
    (tests/golden/rates_model_<mech>.npz) through the device evaluator, against the RCONST the reference's Update_RCONST_x left in
    COMMON /GDATA_x/ for the same layer."""
    g = np.load(os.path.join(REPO, "tests", "golden", "rates_model_%s.npz" % mech))
    got, want = chem.update_rconst(mech, g["env"]), g["rconst"]
    assert np.array_equal(got == 0.0, want == 0.0)
    nz = want != 0.0
    rel = np.abs(got[nz] - want[nz]) / np.abs(want[nz])
    print("%s RCONST of %d model layers on the device: %.1f %% bit-identical, max rel diff %.2e" % (mech, len(want), 100 * float((got[nz] == want[nz]).mean()), rel.max()))
    assert rel.max() <= 1e-13


@pytest.mark.parametrize("mech", ["gas", "aer", "tot"])
def test_rates_feed_the_integrator_on_the_device(chem, mech, oracles):
    """Update_RCONST_x -> INTEGRATE_x without the rate constants leaving the GPU, on layers captured from the running model (env,
    VAR, FIX as they stood at the reference's Update_RCONST_x call): same results as the oracle integrating with the REFERENCE's
    RCONST of that call, identical step bookkeeping, every layer succeeding."""
    import torch
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(REPO, "tests", "golden", "rates_model_%s.npz" % mech))
    n = min(16, g["env"].shape[0])
    env = torch.tensor(g["env"][:n], device=dev)
    var, fix = g["var"][:n], g["fix"][:n]
    rconst = chem.update_rconst(mech, env)
    assert rconst.is_cuda
    res = chem.integrate(mech, torch.tensor(var, device=dev), torch.tensor(fix, device=dev), rconst, 0.0, 10.0)
    torch.cuda.synchronize()
    want, ierr, st = oracles[mech].integrate_batch(var, fix, g["rconst"][:n], 0.0, 10.0)
    assert np.all(ierr == 1) and np.array_equal(res.ierr.cpu().numpy(), ierr)
    assert np.array_equal(res.stats.cpu().numpy(), st), "/Statistics/ differ"
    floor = 1e-12 * np.abs(want).max(axis=1, keepdims=True)
    rel = np.abs(res.var.cpu().numpy() - want) / (np.abs(want) + floor)
    print("%s: rates -> integrator on the device, %d model layers, max rel diff %.2e, steps %s" % (mech, n, rel.max(), st[:, 2].tolist()))
    assert rel.max() <= 2e-5


@pytest.mark.parametrize("mech", ["gas", "aer", "tot"])
def test_host_buffer_call_from_the_rate_inputs(chem, mech, oracles):
    """mistra_chem_integrate_env_ex (what INTEGRATE_BATCH_ENV_x of the Fortran shim calls): VAR, FIX and the rate evaluator's inputs
    go up, RCONST is made on the device.  Same answers as the oracle with the reference's RCONST of those layers; also through the
    in-library split over device slots."""
    g = np.load(os.path.join(REPO, "tests", "golden", "rates_model_%s.npz" % mech))
    n = min(16, g["env"].shape[0])
    var, fix, env = g["var"][:n], g["fix"][:n], g["env"][:n]
    want, ierr, st = oracles[mech].integrate_batch(var, fix, g["rconst"][:n], 0.0, 10.0)
    res, th = chem.integrate_ex(mech, var, fix, None, 0.0, 10.0, env=env)
    assert np.array_equal(res.ierr, ierr) and np.array_equal(res.stats, st)
    floor = 1e-12 * np.abs(want).max(axis=1, keepdims=True)
    assert (np.abs(res.var - want) / (np.abs(want) + floor)).max() <= 2e-5
    assert np.allclose(th[:, 0], 10.0, rtol=1e-12)
    chem.finalize()
    try:
        chem.init_devices([0, 0])
        res2, th2 = chem.integrate_ex(mech, var, fix, None, 0.0, 10.0, env=env)
        assert np.array_equal(res2.var, res.var) and np.array_equal(res2.stats, res.stats) and np.array_equal(th2, th)
    finally:
        chem.finalize()
        chem.init(0)
