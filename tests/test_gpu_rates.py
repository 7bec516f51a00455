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


def test_rates_feed_the_integrator_on_the_device(chem, golden, oracles):
    """Update_RCONST_g -> INTEGRATE_g without the rate constants leaving the GPU: same results as integrating with the
    reference's RCONST of the same inputs (oracle), same step bookkeeping."""
    import torch
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(REPO, "tests", "golden", "rates_gas.npz"))
    n = 16
    env = torch.tensor(g["env"][:n], device=dev)
    var, fix = golden["gas"]["var_in"][:n], g["env"][:n, 58:61]          # FIX as Update_RCONST_g saw it
    rconst = chem.update_rconst("gas", env)
    assert rconst.is_cuda
    res = chem.integrate("gas", torch.tensor(var, device=dev), torch.tensor(fix, device=dev), rconst, 0.0, 10.0)
    torch.cuda.synchronize()
    want, ierr, st = oracles["gas"].integrate_batch(var, fix, g["rconst"][:n], 0.0, 10.0)
    assert np.array_equal(res.ierr.cpu().numpy(), ierr)
    ok = ierr == 1
    assert ok.sum() >= n // 2
    assert np.array_equal(res.stats.cpu().numpy()[ok], st[ok])
    floor = 1e-12 * np.abs(want).max(axis=1, keepdims=True)
    assert (np.abs(res.var.cpu().numpy() - want) / (np.abs(want) + floor))[ok].max() <= 2e-5
