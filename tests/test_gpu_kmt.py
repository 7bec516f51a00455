"""liq_parm, first slice, on the device (-m gpu): the mass-transfer coefficients xkmt of fast_k_mt_a / fast_k_mt_t
(kpp.f90:2683-2947 | 2421-2676) from mistra_chem_fast_k_mt_device, against layers captured from the RUNNING reference model
(tests/golden/kmt_<mech>.npz: what the routine read for the layer, xkmt(:,:,k) before and after its call).  Same summation order, one
rounding per operation: bit for bit, entries the routine leaves alone included.  The same call returns the LWC-weighted sedimentation
velocity vt(kc,k) the routine leaves in /kpp_vt/ for SR sedl (str.f90:2704): bit for bit in the Stokes regime, to the last place of the device
log / exp where Beard's polynomial is involved (radii above 10 um)."""
import os

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mech", ["aer", "tot"])
def test_mass_transfer_coefficients_on_the_device(mech):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mistra_amd import chem
    chem.init(0)
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(REPO, "tests", "golden", "kmt_%s.npz" % mech))
    T = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    xkmt = T(g["xkmt_before"])
    vt = T(np.full(g["vt_after"].shape, -7.0))      # poisoned: bins with cw <= 0 must keep it
    chem.fast_k_mt(mech, T(g["ff"]), T(g["rq"]), g["kw"], int(g["ka"]), int(g["ifeed"]), int(g["nkc_l"]), T(g["cw"]), T(g["cm"]), T(g["freep"]), T(g["alpha"]),
                   T(g["vmean"]), xkmt, T(g["t"]), T(g["p"]), vt)
    torch.cuda.synchronize()
    got = xkmt.cpu().numpy()
    gv, wet = vt.cpu().numpy(), g["cw"] > 0.0
    assert np.all(gv[~wet] == -7.0), "vt written for a bin without liquid water"
    rel = np.abs(gv[wet] - g["vt_after"][wet]) / np.abs(g["vt_after"][wet])
    assert rel.max() <= 1e-14, "vt differs from the reference's (max rel %.2e)" % rel.max()
    print("%s: vt of %d bins within %.1e (%d of them bit-identical; %d bins without chemistry)" % (mech, int(wet.sum()), rel.max(), int((gv[wet] == g["vt_after"][wet]).sum()),
          int((wet & (g["cm"] <= 0)).sum())))
    # without vt the call is the xkmt half alone (and asks for no t, p)
    x2 = T(g["xkmt_before"])
    chem.fast_k_mt(mech, T(g["ff"]), T(g["rq"]), g["kw"], int(g["ka"]), int(g["ifeed"]), int(g["nkc_l"]), T(g["cw"]), T(g["cm"]), T(g["freep"]), T(g["alpha"]), T(g["vmean"]), x2)
    torch.cuda.synchronize()
    assert np.array_equal(x2.cpu().numpy(), got)
    with pytest.raises(chem.MistraChemError):      # kw of the wrong length is refused on the host
        chem.fast_k_mt(mech, T(g["ff"]), T(g["rq"]), g["kw"][:-1], int(g["ka"]), int(g["ifeed"]), int(g["nkc_l"]), T(g["cw"]), T(g["cm"]), T(g["freep"]), T(g["alpha"]), T(g["vmean"]), x2)
    changed = int((g["xkmt_after"] != g["xkmt_before"]).sum())
    assert changed >= 100
    assert np.array_equal(got, g["xkmt_after"]), "xkmt differs from the reference's (max rel %.2e)" % np.nanmax(np.abs(got - g["xkmt_after"]) / (np.abs(g["xkmt_after"]) + 1e-300))
    print("%s: xkmt of %d captured layers bit-identical (%d coefficients rewritten, bins active: %s)" % (mech, got.shape[0], changed, (g["cm"] > 0).sum(axis=0).tolist()))
    # gas has no such routine: the call fails loudly
    with pytest.raises(chem.MistraChemError):
        chem.fast_k_mt("gas", T(g["ff"]), T(g["rq"]), g["kw"], int(g["ka"]), 0, 2, T(g["cw"]), T(g["cm"]), T(g["freep"]), T(g["alpha"]), T(g["vmean"]), xkmt)
