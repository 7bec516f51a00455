"""liq_parm, second slice, on the device (-m gpu): the Henry constants of henry_a / henry_t (kpp.f90:1914-2145 | 1676-1907),
the mean molecular speeds of v_mean_a / v_mean_t (kpp.f90:1472-1670 | 1268-1465, mistra_chem_v_mean_device: bit for bit)
and the equilibrium rate constants of equil_co_a / equil_co_t (kpp.f90:3162-3363 | 2954-3155) from mistra_chem_henry_device /
mistra_chem_equil_co_device, against layers captured from the RUNNING reference model (tests/golden/liq_<mech>.npz).  Entries that are
numbers or products of numbers, conv2 and activity coefficients come out bit for bit; where exp is involved the device library's exp
differs from the host libm's in the last place: 1e-14 relative, stated here."""
import os

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu
TOL = 1e-14


def _rel(a, b):
    nz = b != 0.0
    return float((np.abs(a[nz] - b[nz]) / np.abs(b[nz])).max()) if nz.any() else 0.0


@pytest.mark.parametrize("mech", ["aer", "tot"])
def test_henry_and_equilibrium_constants_on_the_device(mech):
    import json
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mistra_amd import chem
    chem.init(0)
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(REPO, "tests", "golden", "liq_%s.npz" % mech))
    tab = json.load(open(os.path.join(REPO, "mistra_amd", "mech", mech + ".liq.json")))
    T = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    # ---- henry_x: the whole array is written (a poisoned buffer comes back clean)
    want = g["henry"]
    out = torch.full(want.shape, float("nan"), dtype=torch.float64, device=dev)
    chem.henry(mech, T(g["henry_tt"]), out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got == 0.0, want == 0.0) and np.isfinite(got).all()
    plain = np.array([j - 1 for j, _, b0 in tab["henry"]["entries"] if b0 is None])
    assert np.array_equal(got[:, plain], want[:, plain])                                   # number / (number * T): no library function
    assert _rel(got, want) <= TOL
    # ---- equil_co_x: in/out arrays, bins with and without liquid water
    ef, eb = T(g["xkef_before"]), T(g["xkeb_before"])
    chem.equil_co(mech, T(g["equil_tt"]), T(g["conv2"]), T(g["xgamma"]), ef, eb)
    torch.cuda.synchronize()
    gf, gb = ef.cpu().numpy(), eb.cpu().numpy()
    for got, want in ((gf, g["xkef"]), (gb, g["xkeb"])):
        assert np.array_equal(got == 0.0, want == 0.0)
        assert _rel(got, want) <= TOL
    nkc_eq = tab["equil"]["nkc"]
    dry = g["conv2"][:, :nkc_eq] <= 0
    assert dry.any() and (gf[:, :nkc_eq][dry] == 0).all() and (gb[:, :nkc_eq][dry] == 0).all()
    # what the routine does not set is untouched: species without an entry in wet bins, and (aer) the bins it never visits
    listed = np.zeros(gf.shape[2], bool)
    listed[[e[0] - 1 for e in tab["equil"]["entries"]]] = True
    wet = ~dry
    for got, before in ((gf, g["xkef_before"]), (gb, g["xkeb_before"])):
        assert np.array_equal(got[:, :nkc_eq][wet][:, ~listed], before[:, :nkc_eq][wet][:, ~listed])
        assert np.array_equal(got[:, nkc_eq:], before[:, nkc_eq:])
    # entries without exp: bit for bit
    noexp_f = np.array([e[0] - 1 for e in tab["equil"]["entries"] if not any(f[0] == "funa" for f in e[1])])
    noexp_b = np.array([e[0] - 1 for e in tab["equil"]["entries"] if not any(f[0] == "funa" for f in e[2])])
    assert np.array_equal(gf[:, :, noexp_f], g["xkef"][:, :, noexp_f]) and np.array_equal(gb[:, :, noexp_b], g["xkeb"][:, :, noexp_b])
    print("%s: henry of %d layers (max rel %.1e), xkef / xkeb of %d layers (%.1e / %.1e) against the running model" %
          (mech, got.shape[0], _rel(out.cpu().numpy(), g["henry"]), gf.shape[0], _rel(gf, g["xkef"]), _rel(gb, g["xkeb"])))
    # gas has no such routines: the calls fail loudly
    with pytest.raises(chem.MistraChemError):
        chem.henry("gas", T(g["henry_tt"]), out)


@pytest.mark.parametrize("mech", ["aer", "tot"])
def test_mean_molecular_speeds_on_the_device(mech):
    """v_mean_a / v_mean_t (kpp.f90:1472-1670 | 1268-1465) from mistra_chem_v_mean_device against the layers captured from the running reference
    model: vmean(:,k) bit for bit (quotient, IEEE square root, product: no library function with last-place freedom), the species the routine
    does not list exactly 0, the whole array written."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mistra_amd import chem
    chem.init(0)
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(REPO, "tests", "golden", "liq_%s.npz" % mech))
    want = g["vmean"]
    out = torch.full(want.shape, float("nan"), dtype=torch.float64, device=dev)
    chem.v_mean(mech, torch.tensor(np.ascontiguousarray(g["vmean_tt"]), device=dev), out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got, want), "max rel %.2e" % _rel(got, want)
    assert (want != 0).sum(axis=1).min() >= 90 and (want == 0).any()
    with pytest.raises(chem.MistraChemError):
        chem.v_mean("gas", torch.tensor(np.ascontiguousarray(g["vmean_tt"]), device=dev), out)


@pytest.mark.parametrize("mech", ["aer", "tot"])
def test_accommodation_coefficients_on_the_device(mech):
    """st_coeff_a / st_coeff_t (kpp.f90:857-1038 | 664-851) from mistra_chem_st_coeff_device — the postfix programs of <mech>.stcoeff run by the
    evaluator of the rate constants — against the layers captured from the running reference model, both settings of lpJoyce14bc for aer: the
    species whose coefficient is a literal (or the default 0.1) bit for bit, the temperature laws and a_n2o5 to the last place of the device
    exp (asserted: 1e-14), the whole array written."""
    import json
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mistra_amd import chem
    chem.init(0)
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(REPO, "tests", "golden", "stcoeff_%s.npz" % mech))
    tab = json.load(open(os.path.join(REPO, "mistra_amd", "mech", mech + ".stcoeff.json")))
    for jo in sorted(set(g["lp_joyce14bc"].tolist())):
        for bu in sorted(set(g["lp_buxmann15alph"].tolist())):
            pick = (g["lp_joyce14bc"] == jo) & (g["lp_buxmann15alph"] == bu)
            if not pick.any():
                continue
            want = g["alpha"][pick]
            out = torch.full(want.shape, float("nan"), dtype=torch.float64, device=dev)
            chem.st_coeff(mech, torch.tensor(np.ascontiguousarray(g["env"][pick]), device=dev), out, bool(jo), bool(bu))
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            assert np.isfinite(got).all() and np.array_equal(got == 0.0, want == 0.0) and _rel(got, want) <= TOL, _rel(got, want)
            progs = tab["variants"][jo + 2 * bu]["programs"]
            plain = np.array([j for j, p in enumerate(progs) if not any(t[0] == "call" and t[1] != "min" for t in p)])
            assert len(plain) > 200 and np.array_equal(got[:, plain], want[:, plain])
    with pytest.raises(chem.MistraChemError):
        chem.st_coeff("gas", torch.zeros((1, 5), dtype=torch.float64, device=dev), torch.zeros((1, 105), dtype=torch.float64, device=dev))


def test_particle_bin_moments_on_the_device():
    """cw_rc (kpp.f90:2152-2414) and dry_cw_rc (kpp.f90:4580-4690) from mistra_chem_cw_rc against the layers captured from the running reference model
    (tests/golden/cwrc.npz): every output bit for bit (serial sums in the reference's order, no library function), the whole arrays written."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mistra_amd import chem
    chem.init(0)
    g = np.load(os.path.join(REPO, "tests", "golden", "cwrc.npz"))
    rc, cw, cm, cv, below = chem.cw_rc(g["wet_ff"], g["rq"], g["e"], g["kw"], int(g["ka"]), int(g["ifeed"]), g["wet_feu"], g["wet_cloud"], g["crys4"])
    for got, key in ((rc, "rc"), (cw, "cw"), (cm, "cm"), (cv, "conv2")):
        assert np.array_equal(got, g["wet_" + key]), key
    assert np.array_equal(below, (g["wet_feu"] < g["crys4"][:2].min()).astype(np.int32))
    rcd, cwd = chem.cw_rc(g["dry_ff"], g["rq"], g["e"], g["kw"], int(g["ka"]), int(g["ifeed"]), dry=True)
    assert np.array_equal(rcd, g["dry_rc"][:, :2]) and np.array_equal(cwd, g["dry_cw"][:, :2])


@pytest.mark.parametrize("mech", ["gas", "aer", "tot"])
def test_dry_aerosol_uptake_on_the_device(mech):
    """dry_rates_g / _a / _t (kpp.f90:4697-4853 | 4860-5073 | 5079-5198) from mistra_chem_dry_rates against layers captured from the running reference model
    (tests/golden/dryrates.npz): xkmtd of aer and tot bit for bit (no library function: the speeds come in), xeq and the gas routine's Henry constants and
    speeds to the last place of the device exp / sqrt (asserted: 1e-14)."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mistra_amd import chem
    chem.init(0)
    g = np.load(os.path.join(REPO, "tests", "golden", "dryrates.npz"))
    a = (g[mech + "_tt"], g[mech + "_freep"], g[mech + "_rcd"])
    if mech == "gas":
        xk, xeq, h = chem.dry_rates(*a, None, g["gas_henry4_before"])
        assert _rel(h, g["gas_henry4"]) <= TOL and np.array_equal(h == 0, g["gas_henry4"] == 0)
        assert _rel(xk, g["gas_xkmtd"]) <= TOL
    else:
        xk, xeq = chem.dry_rates(*a, g[mech + "_vmean4"])
        assert np.array_equal(xk, g[mech + "_xkmtd"])
    assert _rel(xeq, g[mech + "_xeq"]) <= TOL


def test_registered_caller_memory_gives_the_same_results():
    """mistra_chem_pin_host (what the Fortran drop-ins do once for the model's COMMON arrays): blocks inside a registered range travel without the staging
    copy — same bits out; overlapping and unknown ranges are refused."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mistra_amd import chem
    chem.init(0)
    g = np.load(os.path.join(REPO, "tests", "golden", "cwrc.npz"))
    base = np.zeros((97,) + g["wet_ff"].shape[1:])
    ff = base[:96]
    ff[:] = np.concatenate([g["wet_ff"]] * 8)
    feu, cloud = np.concatenate([g["wet_feu"]] * 8), np.concatenate([g["wet_cloud"]] * 8)
    a = (g["rq"], g["e"], g["kw"], int(g["ka"]), int(g["ifeed"]), feu, cloud, g["crys4"])
    plain = chem.cw_rc(ff, *a)
    chem.pin_host(ff)
    chem.pin_host(ff[1:])                               # inside a registered range: nothing to do
    with pytest.raises(chem.MistraChemError):
        chem.pin_host(base)                             # overlaps the registered range without lying inside it
    try:
        pinned = chem.cw_rc(ff, *a)
        half = chem.cw_rc(ff[:40], *(a[:5] + (feu[:40], cloud[:40], g["crys4"])))      # a sub-run of the registered layers
    finally:
        chem.unpin_host(ff)
    for x, y in zip(plain, pinned):
        assert np.array_equal(x, y)
    for x, y in zip(plain, half):
        assert np.array_equal(x[:40], y)
    assert np.array_equal(plain[0][:12], g["wet_rc"])
    with pytest.raises(chem.MistraChemError):
        chem.unpin_host(ff)                             # not registered any more
