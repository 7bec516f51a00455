"""Drop-in proof for the Fortran surface (-m gpu): a Fortran program fills COMMON /GDATA_x/ and calls
INTEGRATE_x(TIN, TOUT) — the reference's own signature (gas.f:710 | aer.f:1408 | tot.f:2812), here provided by
shim/mistra_kpp_shim.f90 over the C ABI — and gets the captured reference results back; and the batched form a two-pass
kpp_driver would call (INTEGRATE_BATCH_x) replays whole captured column steps of the reference model
(BASELINE.json configs[4]: all 148 layers of one 10-s step) as ONE call per mechanism."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import MECHS, REPO, rel_diff

pytestmark = pytest.mark.gpu
FLANG = "/opt/rocm/lib/llvm/bin/flang"
DRIVER = os.path.join(REPO, "shim", "shim_driver")
needs_flang = pytest.mark.skipif(not os.path.exists(FLANG), reason="no Fortran compiler on this box")


def _write_cells(path, var, fix, rconst):
    rec = np.concatenate([var, fix, rconst], axis=1)
    with open(path, "wb") as f:
        f.write(np.array([float(var.shape[0])]).tobytes())
        f.write(np.ascontiguousarray(rec).tobytes())


@needs_flang
@pytest.mark.parametrize("mech", MECHS)
def test_fortran_integrate_x_through_shim(mech, golden, tmp_path):
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "shim")], check=True)
    g = golden[mech]
    n = 4
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    _write_cells(fin, g["var_in"][:n], g["fix"][:n], g["rconst"][:n])
    subprocess.run([DRIVER, mech[0], str(fin), str(fout)], check=True, timeout=300)
    nvar = g["var_in"].shape[1]
    out = np.fromfile(fout, np.float64).reshape(n, nvar + 4)
    assert rel_diff(out[:, :nvar], g["var_out"][:n]).max() <= 2e-5
    assert np.allclose(out[:, nvar], g["tin_out"][:n], rtol=1e-12)              # TIN <- exit time
    # STEPMIN <- last step size: it follows the error estimate of the most sensitive trace species, so it carries the
    # same round-off-level spread as the concentrations (tests/test_gpu_parity.py), not more
    assert np.allclose(out[:, nvar + 1], g["stepmin_out"][:n], rtol=2e-5)
    assert np.all(out[:, nvar + 2] == 1.0e-25) and np.all(out[:, nvar + 3] == 1.0e-3)    # ATOL, RTOL as INTEGRATE_x leaves them (gas.f:745-746)


@needs_flang
def test_fortran_prints_the_reference_messages_on_failure(golden, tmp_path):
    """ros_ErrorMsg_x (gas.f:1474-1509) and INTEGRATE_x's PRINT (gas.f:764-767) come out of the Fortran shim on unit 6, and the
    model carries on with the next cell.  A NaN concentration makes the step-size test fail at once (IERR = -7)."""
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "shim")], check=True)
    g = golden["gas"]
    var = g["var_in"][:2].copy()
    var[0, 5] = np.nan
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    _write_cells(fin, var, g["fix"][:2], g["rconst"][:2])
    r = subprocess.run([DRIVER, "g", str(fin), str(fout)], check=True, timeout=300, capture_output=True, text=True)
    text = r.stdout
    assert "Forced exit from Rosenbrock_g due to the following error:" in text
    assert "--> Step size too small: T + 10*H = T or H < Roundoff" in text
    assert "T=" in text and "and H=" in text
    assert "Rosenbrock: Unsucessful step at T=" in text and re.search(r"\(IERR=\s*-7\s*\)", text)
    out = np.fromfile(fout, np.float64).reshape(2, 102 + 4)
    assert rel_diff(out[1:, :102], g["var_out"][1:2]).max() <= 2e-5      # the second cell is integrated as if nothing had happened


@needs_flang
def test_fortran_prints_the_zero_pivot_row(golden, tmp_path):
    """ros_PrepareMatrix_x's "Warning: LU Decomposition returned ising = <row>" (gas.f:1456) from the Fortran shim, with the row
    KppDecomp_x would return (gas.f:6157): a first-order loss with rate constant -1/(Hstart*gamma) puts an exact zero on that
    species' diagonal at the first attempt; H is halved and the integration goes on."""
    from mistra_amd.mechtab import load
    from test_gpu_phases import GAMMA, _first_order_losses
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "shim")], check=True)
    t, g = load("gas"), golden["gas"]
    r, s = _first_order_losses(t)[0]
    K = np.zeros((1, t.nreact))
    K[0, r] = -1.0 / (1.0e-3 * GAMMA[0])
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    _write_cells(fin, g["var_in"][:1], g["fix"][:1], K)
    out = subprocess.run([DRIVER, "g", str(fin), str(fout)], check=True, timeout=300, capture_output=True, text=True).stdout
    m = re.search(r"Warning: LU Decomposition returned ising =\s+(\d+)", out)
    assert m, out
    assert int(m.group(1)) == s + 1


@needs_flang
@pytest.mark.parametrize("mech", MECHS)
def test_fortran_rates_then_integrator_from_env(mech, tmp_path):
    """SURVEY §8 f1 from the Fortran side: the vectors MISTRA_RATES_ENV_x packed inside the running reference model
    (tests/golden/rates_model_<mech>.npz) go through UPDATE_RCONST_BATCH_x — RCONST as the reference's Update_RCONST_x made them of
    the same COMMON blocks, to 1e-13 relative (the device's exp / log / pow against the host libm, tests/test_gpu_rates.py; zeros
    exactly where the reference has zeros) — and through INTEGRATE_BATCH_ENV_x, rates and integrator on the device in one call: the
    same bits as the integrator fed those device-made RCONST, and the oracle's results on the REFERENCE's RCONST within the
    integrator's parity bound."""
    from mistra_amd import chem
    from oracle.oracle import Oracle
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "shim")], check=True)
    g = np.load(os.path.join(REPO, "tests", "golden", "rates_model_%s.npz" % mech))
    var, fix, env, rconst = g["var"], g["fix"], g["env"], g["rconst"]
    n, nvar = var.shape
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    _write_cells(fin, var, fix, env)
    subprocess.run([DRIVER, "E" + mech[0], str(fin), str(fout)], check=True, timeout=300)
    raw = np.fromfile(fout, np.float64)
    out = raw[:n * nvar].reshape(n, nvar)
    tail = raw[n * nvar:n * nvar + 9 * n].reshape(n, 9)
    k = raw[n * nvar + 9 * n:].reshape(n, rconst.shape[1])
    assert np.array_equal(k == 0.0, rconst == 0.0)
    nz = rconst != 0.0
    assert (np.abs(k[nz] - rconst[nz]) / np.abs(rconst[nz])).max() <= 1e-13, "UPDATE_RCONST_BATCH_%s against the reference's Update_RCONST_%s" % (mech[0], mech[0])
    assert np.all(tail[:, 0] == 1)
    same = chem.integrate(mech, var, fix, k)
    assert np.array_equal(out, same.var) and np.array_equal(tail[:, 1:].astype(np.int32), same.stats)
    want = Oracle(mech).integrate_batch(var, fix, rconst)
    assert np.array_equal(tail[:, 1:].astype(np.int32), want[2])
    assert rel_diff(out, want[0]).max() <= 2e-5


def _column(name):
    return dict(np.load(os.path.join(REPO, "tests", "golden", "column_%s.npz" % name)))


@needs_flang
@pytest.mark.parametrize("name", ["Joyce2014", "base1", "BTZ96"])
def test_fortran_batched_column_step(name, tmp_path, capsys):
    """Every captured 10-s step of the whole column (two of Joyce2014, one of the others), every layer's INTEGRATE_x call of the
    reference model, replayed from Fortran as ONE INTEGRATE_BATCH_x call per mechanism and step (what kpp_driver's layer loop, kpp.f90:4310-4470, becomes with the
    two-pass patch of INTEGRATION.md).  Results, exit times, last steps and /Statistics/ against the capture."""
    path = os.path.join(REPO, "tests", "golden", "column_%s.npz" % name)
    if not os.path.exists(path):
        pytest.skip("no column fixture for %s" % name)
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "shim")], check=True)
    col = _column(name)
    per_step = int(col["cells_per_step"])
    first = min(int(col[m + "_seq"].min()) for m in MECHS if m + "_seq" in col)
    nsteps = sum(len(col[m + "_seq"]) for m in MECHS if m + "_seq" in col) // per_step
    assert nsteps >= (2 if name == "Joyce2014" else 1)
    for step in range(nsteps):
        total_ms, ncells = 0.0, 0
        for mech in MECHS:
            if mech + "_seq" not in col:
                continue
            sel = (col[mech + "_seq"] - first) // per_step == step          # this column step's calls
            var, fix, rconst = col[mech + "_var_in"][sel], col[mech + "_fix"][sel], col[mech + "_rconst"][sel]
            n, nvar = var.shape
            fin, fout = tmp_path / ("in_%s.bin" % mech), tmp_path / ("out_%s.bin" % mech)
            _write_cells(fin, var, fix, rconst)
            subprocess.run([DRIVER, mech[0].upper(), str(fin), str(fout)], check=True, timeout=300)
            raw = np.fromfile(fout, np.float64)
            out = raw[:n * (nvar + 4)].reshape(n, nvar + 4)
            tail = raw[n * (nvar + 4):n * (nvar + 4) + 9 * n].reshape(n, 9)
            ms = float(raw[-1])
            assert np.all(tail[:, 0] == 1), "IERR"
            assert np.array_equal(tail[:, 1:].astype(np.int32), col[mech + "_stats"][sel]), "/Statistics/ differ from the reference's"
            assert rel_diff(out[:, :nvar], col[mech + "_var_out"][sel]).max() <= 2e-5
            assert np.allclose(out[:, nvar], col[mech + "_tin_out"][sel], rtol=1e-12)
            # STEPMIN <- the step size proposed after the last accepted step = H * 0.9 / Err^(1/3): Err is built from Yerr = E1*K1 +
            # E2*K2 + E3*K3, a small difference of large terms, so it carries a far larger relative round-off spread than the
            # concentrations do (7e-5 seen on one aer cell of this column with every /Statistics/ entry and exit time identical);
            # nothing in the model reads STEPMIN back (INTEGRATE_x restarts every call at Hstart = 1e-3, gas.f:743)
            assert np.allclose(out[:, nvar + 1], col[mech + "_stepmin_out"][sel], rtol=1e-3)
            total_ms += ms
            ncells += n
            with capsys.disabled():
                print("\n  column %s, %s: %d layers in one call from Fortran, %.2f ms" % (name, mech, n, ms))
        assert ncells == per_step
        with capsys.disabled():
            # the reference spends 69 us per gas call inside kpp_driver on one 2.1 GHz core (pack + rates + integrator)
            print("  column %s: %d cells of one 10-s step in %.2f ms through the batched Fortran surface" % (name, ncells, total_ms))


@needs_flang
@pytest.mark.parametrize("mech", ["aer", "tot"])
def test_fortran_liq_parm_kernels(mech, tmp_path):
    """SURVEY §8 f3 from Fortran: FAST_K_MT_BATCH (xkmt and the sedimentation velocity vt), HENRY_BATCH, V_MEAN_BATCH, ST_COEFF_BATCH, EQUIL_CO_BATCH of
    shim/mistra_kpp_liq.f90 — the host-buffer calls the drop-ins of shim/mistra_kpp_model.f90 make with the model's arrays in place — on the
    layers captured from the running reference model.  The results must be the device-pointer entry points' bit for bit (same kernels), i.e.
    what tests/test_gpu_kmt.py and test_gpu_liq.py pin against the captures."""
    import torch
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "shim")], check=True)
    from mistra_amd import chem
    chem.init(0)
    dev = torch.device("cuda", 0)
    T = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    f8 = lambda *a: np.array(a, np.float64)
    # ---- fast_k_mt_x
    g = np.load(os.path.join(REPO, "tests", "golden", "kmt_%s.npz" % mech))
    nl, nkc, nspec = g["xkmt_before"].shape
    nka, nkt = g["rq"].shape
    vt0 = np.full((nl, nkc), -7.0)
    with open(fin, "wb") as f:
        f8(nl, nka, nkt, nkc, nspec, int(g["ka"]), int(g["ifeed"]), int(g["nkc_l"])).tofile(f)
        for a in (g["kw"].astype(np.float64), g["rq"], g["ff"], g["cw"], g["cm"], g["freep"], g["alpha"], g["vmean"], g["xkmt_before"], g["t"], g["p"], vt0):
            np.ascontiguousarray(a, np.float64).tofile(f)
    subprocess.run([DRIVER, "K" + mech[0], str(fin), str(fout)], check=True, timeout=300)
    raw = np.fromfile(fout, np.float64)
    xk, vt = raw[:nl * nkc * nspec].reshape(nl, nkc, nspec), raw[nl * nkc * nspec:].reshape(nl, nkc)
    assert np.array_equal(xk, g["xkmt_after"]), "xkmt from Fortran differs from the reference's"
    wet = g["cw"] > 0
    assert np.all(vt[~wet] == -7.0) and (np.abs(vt[wet] - g["vt_after"][wet]) / np.abs(g["vt_after"][wet])).max() <= 1e-14
    dx, dv = T(g["xkmt_before"]), T(vt0)
    chem.fast_k_mt(mech, T(g["ff"]), T(g["rq"]), g["kw"], int(g["ka"]), int(g["ifeed"]), int(g["nkc_l"]), T(g["cw"]), T(g["cm"]), T(g["freep"]), T(g["alpha"]),
                   T(g["vmean"]), dx, T(g["t"]), T(g["p"]), dv)
    torch.cuda.synchronize()
    assert np.array_equal(dx.cpu().numpy(), xk) and np.array_equal(dv.cpu().numpy(), vt)
    # ---- henry_x, equil_co_x
    g = np.load(os.path.join(REPO, "tests", "golden", "liq_%s.npz" % mech))
    nl, nspec = g["henry"].shape
    with open(fin, "wb") as f:
        f8(nl, nspec).tofile(f)
        np.ascontiguousarray(g["henry_tt"], np.float64).tofile(f)
    subprocess.run([DRIVER, "H" + mech[0], str(fin), str(fout)], check=True, timeout=300)
    hen = np.fromfile(fout, np.float64).reshape(nl, nspec)
    dh = torch.full((nl, nspec), float("nan"), dtype=torch.float64, device=dev)
    chem.henry(mech, T(g["henry_tt"]), dh)
    torch.cuda.synchronize()
    assert np.array_equal(hen, dh.cpu().numpy())
    nz = g["henry"] != 0
    assert np.array_equal(hen == 0, ~nz) and (np.abs(hen[nz] - g["henry"][nz]) / np.abs(g["henry"][nz])).max() <= 1e-14
    # ---- v_mean_x: bit for bit against the running model
    nl, nspec = g["vmean"].shape
    with open(fin, "wb") as f:
        f8(nl, nspec).tofile(f)
        np.ascontiguousarray(g["vmean_tt"], np.float64).tofile(f)
    subprocess.run([DRIVER, "V" + mech[0], str(fin), str(fout)], check=True, timeout=300)
    assert np.array_equal(np.fromfile(fout, np.float64).reshape(nl, nspec), g["vmean"])
    # ---- st_coeff_x: what the device-pointer entry returns, bit for bit; against the running model to the last place of exp
    gs = np.load(os.path.join(REPO, "tests", "golden", "stcoeff_%s.npz" % mech))
    for jo in sorted(set(gs["lp_joyce14bc"].tolist())):
        pick = gs["lp_joyce14bc"] == jo
        env, want = gs["env"][pick], gs["alpha"][pick]
        with open(fin, "wb") as f:
            f8(env.shape[0], want.shape[1], jo, 0).tofile(f)
            np.ascontiguousarray(env, np.float64).tofile(f)
        subprocess.run([DRIVER, "S" + mech[0], str(fin), str(fout)], check=True, timeout=300)
        al = np.fromfile(fout, np.float64).reshape(want.shape)
        da = torch.full(want.shape, float("nan"), dtype=torch.float64, device=dev)
        chem.st_coeff(mech, T(env), da, bool(jo), False)
        torch.cuda.synchronize()
        nz = want != 0
        assert np.array_equal(al, da.cpu().numpy()) and np.array_equal(al == 0, ~nz) and (np.abs(al[nz] - want[nz]) / np.abs(want[nz])).max() <= 1e-14
    nl, nkc, j6 = g["xgamma"].shape
    with open(fin, "wb") as f:
        f8(nl, nkc, j6, nspec).tofile(f)
        for a in (g["equil_tt"], g["conv2"], g["xgamma"], g["xkef_before"], g["xkeb_before"]):
            np.ascontiguousarray(a, np.float64).tofile(f)
    subprocess.run([DRIVER, "Q" + mech[0], str(fin), str(fout)], check=True, timeout=300)
    raw = np.fromfile(fout, np.float64)
    ef, eb = raw[:nl * nkc * nspec].reshape(nl, nkc, nspec), raw[nl * nkc * nspec:].reshape(nl, nkc, nspec)
    df, db = T(g["xkef_before"]), T(g["xkeb_before"])
    chem.equil_co(mech, T(g["equil_tt"]), T(g["conv2"]), T(g["xgamma"]), df, db)
    torch.cuda.synchronize()
    assert np.array_equal(ef, df.cpu().numpy()) and np.array_equal(eb, db.cpu().numpy())
    for got, want in ((ef, g["xkef"]), (eb, g["xkeb"])):
        nz = want != 0
        assert np.array_equal(got == 0, ~nz) and (np.abs(got[nz] - want[nz]) / np.abs(want[nz])).max() <= 1e-14


@needs_flang
def test_fortran_particle_bin_moments(tmp_path):
    """cw_rc and dry_cw_rc from Fortran (CW_RC_BATCH of shim/mistra_kpp_liq.f90, what the drop-ins CW_RC_HIP / DRY_CW_RC_HIP of shim/mistra_kpp_model.f90
    call with the model's arrays in place) on the layers captured from the running reference model: bit for bit."""
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "shim")], check=True)
    g = np.load(os.path.join(REPO, "tests", "golden", "cwrc.npz"))
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    for dry, pre in ((0, "wet"), (1, "dry")):
        ff = g[pre + "_ff"]
        nl, nka, nkt = ff.shape
        with open(fin, "wb") as f:
            np.array([nl, nkt, nka, dry, int(g["ka"]), int(g["ifeed"])], np.float64).tofile(f)
            for a in (g["kw"].astype(np.float64), g["rq"], g["e"], g["crys4"], ff, g[pre + "_feu"], g[pre + "_cloud"].astype(np.float64)):
                np.ascontiguousarray(a, np.float64).tofile(f)
        subprocess.run([DRIVER, "Ca", str(fin), str(fout)], check=True, timeout=300)
        both = np.fromfile(fout, np.float64)
        nb = 2 if dry else 4
        assert both.size == 2 * (4 * nl * nb + nl)      # the driver makes the call twice: plain, then with the spectrum registered (PIN_HOST)
        for raw in (both[:both.size // 2], both[both.size // 2:]):
            rc, cw, cm, cv = (raw[i * nl * nb:(i + 1) * nl * nb].reshape(nl, nb) for i in range(4))
            assert np.array_equal(rc, g[pre + "_rc"][:, :nb]) and np.array_equal(cw, g[pre + "_cw"][:, :nb])
            if not dry:
                assert np.array_equal(cm, g["wet_cm"]) and np.array_equal(cv, g["wet_conv2"])
                assert np.array_equal(raw[4 * nl * nb:], (g["wet_feu"] < g["crys4"][:2].min()).astype(np.float64))


@needs_flang
@pytest.mark.parametrize("mech", ["gas", "aer", "tot"])
def test_fortran_dry_aerosol_uptake(mech, tmp_path):
    """dry_rates_x from Fortran (DRY_RATES_BATCH of shim/mistra_kpp_liq.f90, what the drop-ins DRY_RATES_HIP_g/_a/_t call after gathering the four species of
    the routines' idr list) on the captured layers: what the Python entry returns, bit for bit."""
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "shim")], check=True)
    from mistra_amd import chem
    chem.init(0)
    g = np.load(os.path.join(REPO, "tests", "golden", "dryrates.npz"))
    nl = len(g[mech + "_k"])
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    four = g["gas_henry4_before"] if mech == "gas" else g[mech + "_vmean4"]
    with open(fin, "wb") as f:
        np.array([nl], np.float64).tofile(f)
        for a in (g[mech + "_tt"], g[mech + "_freep"], g[mech + "_rcd"], four):
            np.ascontiguousarray(a, np.float64).tofile(f)
    subprocess.run([DRIVER, "R" + mech[0], str(fin), str(fout)], check=True, timeout=300)
    raw = np.fromfile(fout, np.float64)
    xk, xeq, h = raw[:8 * nl].reshape(nl, 2, 4), raw[8 * nl:9 * nl], raw[9 * nl:].reshape(nl, 4)
    if mech == "gas":
        wk, wq, wh = chem.dry_rates(g["gas_tt"], g["gas_freep"], g["gas_rcd"], None, g["gas_henry4_before"])
        assert np.array_equal(h, wh)
    else:
        wk, wq = chem.dry_rates(g[mech + "_tt"], g[mech + "_freep"], g[mech + "_rcd"], g[mech + "_vmean4"])
    assert np.array_equal(xk, wk) and np.array_equal(xeq, wq)
