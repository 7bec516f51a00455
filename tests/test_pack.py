"""Host logic of the drivers' hand-over halves, CPU: the tables tools/extract_pack.py cuts out of gas_drive / aer_drive /
tot_drive (with aer_mk.dat / aer_km.dat) and of the budget routines, evaluated by the numpy restatement oracle/pack_py.py, reproduce what
the RUNNING REFERENCE MODEL did in captured driver calls (tests/golden/drive_<mech>.npz, oracle/capture_drive_wrap.f90) — bit for bit:
C as handed to INTEGRATE_x, the clamped sl1 / sion1, s1 / s3 / sl1 / sion1 after the hand-over, bg and bgs."""
import os

import numpy as np
import pytest

from conftest import MECHS, REPO

NVAR = {"gas": 102, "aer": 257, "tot": 417}
NARG = {"gas": 8, "aer": 12, "tot": 16}


def scalars(mech, args):
    """(dt, air, h2o, cvv[4]) from the driver's argument list (gas.f:60-61 | aer.f:59-61 | tot.f:59-61)"""
    if mech == "gas":      # tkpp, dt_ch, yhal, yiod, yhet1, yhet2, air, h2o
        return args[1], args[6], args[7], [0.0, 0.0, 0.0, 0.0]
    if mech == "aer":      # tkpp, dt_ch, xcvv1, xcvv2, yhal, yiod, yliq1, yliq2, yhet1, yhet2, air, h2o
        return args[1], args[10], args[11], [args[2], args[3], 0.0, 0.0]
    return args[1], args[14], args[15], list(args[2:6])


@pytest.mark.parametrize("mech", MECHS)
def test_tables_reproduce_captured_driver_calls(mech):
    from mistra_amd.mechtab import load as load_mech
    from oracle import pack_py
    tab, t = pack_py.load(mech), load_mech(mech)
    g = np.load(os.path.join(REPO, "tests", "golden", "drive_%s.npz" % mech))
    n = g["c_in"].shape[0]
    assert n >= 16 and tab["nvar"] == NVAR[mech]
    nlev = 0
    for i in range(n):
        dt, air, h2o, cvv = scalars(mech, g["args"][i])
        # ---- pack: entries the driver does not set (KPP's dummy products) keep what COMMON held: take them from the capture itself
        C, L, I = pack_py.pack(tab, g["c_in"][i], g["s1_in"][i], g["s3_in"][i], g["sl1_in"][i], g["sion1_in"][i], air, h2o, cvv, g["gas_m2k"], g["rad_m2k"])
        assert np.array_equal(C, g["c_in"][i]), "C handed to INTEGRATE_%s differs (call %d)" % (mech[0], i)
        # (that the pack really writes: a poisoned start must give the same C on every entry the tables name)
        Cp, _, _ = pack_py.pack(tab, np.full_like(C, -7.0), g["s1_in"][i], g["s3_in"][i], g["sl1_in"][i], g["sion1_in"][i], air, h2o, cvv, g["gas_m2k"], g["rad_m2k"])
        written = Cp != -7.0
        assert written.sum() >= len(tab["pack"]) + len(tab["fix"]) and np.array_equal(Cp[written], g["c_in"][i][written])
        # ---- hand-over
        s1, s3, L2, I2 = pack_py.unpack(tab, g["c_out"][i], g["s1_in"][i], g["s3_in"][i], L, I, g["gas_k2m"], g["rad_k2m"])
        for got, key in ((s1, "s1_out"), (s3, "s3_out"), (L2, "sl1_out"), (I2, "sion1_out")):
            assert np.array_equal(got, g[key][i]), "%s differs (call %d)" % (key, i)
        # ---- budgets
        bg, bgs = pack_py.budgets(tab, t, g["c_out"][i], g["rconst"][i], dt, g["bg_in"][i], g["bgs_in"][i])
        assert np.array_equal(bgs.ravel(), g["bgs_out"][i]), "bgs differs (call %d)" % i
        if g["level"][i] > 0:
            nlev += 1
            assert np.array_equal(bg.ravel(), g["bg_out"][i]), "bg differs (call %d)" % i
    assert nlev >= 1


@pytest.mark.parametrize("mech", ["aer", "tot"])
def test_mass_transfer_coefficients_of_captured_layers(mech):
    """SURVEY §8 f3, first slice: xkmt of fast_k_mt_a / fast_k_mt_t (kpp.f90:2683-2947 | 2421-2676) restated in numpy from the extracted
    species list, on layers captured from the running reference model: bit for bit, untouched entries included."""
    from oracle import kmt_py
    tab = kmt_py.load(mech)
    g = np.load(os.path.join(REPO, "tests", "golden", "kmt_%s.npz" % mech))
    assert g["ff"].shape[0] >= 4
    rewritten = 0
    for i in (0, g["ff"].shape[0] // 2, g["ff"].shape[0] - 1):      # (pure-Python loops: a few layers; the last one has a droplet bin active)
        got = kmt_py.fast_k_mt_layer(tab, g["ff"][i], g["rq"], g["kw"], int(g["ka"]), int(g["ifeed"]), int(g["nkc_l"]), g["cw"][i], g["cm"][i], float(g["freep"][i]),
                                     g["alpha"][i], g["vmean"][i], g["xkmt_before"][i])
        assert np.array_equal(got, g["xkmt_after"][i]), "xkmt of layer k=%d differs" % int(g["k"][i])
        rewritten += int((got != g["xkmt_before"][i]).sum())
    assert rewritten > 0
    # the LWC-weighted sedimentation velocity vt(kc,k) the same routine leaves in /kpp_vt/ for SR sedl ("whatever LWC": bins without
    # chemistry too), every captured layer, from a poisoned start: bit for bit (Stokes and Beard regimes; the host libm is the reference's)
    dry = 0
    for i in range(g["ff"].shape[0]):
        got = kmt_py.vt_layer(tab, g["ff"][i], g["rq"], g["kw"], int(g["ka"]), int(g["ifeed"]), int(g["nkc_l"]), g["cw"][i], float(g["t"][i]), float(g["p"][i]),
                              np.full(g["vt_after"].shape[1], -7.0))
        wet = g["cw"][i] > 0.0
        assert np.array_equal(got[wet], g["vt_after"][i][wet]) and np.all(got[~wet] == -7.0), "vt of layer k=%d differs" % int(g["k"][i])
        dry += int((wet & (g["cm"][i] <= 0.0)).sum())
    assert dry >= 1, "no bin without chemistry but with liquid water in the fixture"


def test_clamps_follow_the_compiled_max():
    """MAX(0.d0, x) as flang compiles it (checked with the box's flang -O2, scalar and whole-array forms): -0.0 stays -0.0, NaN stays NaN."""
    from oracle import pack_py
    v = pack_py.fmax0(np.array([-0.0, 0.0, -1.0, 2.5, np.nan, -np.inf]))
    assert np.signbit(v[0]) and v[0] == 0.0 and not np.signbit(v[1]) and v[2] == 0.0 and not np.signbit(v[2]) and v[3] == 2.5 and np.isnan(v[4]) and v[5] == 0.0


@pytest.mark.parametrize("mech", ["aer", "tot"])
def test_henry_and_equilibrium_constants_of_captured_layers(mech):
    """SURVEY §8 f3, second slice: henry_a/t (kpp.f90:1914-2145 | 1676-1907) and equil_co_a/t (kpp.f90:3162-3363 | 2954-3155) restated from
    the tables tools/extract_liq.py cuts out of them, on layers captured from the running reference model (tests/golden/liq_<mech>.npz:
    16 layers of two calls, bins with and without liquid water): bit for bit, the entries the routines leave alone included."""
    from oracle import liq_py
    tab = liq_py.load(mech)
    g = np.load(os.path.join(REPO, "tests", "golden", "liq_%s.npz" % mech))
    assert len(g["henry_k"]) >= 8 and len(g["equil_k"]) >= 8
    for i in range(len(g["henry_k"])):
        got = liq_py.henry_layer(tab, float(g["henry_tt"][i]))
        assert np.array_equal(got, g["henry"][i]), "henry of layer k=%d differs" % int(g["henry_k"][i])
    assert (g["henry"] > 0).sum(axis=1).min() == sum(1 for e in tab["henry"]["entries"] if e[1] > 0.0)      # (four of the 57 listed constants are 0.)
    wet = dry = 0
    for i in range(len(g["equil_k"])):
        ef, eb = liq_py.equil_co_layer(tab, float(g["equil_tt"][i]), g["conv2"][i], g["xgamma"][i], g["xkef_before"][i], g["xkeb_before"][i])
        assert np.array_equal(ef, g["xkef"][i]) and np.array_equal(eb, g["xkeb"][i]), "xkef / xkeb of layer k=%d differ" % int(g["equil_k"][i])
        wet += int((g["conv2"][i][:tab["equil"]["nkc"]] > 0).sum())
        dry += int((g["conv2"][i][:tab["equil"]["nkc"]] <= 0).sum())
    assert wet >= 8 and dry >= 2      # both branches of the routine are in the capture


@pytest.mark.parametrize("mech", ["aer", "tot"])
def test_mean_molecular_speeds_of_captured_layers(mech):
    """v_mean_a / v_mean_t (kpp.f90:1472-1670 | 1268-1465) restated from the table tools/extract_vmean.py cuts out of them, on the layers
    captured from the running reference model: bit for bit, zeros where the routine sets nothing."""
    from oracle import liq_py
    tab = liq_py.load_vmean(mech)
    g = np.load(os.path.join(REPO, "tests", "golden", "liq_%s.npz" % mech))
    assert len(g["vmean_k"]) >= 8
    for i in range(len(g["vmean_k"])):
        assert np.array_equal(liq_py.v_mean_layer(tab, float(g["vmean_tt"][i])), g["vmean"][i])


def test_particle_bin_moments_of_captured_layers():
    """cw_rc (kpp.f90:2152-2414) and dry_cw_rc (kpp.f90:4580-4690) restated (oracle/liq_py.py: cw_rc_layer) on layers captured from the running reference
    model (tests/golden/cwrc.npz): liquid water, mean radius, water mass and the chemistry switch conv2 of the four bins — and rcd, cwd of the dry layers
    above, whose 4./3. is a single-precision quotient — bit for bit; bins switched on and off are both in the capture."""
    from oracle import liq_py
    g = np.load(os.path.join(REPO, "tests", "golden", "cwrc.npz"))
    assert len(g["wet_k"]) >= 8 and len(g["dry_k"]) >= 8
    for i in range(len(g["wet_k"])):
        rc, cw, cm, cv, below = liq_py.cw_rc_layer(g["wet_ff"][i], g["rq"], g["e"], g["kw"], int(g["ka"]), int(g["ifeed"]), float(g["wet_feu"][i]), g["wet_cloud"][i], g["crys4"])
        for got, key in ((rc, "rc"), (cw, "cw"), (cm, "cm"), (cv, "conv2")):
            assert np.array_equal(got, g["wet_" + key][i]), (i, key)
    assert (g["wet_conv2"] > 0).any() and (g["wet_conv2"] == 0).any()
    for i in range(len(g["dry_k"])):
        rc, cw = liq_py.cw_rc_layer(g["dry_ff"][i], g["rq"], g["e"], g["kw"], int(g["ka"]), int(g["ifeed"]), dry=True)
        assert np.array_equal(rc, g["dry_rc"][i][:2]) and np.array_equal(cw, g["dry_cw"][i][:2])


@pytest.mark.parametrize("mech", ["gas", "aer", "tot"])
def test_dry_aerosol_uptake_of_captured_layers(mech):
    """dry_rates_g / dry_rates_a / dry_rates_t (kpp.f90:4697-4853 | 4860-5073 | 5079-5198) restated (oracle/liq_py.py: dry_rates_layer) on layers captured
    from the running reference model (tests/golden/dryrates.npz): xkmtd of the four species in both bins, xeq(HNO3) and — gas — the Henry entries, bit for bit."""
    from oracle import liq_py
    g = np.load(os.path.join(REPO, "tests", "golden", "dryrates.npz"))
    assert len(g[mech + "_k"]) >= 8 and (g[mech + "_rcd"] > 0).any()
    for i in range(len(g[mech + "_k"])):
        a = (float(g[mech + "_tt"][i]), float(g[mech + "_freep"][i]), g[mech + "_rcd"][i])
        if mech == "gas":
            xk, xeq, h = liq_py.dry_rates_layer(*a, None, g["gas_henry4_before"][i])
            assert np.array_equal(h, g["gas_henry4"][i])
        else:
            xk, xeq = liq_py.dry_rates_layer(*a, g[mech + "_vmean4"][i])
        assert np.array_equal(xk, g[mech + "_xkmtd"][i]) and xeq == g[mech + "_xeq"][i]


def test_tables_in_the_repo_are_what_the_extractor_writes(tmp_path):
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("no reference tree here")
    import subprocess
    import sys
    for mech in MECHS:
        before = open(os.path.join(REPO, "mistra_amd", "mech", mech + ".pack"), "rb").read()
        subprocess.run([sys.executable, os.path.join(REPO, "tools", "extract_pack.py"), mech], check=True, stdout=subprocess.DEVNULL)
        assert open(os.path.join(REPO, "mistra_amd", "mech", mech + ".pack"), "rb").read() == before
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "extract_liq.py"), "--check"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "extract_vmean.py"), "--check"], check=True, stdout=subprocess.DEVNULL)
