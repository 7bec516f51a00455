"""The hand-over halves of the per-layer drivers on the device (-m gpu): pack, budgets and hand-over kernels of
mistra_amd/csrc/pack.hip through the C ABI, against whole gas_drive / aer_drive / tot_drive calls captured from the RUNNING reference
model (tests/golden/drive_<mech>.npz: oracle/capture_drive_wrap.f90 around the real drivers of namelist.BTZ96, chem=T).

Index work and plain products in the reference's operation order: every comparison is bit for bit.  The device-resident chain
(pack -> Update_RCONST_x -> INTEGRATE_x -> budgets -> hand-over) is compared with the reference's own end state of the same driver
call to the integrator's stated tolerance, its step bookkeeping being checked through the oracle (the capture holds no /Statistics/)."""
import os

import numpy as np
import pytest

from conftest import MECHS, REPO, rel_diff

pytestmark = pytest.mark.gpu
NVAR = {"gas": 102, "aer": 257, "tot": 417}


@pytest.fixture(scope="module")
def chem():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mistra_amd import chem as c
    c.init(0)
    return c


def _load(mech):
    return dict(np.load(os.path.join(REPO, "tests", "golden", "drive_%s.npz" % mech)))


def _scal(mech, args):
    """[ncell][6] = air, h2o, cvv1..4 and dt from the drivers' argument lists (gas.f:60-61 | aer.f:59-61 | tot.f:59-61)"""
    n = args.shape[0]
    sc = np.zeros((n, 6))
    if mech == "gas":
        sc[:, 0], sc[:, 1] = args[:, 6], args[:, 7]
    elif mech == "aer":
        sc[:, 0], sc[:, 1], sc[:, 2:4] = args[:, 10], args[:, 11], args[:, 2:4]
    else:
        sc[:, 0], sc[:, 1], sc[:, 2:6] = args[:, 14], args[:, 15], args[:, 2:6]
    return sc, float(args[0, 1])


@pytest.mark.parametrize("mech", MECHS)
def test_pack_budgets_unpack_against_captured_driver_calls(chem, mech):
    import torch
    dev = torch.device("cuda", 0)
    g = _load(mech)
    nv = NVAR[mech]
    T = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    chem.set_species_maps(mech, g["gas_m2k"], g["gas_k2m"], g["rad_m2k"], g["rad_k2m"])
    j2, j6, nkc, nbgs = chem.drive_dims(mech)
    assert g["sl1_in"].shape[1] == j2 * nkc and g["sion1_in"].shape[1] == j6 * nkc and g["bgs_in"].shape[1] == 2 * nbgs
    n = g["c_in"].shape[0]
    sc, dt = _scal(mech, g["args"])
    # ---- pack: C as the driver hands it to INTEGRATE_x.  Entries the driver does not set keep the caller's content: start from the
    #      capture with every entry the tables DO set poisoned, so that a missed assignment shows
    start = g["c_in"].copy()
    poison = np.ones(start.shape[1], bool)
    from oracle import pack_py
    tab = pack_py.load(mech)
    Cp, _, _ = pack_py.pack(tab, np.full(start.shape[1], -7.0), g["s1_in"][0], g["s3_in"][0], g["sl1_in"][0], g["sion1_in"][0], sc[0, 0], sc[0, 1], sc[0, 2:6],
                            g["gas_m2k"], g["rad_m2k"])
    poison = Cp != -7.0
    start[:, poison] = -7.0
    var, fix = T(start[:, :nv]), T(start[:, nv:])
    sl1, sion1 = T(g["sl1_in"]), T(g["sion1_in"])
    chem.pack(mech, T(g["s1_in"]), T(g["s3_in"]), sl1, sion1, T(sc), var, fix)
    torch.cuda.synchronize()
    got = np.concatenate([var.cpu().numpy(), fix.cpu().numpy()], axis=1)
    assert np.array_equal(got, g["c_in"]), "%s: C handed to INTEGRATE differs from the reference's" % mech
    # ---- budgets on the state the reference's integration left
    bg, bgs = T(g["bg_in"]), T(g["bgs_in"])
    chem.budgets(mech, T(g["c_out"][:, :nv]), T(g["c_out"][:, nv:]), T(g["rconst"]), dt, bg, bgs)
    # ---- hand-over
    s1, s3 = T(g["s1_in"]), T(g["s3_in"])
    chem.unpack(mech, T(g["c_out"][:, :nv]), s1, s3, sl1, sion1)
    torch.cuda.synchronize()
    assert np.array_equal(bgs.cpu().numpy(), g["bgs_out"]), "bgs"
    lev = g["level"] > 0
    assert lev.sum() >= 1 and np.array_equal(bg.cpu().numpy()[lev], g["bg_out"][lev]), "bg"
    for t, key in ((s1, "s1_out"), (s3, "s3_out"), (sl1, "sl1_out"), (sion1, "sion1_out")):
        assert np.array_equal(t.cpu().numpy(), g[key]), key
    print("%s: pack, budgets and hand-over of %d captured driver calls bit-identical (%d budget levels)" % (mech, n, int(lev.sum())))


@pytest.mark.parametrize("mech", MECHS)
def test_clamps_keep_minus_zero_and_nan_like_the_compiled_max(chem, mech):
    """MAX(0.d0, x) of the drivers as flang compiles it: -0.0 stays -0.0 and NaN stays NaN (oracle/pack_py.py: fmax0, pinned on the CPU).  The
    liquid-phase arrays of a captured call get -0.0, NaN and negative entries; pack (clamped sl1 / sion1 and C) and the hand-over must agree
    with the restatement bit for bit, signs of zero and NaN payload-free positions included."""
    import torch
    from oracle import pack_py
    dev = torch.device("cuda", 0)
    g = _load(mech)
    nv = NVAR[mech]
    T = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    chem.set_species_maps(mech, g["gas_m2k"], g["gas_k2m"], g["rad_m2k"], g["rad_k2m"])
    tab = pack_py.load(mech)
    sc, dt = _scal(mech, g["args"])
    sl1, sion1 = g["sl1_in"][:1].copy(), g["sion1_in"][:1].copy()
    rng = np.random.default_rng(7)
    for arr in (sl1, sion1):
        idx = rng.permutation(arr.shape[1])
        arr[0, idx[0::4]] = -0.0
        arr[0, idx[1::4]] = np.nan
        arr[0, idx[2::4]] = -1.5e-9
    want_c, want_l, want_i = pack_py.pack(tab, g["c_in"][0], g["s1_in"][0], g["s3_in"][0], sl1[0], sion1[0], sc[0, 0], sc[0, 1], sc[0, 2:6], g["gas_m2k"], g["rad_m2k"])
    var, fix, tl, ti = T(g["c_in"][:1, :nv]), T(g["c_in"][:1, nv:]), T(sl1), T(sion1)
    chem.pack(mech, T(g["s1_in"][:1]), T(g["s3_in"][:1]), tl, ti, T(sc[:1]), var, fix)
    torch.cuda.synchronize()
    def same(a, b):      # NaN in the same places, every other entry the same BITS (the sign of a zero included)
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.where(np.isnan(a), 0.0, a).view(np.uint64), np.where(np.isnan(b), 0.0, b).view(np.uint64))
    got_c = np.concatenate([var.cpu().numpy()[0], fix.cpu().numpy()[0]])
    assert same(got_c, want_c), "C"
    if tab["preclamp"]:
        assert same(tl.cpu().numpy()[0], want_l) and same(ti.cpu().numpy()[0], want_i), "clamped model arrays"
    # hand-over of a state with -0.0 / NaN / negative liquid-phase species
    c_out = g["c_out"][0].copy()
    lst = sorted({c for _, _, _, c, cl in tab["unpack"] if cl})
    if lst:
        c_out[np.array(lst[0::3]) - 1] = -0.0
        c_out[np.array(lst[1::3]) - 1] = np.nan
        c_out[np.array(lst[2::3]) - 1] = -2.0e-12
    ws1, ws3, wl, wi = pack_py.unpack(tab, c_out, g["s1_in"][0], g["s3_in"][0], want_l, want_i, g["gas_k2m"], g["rad_k2m"])
    s1, s3 = T(g["s1_in"][:1]), T(g["s3_in"][:1])
    chem.unpack(mech, T(c_out[None, :nv]), s1, s3, tl, ti)
    torch.cuda.synchronize()
    assert same(tl.cpu().numpy()[0], wl) and same(ti.cpu().numpy()[0], wi) and same(s1.cpu().numpy()[0], ws1) and same(s3.cpu().numpy()[0], ws3)


@pytest.mark.parametrize("mech", MECHS)
def test_device_resident_driver_chain(chem, mech, oracles):
    """mistra_chem_drive_device: the whole driver for a batch of layers — pack -> env <- C -> Update_RCONST_x -> INTEGRATE_x -> budgets ->
    hand-over — with only the model arrays and the rate evaluator's inputs crossing PCIe.  The inputs are those of captured driver
    calls (the env vector as MISTRA_RATES_ENV_x packed it inside the model at that call's Update_RCONST_x); the concentrations in it
    are blanked here, the chain has to refill them from its own packed C.  Against the reference's end state of the same driver call:
    the integrator's stated tolerance; /Statistics/ against the oracle on the reference's RCONST."""
    import json
    import torch
    dev = torch.device("cuda", 0)
    g = _load(mech)
    nv = NVAR[mech]
    T = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    chem.set_species_maps(mech, g["gas_m2k"], g["gas_k2m"], g["rad_m2k"], g["rad_k2m"])
    n = g["c_in"].shape[0]
    sc, dt = _scal(mech, g["args"])
    names = json.load(open(os.path.join(REPO, "mistra_amd", "mech", mech + ".rates_env.json")))["env"]
    env = g["env"].copy()
    conc = [i for i, nm in enumerate(names) if nm.startswith(("c(", "fix("))]
    assert conc
    env[:, conc] = np.nan
    # entries of C the driver does not set carry over from COMMON in the reference: give the chain the same
    var, fix = T(g["c_in"][:, :nv]), T(g["c_in"][:, nv:])
    s1, s3, sl1, sion1 = T(g["s1_in"]), T(g["s3_in"]), T(g["sl1_in"]), T(g["sion1_in"])
    bg, bgs = T(g["bg_in"]), T(g["bgs_in"])
    ierr = torch.empty(n, dtype=torch.int32, device=dev)
    stats = torch.empty((n, 8), dtype=torch.int32, device=dev)
    th = torch.empty((n, 2), dtype=torch.float64, device=dev)
    envt = T(env)
    chem.drive(mech, s1, s3, sl1, sion1, T(sc), envt, var, fix, 0.0, dt, ierr, stats, th, bg, bgs)
    torch.cuda.synchronize()
    assert np.all(ierr.cpu().numpy() == 1)
    assert np.array_equal(envt.cpu().numpy(), g["env"]), "the concentrations of the rate evaluator's input were not refilled from the packed C"
    want, oierr, st = oracles[mech].integrate_batch(g["c_in"][:, :nv], g["c_in"][:, nv:], g["rconst"], 0.0, dt)
    assert np.array_equal(stats.cpu().numpy(), st), "/Statistics/"
    assert np.allclose(th.cpu().numpy()[:, 0], dt, rtol=1e-12)
    d = rel_diff(var.cpu().numpy(), g["c_out"][:, :nv]).max()
    assert d <= 2e-5        # the reference's own end state of the same driver call
    for t, key in ((s1, "s1_out"), (s3, "s3_out"), (sl1, "sl1_out"), (sion1, "sion1_out")):
        want_arr, got_arr = g[key], t.cpu().numpy()
        floor = 1e-12 * np.abs(g["c_out"][:, :nv]).max(axis=1, keepdims=True)
        assert (np.abs(got_arr - want_arr) / (np.abs(want_arr) + floor)).max() <= 2e-5, key
    # budgets are products of the end state and of RCONST (device transcendentals: 1e-13): relative to the layer's largest rate
    for got_b, want_b, sel in ((bgs.cpu().numpy(), g["bgs_out"], slice(None)), (bg.cpu().numpy(), g["bg_out"], g["level"] > 0)):
        scale = np.abs(want_b[sel]).max(axis=1, keepdims=True) + 1e-300
        assert (np.abs(got_b[sel] - want_b[sel]) / scale).max() <= 2e-5
    print("%s: the whole driver on the device, %d layers: end state within %.1e of the reference's, /Statistics/ identical" % (mech, n, d))
