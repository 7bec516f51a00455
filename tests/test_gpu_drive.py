"""The single-pass batched driver on the GPU (-m gpu; SURVEY §8 f1 + f2 delivered to the model): ONE
mistra_chem_drive call per mechanism does pack -> Update_RCONST_x -> INTEGRATE_x -> bud_x / bud_s_x -> hand-over for all layers of a
10-s column step, on the model's own arrays in host memory.  Fixtures: tests/golden/drivecol_<case>.npz — every gas_drive / aer_drive /
tot_drive call of ONE whole step of the running reference model in model order (what the driver read and what it left behind), joined
with the /Statistics/ of the same INTEGRATE_x calls.

  * through the C ABI from Python (host arrays, rows of other layers poisoned): C as packed bit-identical to the reference's, /Statistics/
    and exit times identical, model arrays and budgets within the integrator's stated tolerance of the reference's end state, every row the
    step does not own untouched — and everything BIT-IDENTICAL to the device-resident chain (mistra_chem_drive_device) whose pack / budget /
    hand-over kernels tests/test_gpu_pack.py pins bit for bit against the same kind of captures;
  * from Fortran (shim/shim_driver D): the staging calls the patched x_drive makes (KPP_DRIVE_STAGE_x reading the COMMON blocks) and
    kpp_drive_run_arrays behind the loop, same checks, and the WALL TIME of the whole step (host staging + device call) printed beside
    the reference's cost of the same step."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu
MECHS = ("gas", "aer", "tot")
NVAR = {"gas": 102, "aer": 257, "tot": 417}
NFIX = {"gas": 3, "aer": 5, "tot": 7}
NREACT = {"gas": 331, "aer": 979, "tot": 1627}
NENV = {"gas": 74, "aer": 330, "tot": 544}
N, NLEV, NRXN, NBGS = 150, 15, 1627, 122      # global_params.f90: n, nlev, nrxn; bud_s_g.f:63
CASES = ["Joyce2014", "base1", "BTZ96"]
DRIVER = os.path.join(REPO, "shim", "shim_driver")
POISON = -7.25


@pytest.fixture(scope="module")
def chem():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mistra_amd import chem as c
    c.init(0)
    return c


def _load(case):
    return dict(np.load(os.path.join(REPO, "tests", "golden", "drivecol_%s.npz" % case)))


def _model_arrays(g):
    """The model's arrays as they stood in front of the step: rows of the step's layers from the capture, every other row poisoned."""
    j1, j5 = g["s1_in"].shape[1], g["s3_in"].shape[1]
    a = dict(s1=np.full((N, j1), POISON), s3=np.full((N, j5), POISON), sl1=np.full((N, g["sl1_in"].shape[1]), POISON),
             sion1=np.full((N, g["sion1_in"].shape[1]), POISON), bgs=np.full((N, NBGS, 2), POISON), bg=np.full((NLEV, NRXN, 2), POISON))
    k = g["k"] - 1
    a["s1"][k], a["s3"][k], a["sl1"][k], a["sion1"][k] = g["s1_in"], g["s3_in"], g["sl1_in"], g["sion1_in"]
    a["bgs"][k] = g["bgs_in"].reshape(-1, NBGS, 2)
    for j, i in enumerate(g["bg_layers"]):
        nr = NREACT[MECHS[g["mech"][i]]]
        a["bg"][g["level"][i] - 1, :nr] = g["bg_in"][j, :2 * nr].reshape(nr, 2)
    return a


def _rel(got, want, scale):
    return float((np.abs(got - want) / (np.abs(want) + scale)).max()) if got.size else 0.0


def _check_step(g, a, per_layer, written_of):
    """a: the model arrays after the step; per_layer: {layer index i: (ierr, stats[8], texit)}"""
    k = g["k"] - 1
    untouched = np.ones(N, bool)
    untouched[k] = False
    for key in ("s1", "s3", "sl1", "sion1", "bgs"):
        assert np.all(a[key][untouched] == POISON), "%s: a row of a layer outside the step was written" % key
    lev_rows = np.zeros(NLEV, bool)
    lev_rows[g["level"][g["bg_layers"]] - 1] = True
    assert np.all(a["bg"][~lev_rows] == POISON)
    worst = 0.0
    for i in range(len(k)):
        mech = MECHS[g["mech"][i]]
        nv = NVAR[mech]
        ierr, stats, texit = per_layer[i]
        assert ierr == 1
        assert np.array_equal(stats, g["stats"][i]), "layer %d (k=%d, %s): /Statistics/ %s vs the reference's %s" % (i, k[i] + 1, mech, stats, g["stats"][i])
        assert texit == g["tin_out"][i]
        floor = 1e-12 * np.abs(g["c_out"][i, :nv]).max()
        for key in ("s1", "s3", "sl1", "sion1"):
            worst = max(worst, _rel(a[key][k[i]], g[key + "_out"][i], floor))
        b_want = g["bgs_out"][i].reshape(NBGS, 2)
        worst = max(worst, _rel(a["bgs"][k[i]], b_want, 1e-12 * np.abs(b_want).max() + 1e-300))
    for j, i in enumerate(g["bg_layers"]):
        nr = NREACT[MECHS[g["mech"][i]]]
        want = g["bg_out"][j, :2 * nr].reshape(nr, 2)
        got = a["bg"][g["level"][i] - 1]
        assert np.all(got[nr:] == POISON), "bg: slots past the mechanism's NREACT were written"
        worst = max(worst, _rel(got[:nr], want, 1e-9 * np.abs(want).max() + 1e-300))
    assert worst <= 2e-5, worst
    return worst


def _written_mask(mech, g, i):
    """entries of C the pack half sets (the others are KPP's dummy products)"""
    from oracle import pack_py
    tab = pack_py.load(mech)
    nv, nf = NVAR[mech], NFIX[mech]
    Cp, _, _ = pack_py.pack(tab, np.full(nv + nf, POISON), g["s1_in"][i], g["s3_in"][i], g["sl1_in"][i], g["sion1_in"][i], g["scal"][i, 0], g["scal"][i, 1],
                            g["scal"][i, 2:6], g[mech + "_gas_m2k"], g[mech + "_rad_m2k"])
    return Cp != POISON


@pytest.mark.parametrize("case", CASES)
def test_host_driver_call_against_a_captured_column_step(chem, case):
    import torch
    g = _load(case)
    a = _model_arrays(g)
    dev = torch.device("cuda", 0)
    per_layer, worst_bits = {}, 0
    for m, mech in enumerate(MECHS):
        idx = np.nonzero(g["mech"] == m)[0]
        if idx.size == 0:
            continue
        nv, nf, ne = NVAR[mech], NFIX[mech], NENV[mech]
        chem.set_species_maps(mech, g[mech + "_gas_m2k"], g[mech + "_gas_k2m"], g[mech + "_rad_m2k"], g[mech + "_rad_k2m"])
        names = json.load(open(os.path.join(REPO, "mistra_amd", "mech", mech + ".rates_env.json")))["env"]
        env = g["env"][idx, :ne].copy()
        env[:, [j for j, nm in enumerate(names) if nm.startswith(("c(", "fix("))]] = np.nan      # the device refills them from its own packed C
        level = g["level"][idx]
        before = {key: a[key].copy() for key in a}
        ierr, stats, th, cp = chem.drive_host(mech, g["k"][idx], a["s1"], a["s3"], a["sl1"], a["sion1"], g["scal"][idx], env, 0.0, 10.0, bg=a["bg"],
                                              bg_level=level, bgs=a["bgs"], want_c=True)
        for j, i in enumerate(idx):
            per_layer[int(i)] = (int(ierr[j]), stats[j], float(th[j, 0]))
        # ---- the pack half: C as handed to INTEGRATE_x, bit for bit; KPP's dummy products start from 0
        w = _written_mask(mech, g, idx[0])
        assert np.array_equal(cp[:, w], g["c_in"][idx][:, :nv + nf][:, w]), "%s: packed C differs from the reference's" % mech
        assert np.all(cp[:, ~w] == 0.0)
        # ---- the same layers through the device-resident chain: identical bits everywhere
        T = lambda x: torch.tensor(np.ascontiguousarray(x), device=dev)
        kk = g["k"][idx] - 1
        d = dict(s1=T(before["s1"][kk]), s3=T(before["s3"][kk]), sl1=T(before["sl1"][kk]), sion1=T(before["sion1"][kk]), bgs=T(before["bgs"][kk]))
        bgd = np.zeros((idx.size, NREACT[mech], 2))
        for j in range(idx.size):
            if level[j] > 0:
                bgd[j] = before["bg"][level[j] - 1, :NREACT[mech]]
        bgt = T(bgd)
        var, fix = torch.zeros((idx.size, nv), dtype=torch.float64, device=dev), torch.zeros((idx.size, nf), dtype=torch.float64, device=dev)
        di, ds = torch.empty(idx.size, dtype=torch.int32, device=dev), torch.empty((idx.size, 8), dtype=torch.int32, device=dev)
        chem.drive(mech, d["s1"], d["s3"], d["sl1"], d["sion1"], T(g["scal"][idx]), T(env), var, fix, 0.0, 10.0, di, ds, None, bgt, d["bgs"])
        torch.cuda.synchronize()
        for key in ("s1", "s3", "sl1", "sion1", "bgs"):
            assert np.array_equal(d[key].cpu().numpy(), a[key][kk]), "%s: host-array call and device-resident chain differ" % key
        assert np.array_equal(ds.cpu().numpy(), stats)
        for j in range(idx.size):
            if level[j] > 0:
                assert np.array_equal(bgt.cpu().numpy()[j], a["bg"][level[j] - 1, :NREACT[mech]])
        worst_bits += idx.size
    worst = _check_step(g, a, per_layer, None)
    print("%s: %d layers (%s) through mistra_chem_drive: packed C bit-identical, /Statistics/ identical, arrays within %.1e of the reference's end state, "
          "bit-identical to the device-resident chain" % (case, worst_bits, np.bincount(g["mech"], minlength=3).tolist(), worst))


@pytest.mark.parametrize("case", ["base1", "BTZ96"])
def test_mechanisms_of_a_step_side_by_side(chem, case):
    """mistra_chem_drive_begin for every mechanism of the step, then mistra_chem_drive_end in mechanism order (what KPP_DRIVE_RUN does): the batches touch
    disjoint rows of the model arrays and run on streams of their own — every bit as one mechanism after the other, and the step lasts as long as its
    slowest mechanism."""
    import time
    g = _load(case)

    for m, mech in enumerate(MECHS):      # (once, as the model does: the call uploads tables and waits for the device)
        if (g["mech"] == m).any():
            chem.set_species_maps(mech, g[mech + "_gas_m2k"], g[mech + "_gas_k2m"], g[mech + "_rad_m2k"], g[mech + "_rad_k2m"])

    def step(side_by_side):
        a = _model_arrays(g)
        open_, out, keep = [], {}, []
        for m, mech in enumerate(MECHS):
            idx = np.nonzero(g["mech"] == m)[0]
            if idx.size == 0:
                continue
            r = chem.drive_host(mech, g["k"][idx], a["s1"], a["s3"], a["sl1"], a["sion1"], g["scal"][idx], g["env"][idx, :NENV[mech]].copy(), 0.0, 10.0, bg=a["bg"],
                                bg_level=g["level"][idx], bgs=a["bgs"], begin_only=side_by_side)
            out[mech] = r[:3]
            keep.append(r)
            if side_by_side:
                open_.append(mech)
        for mech in open_:
            chem.drive_host_end(mech)
        return a, out

    a1, o1 = step(False)
    a2, o2 = step(True)
    for key in a1:
        assert np.array_equal(a1[key], a2[key]), "%s differs between the mechanisms one after the other and side by side" % key
    for mech in o1:
        for x, y in zip(o1[mech], o2[mech]):
            assert np.array_equal(x, y)
    with pytest.raises(chem.MistraChemError):
        chem.drive_host_end("gas")      # nothing open
    t = {}
    for mode in (False, True, False, True):
        t0 = time.perf_counter(); step(mode); t.setdefault(mode, []).append((time.perf_counter() - t0) * 1e3)
    print("%s: column step, mechanisms one after the other %.2f ms, side by side %.2f ms (wall, Python staging included)" % (case, min(t[False]), min(t[True])))
    assert min(t[True]) < 1.05 * min(t[False])


needs_flang = pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/flang"), reason="no Fortran compiler here")


@needs_flang
@pytest.mark.parametrize("case", CASES)
def test_fortran_single_pass_column_step_and_its_wall_time(case, tmp_path, capsys):
    """The column step as the patched kpp_driver runs it (shim/kpp_drive.patch): per layer the COMMON blocks get the layer's values and
    KPP_DRIVE_STAGE_x records its share, behind the loop ONE kpp_drive_run_arrays.  Same checks as above; the wall time of the staging
    loop and of the device call(s) is printed for INTEGRATION.md's table."""
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "shim")], check=True)
    g = _load(case)
    a = _model_arrays(g)
    j1, j5 = a["s1"].shape[1], a["s3"].shape[1]
    nl, nrep = len(g["k"]), 6
    il = np.zeros(NLEV)
    il[g["level"][g["bg_layers"]] - 1] = g["k"][g["bg_layers"]]
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        np.array([N, j1, j5, NLEV, NRXN, nl, nrep], np.float64).tofile(f)
        first = next(m for m in MECHS if m + "_gas_m2k" in g)
        for mech in MECHS:
            src = mech if mech + "_gas_m2k" in g else first          # (a mechanism without layers in this step: its maps are never used)
            for key in ("gas_m2k", "gas_k2m", "rad_m2k", "rad_k2m"):
                g[src + "_" + key].astype(np.float64).tofile(f)
        il.tofile(f)
        for key in ("s1", "s3", "sl1", "sion1", "bg", "bgs"):
            a[key].tofile(f)
        for i in range(nl):
            mech = MECHS[g["mech"][i]]
            np.array([g["mech"][i] + 1, g["k"][i], g["scal"][i, 0], g["scal"][i, 1]], np.float64).tofile(f)
            g["env"][i, :NENV[mech]].tofile(f)
    subprocess.run([DRIVER, "D", str(fin), str(fout)], check=True, timeout=600)
    raw = np.fromfile(fout, np.float64)
    off = 0
    for key in ("s1", "s3", "sl1", "sion1", "bg", "bgs"):
        a[key] = raw[off:off + a[key].size].reshape(a[key].shape)
        off += a[key].size
    rec = raw[off:off + 13 * nl].reshape(nl, 13)
    times = raw[off + 13 * nl:].reshape(nrep, 2)
    per_layer = {}
    for m in range(3):       # the driver reports per mechanism in layer order
        idx = np.nonzero(g["mech"] == m)[0]
        mine = rec[rec[:, 0] == m + 1]
        assert len(mine) == len(idx) and np.array_equal(mine[:, 1].astype(int), g["k"][idx])
        for j, i in enumerate(idx):
            per_layer[int(i)] = (int(mine[j, 2]), mine[j, 3:11].astype(np.int32), float(mine[j, 11]))
    worst = _check_step(g, a, per_layer, None)
    counts = np.bincount(g["mech"], minlength=3)
    best = times[1:].sum(axis=1).argmin() + 1
    # the reference's cost of the same step on one 2.1 GHz core of the build container: SURVEY.md §6 (gas: 69 us per layer inside kpp_driver,
    # pack + rates + integrator; aer ~0.6 ms, tot ~24 ms per layer)
    ref_ms = counts[0] * 0.069 + counts[1] * 0.6 + counts[2] * 24.0
    with capsys.disabled():
        print("\n  column %s through the single-pass batched driver from Fortran: %d gas + %d aer + %d tot layers; staging loop %.3f ms + device call(s) %.3f ms"
              " = %.3f ms per 10-s step (best of %d; first call %.1f ms); reference ~%.1f ms; worst deviation from the reference's end state %.1e"
              % (case, counts[0], counts[1], counts[2], times[best, 0], times[best, 1], times[best].sum(), nrep - 1, times[0].sum(), ref_ms, worst))
