"""Parity tests proper (-m gpu): the HIP path, called through the C ABI, against the captured reference calls
(tests/golden) and against the oracle on seeded synthetic batches.

Tolerance.  The kernel keeps the reference's operation order in Fun, Jac_SP, the matrix preparation, the LU and the
forward sweep, uses no FMA contraction and IEEE division; it departs from the reference in three places only:
the error norm is a tree reduction (gas.f:1360 sums sequentially), pow() in the step-size factor is the device
library's (<= 1 ulp from libm's), and the backward sweep subtracts its terms in readiness order.  Each perturbs a
step size or a K vector at the 1e-16 level.  How far such perturbations move the answer is a property of the
chemistry, not of the kernel: the REFERENCE ALGORITHM ITSELF, re-associated the way another compiler would
(a*b+c contracted to fma, or the backward sweep summed in the other direction — oracle.set_variant, measured in
test_reference_sensitivity below and on the CPU in tests/test_oracle.py), moves by up to 1.9e-6 (aer), 2.4e-7 (tot),
1e-16 (gas) in the worst species, which are trace species ~1e-9 of the largest concentration.
Stated bound, every species of every cell:   |dc| / (|c| + 1e-12 * max|c| of the cell)  <=  2e-5   (10x that spread),
species above 1e-4 of the cell maximum <= 1e-12 (their spread under re-association is 1e-15), and identical step
bookkeeping (COMMON /Statistics/).
"""
import numpy as np
import pytest

from conftest import EXTRA_SETS, MECHS, load_golden, rel_diff

pytestmark = pytest.mark.gpu
RTOL = 2e-5          # all species, floor 1e-12 of the cell maximum
RTOL_MAJOR = 1e-12   # species above 1e-4 of the cell maximum


def check(got, want, tag=""):
    d = rel_diff(got, want)
    major = np.abs(want) >= 1e-4 * np.abs(want).max(axis=1, keepdims=True)
    dm = np.where(major, d, 0.0)
    print("%s max rel diff %.3e (all species), %.3e (major species), median of per-cell max %.3e"
          % (tag, d.max() if d.size else 0.0, dm.max() if d.size else 0.0, np.median(d.max(axis=1)) if d.size else 0.0))
    assert d.size == 0 or (d.max() <= RTOL and dm.max() <= RTOL_MAJOR)


@pytest.fixture(scope="module")
def chem():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mistra_amd import chem as c
    c.init(0)
    return c


@pytest.mark.parametrize("mech", MECHS)
def test_golden_reference_calls_host_buffers(chem, mech, golden):
    g = golden[mech]
    res = chem.integrate(mech, g["var_in"], g["fix"], g["rconst"], 0.0, 10.0)
    assert np.all(res.ierr == 1)
    check(res.var, g["var_out"], "%s: %d reference calls," % (mech, len(res.var)))
    assert np.array_equal(res.stats, g["stats"]), "COMMON /Statistics/ differs from the reference"


@pytest.mark.parametrize("which,mech", EXTRA_SETS)
def test_golden_further_reference_captures(chem, which, mech):
    g = load_golden(mech, "_" + which)
    res = chem.integrate(mech, g["var_in"], g["fix"], g["rconst"], 0.0, 10.0)
    assert np.all(res.ierr == 1)
    check(res.var, g["var_out"], "%s: %d reference calls of set '%s'," % (mech, len(res.var), which))
    assert np.array_equal(res.stats, g["stats"]), "COMMON /Statistics/ differs from the reference"


@pytest.mark.parametrize("mech", MECHS)
def test_golden_reference_calls_device_buffers(chem, mech, golden):
    import torch
    g = golden[mech]
    dev = torch.device("cuda", 0)
    res = chem.integrate(mech, torch.tensor(g["var_in"], device=dev), torch.tensor(g["fix"], device=dev),
                         torch.tensor(g["rconst"], device=dev))
    torch.cuda.synchronize()
    check(res.var.cpu().numpy(), g["var_out"], mech + " device buffers:")
    assert np.array_equal(res.stats.cpu().numpy()[:, 2:5], g["stats"][:, 2:5])


@pytest.mark.parametrize("mech,ncell", [("gas", 1000), ("aer", 300), ("tot", 200)])
def test_synthetic_batch_against_oracle(chem, mech, ncell, oracles):
    """BASELINE configs[1] (gas, 1000 cells) and seeded perturbed batches of the other two, sized for the oracle."""
    import torch
    from mistra_amd.workload import make_batch
    var, fix, rconst = make_batch(mech, 0, ncell, torch.device("cuda", 0))
    res = chem.integrate(mech, var, fix, rconst)
    torch.cuda.synchronize()
    want, ierr, st = oracles[mech].integrate_batch(var.cpu().numpy(), fix.cpu().numpy(), rconst.cpu().numpy())
    assert np.all(ierr == 1) and np.all(res.ierr.cpu().numpy() == 1)
    check(res.var.cpu().numpy(), want, "%s: %d synthetic cells, steps/cell %.1f," % (mech, ncell, st[:, 2].mean()))
    assert np.array_equal(res.stats.cpu().numpy(), st)


def test_replicated_cells_are_identical(chem, golden):
    """BASELINE configs[1]: 1000 copies of one captured gas state -> 1000 identical answers, 7 steps each."""
    g = golden["gas"]
    var = np.repeat(g["var_in"][:1], 1000, axis=0)
    fix = np.repeat(g["fix"][:1], 1000, axis=0)
    rconst = np.repeat(g["rconst"][:1], 1000, axis=0)
    res = chem.integrate("gas", var, fix, rconst)
    assert np.all(res.var == res.var[0]) and np.all(res.stats[:, 2] == 7)
    assert rel_diff(res.var[:1], g["var_out"][:1]).max() <= RTOL


def test_edge_cases(chem, golden, oracles):
    g = golden["gas"]
    # empty batch
    res = chem.integrate("gas", np.zeros((0, 102)), np.zeros((0, 3)), np.zeros((0, 331)))
    assert res.var.shape == (0, 102)
    # single cell, in-place (var_out aliases var_in inside the library's staging buffer)
    res = chem.integrate("gas", g["var_in"][:1], g["fix"][:1], g["rconst"][:1])
    assert rel_diff(res.var, g["var_out"][:1]).max() <= RTOL
    # zero-length interval: untouched state, IERR = 1, no steps
    res = chem.integrate("gas", g["var_in"][:3], g["fix"][:3], g["rconst"][:3], 5.0, 5.0)
    assert np.array_equal(res.var, g["var_in"][:3]) and np.all(res.ierr == 1) and np.all(res.stats[:, 2] == 0)
    # ragged intervals: a shorter and a longer chemistry timestep than the model's 10 s
    for tout in (1.0, 60.0):
        res = chem.integrate("gas", g["var_in"][:4], g["fix"][:4], g["rconst"][:4], 0.0, tout)
        want, ierr, st = oracles["gas"].integrate_batch(g["var_in"][:4], g["fix"][:4], g["rconst"][:4], 0.0, tout)
        assert rel_diff(res.var, want).max() <= RTOL and np.array_equal(res.stats, st)
    # failure code: NaN state -> IERR = -7 ("step size too small"), same bookkeeping as the oracle, and the call returns
    bad = g["var_in"][:2].copy()
    bad[1, :] = np.nan
    res = chem.integrate("gas", bad, g["fix"][:2], g["rconst"][:2])
    want, ierr, st = oracles["gas"].integrate_batch(bad, g["fix"][:2], g["rconst"][:2])
    assert list(res.ierr) == [1, -7] and list(ierr) == [1, -7]
    assert np.array_equal(res.stats, st)
    assert rel_diff(res.var[:1], want[:1]).max() <= RTOL


def test_full_size_properties(chem):
    """BASELINE configs[2] at full size (tot, 100 000 cells): properties that do not need the oracle on every cell —
    all cells succeed, results finite and non-negative where the inputs were, cell order preserved (a strided sample is
    re-integrated alone and must reproduce bit for bit), and a sample is checked against the oracle."""
    import torch
    from mistra_amd.workload import make_batch
    from oracle.oracle import Oracle
    dev = torch.device("cuda", 0)
    n = 100000
    var, fix, rconst = make_batch("tot", 0, n, dev)
    res = chem.integrate("tot", var, fix, rconst)
    torch.cuda.synchronize()
    assert int((res.ierr != 1).sum()) == 0
    assert bool(torch.isfinite(res.var).all())
    steps = res.stats[:, 2].double()
    print("tot 1e5 cells: steps/cell mean %.1f min %d max %d" % (steps.mean().item(), int(steps.min()), int(steps.max())))
    sel = torch.arange(0, n, 9973, device=dev)
    again = chem.integrate("tot", var[sel].contiguous(), fix[sel].contiguous(), rconst[sel].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(again.var, res.var[sel]) and torch.equal(again.stats, res.stats[sel])
    want, ierr, st = Oracle("tot").integrate_batch(var[sel].cpu().numpy(), fix[sel].cpu().numpy(), rconst[sel].cpu().numpy())
    check(res.var[sel].cpu().numpy(), want, "tot 1e5 sample:")
    assert np.array_equal(res.stats[sel].cpu().numpy(), st)


def test_opt_in_hstart_reuse(chem, golden, oracles):
    """SURVEY §8 f4, opt-in and NOT the reference's behaviour: a first step size per cell (the previous call's last step)
    instead of INTEGRATE_x's 1e-3.  Default (no hstart, or entries <= 0) is the reference path bit for bit.  With reuse the kernel is
    checked against the ORACLE run with the same first step per cell (oracle.integrate_batch(hstart=...), the restated
    RosenbrockIntegrator_x started at that H instead of gas.f:743's constant): identical /Statistics/, the stated tolerance; and the
    second of two consecutive calls needs fewer steps than the reference path (study: profiles/r02_hstart_reuse_study.txt)."""
    import torch
    dev = torch.device("cuda", 0)
    g = golden["tot"]
    var, fix, rconst = (torch.tensor(g[k][:16], device=dev) for k in ("var_in", "fix", "rconst"))
    n = var.shape[0]

    def call(v, hstart):
        out = torch.empty_like(v)
        ierr = torch.empty(n, dtype=torch.int32, device=dev)
        stats = torch.empty((n, 8), dtype=torch.int32, device=dev)
        th = torch.empty((n, 2), dtype=torch.float64, device=dev)
        chem.integrate_into("tot", v, fix, rconst, out, ierr, stats, 0.0, 10.0, texit_hexit=th, hstart=hstart)
        torch.cuda.synchronize()
        assert int((ierr != 1).sum()) == 0
        return out, stats, th

    o1, s1, th1 = call(var, None)
    o1z, s1z, _ = call(var, torch.zeros(n, dtype=torch.float64, device=dev))
    assert torch.equal(o1, o1z) and torch.equal(s1, s1z)                 # entries <= 0: the reference's Hstart
    assert np.array_equal(s1.cpu().numpy(), g["stats"][:16])
    o2_ref, s2_ref, _ = call(o1, None)                                     # second timestep, as the reference runs it
    hstart = th1[:, 1].contiguous()
    o2, s2, _ = call(o1, hstart)                                           # ... and from the first call's last step size
    assert int(s2[:, 2].sum()) < int(s2_ref[:, 2].sum())
    # the oracle on the same inputs with the same first step per cell
    want, ierr, st = oracles["tot"].integrate_batch(o1.cpu().numpy(), g["fix"][:16], g["rconst"][:16], 0.0, 10.0, hstart=hstart.cpu().numpy())
    assert np.all(ierr == 1)
    assert np.array_equal(s2.cpu().numpy(), st), "/Statistics/ of the Hstart-reuse path differ from the oracle started at the same H"
    check(o2.cpu().numpy(), want, "tot, Hstart reuse vs the oracle at the same first steps:")
    # mixed: some cells reuse, some (entry 0) start at 1e-3 — per-cell, as the header says
    mixed = hstart.clone()
    mixed[::2] = 0.0
    o3, s3, _ = call(o1, mixed)
    want3, _, st3 = oracles["tot"].integrate_batch(o1.cpu().numpy(), g["fix"][:16], g["rconst"][:16], 0.0, 10.0, hstart=mixed.cpu().numpy())
    assert np.array_equal(s3.cpu().numpy(), st3)
    check(o3.cpu().numpy(), want3, "tot, Hstart reuse on every other cell:")
    d = rel_diff(o2.cpu().numpy(), o2_ref.cpu().numpy())
    print("Hstart reuse, second call: %d steps instead of %d; max rel diff to the reference path %.2e"
          % (int(s2[:, 2].sum()), int(s2_ref[:, 2].sum()), d.max()))
    assert d.max() <= 1e-3      # the integrator's own tolerance (RTOL 1e-3): both paths solve the same ODE
