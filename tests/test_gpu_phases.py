"""Phase-level parity on the GPU (-m gpu): the kernel's first-step dump variant (mistra_chem_debug_first_step) hands out
Fcn0, Ghimj as prepared, Ghimj as factorised, the pivots' reciprocals, K(1..3) and the error norm of the first attempt of
the first step, and each is compared with what it should be — so that a difference at the end can be pinned on a phase:

  Fun_x, ros_PrepareMatrix_x      bit for bit against the oracle (same operation order, no contraction)
  KppDecomp_x                     against the TEST-ONLY emulator of the kernel's own programs (tests/emu), which repeats the
                                  kernel's arithmetic operation by operation (LDS VM rounds, scaling pass, the dense tail
                                  block's MFMA steps as fused multiply-add chains), and against the oracle's factors to the
                                  bound tests/test_schedule.py holds the emulator to
  KppSolve_x (K1, K2, K3)         against the oracle's solve with the oracle's factors, stage by stage on the GPU's own
                                  previous-stage vectors
  ros_ErrorNorm_x                 against the oracle's formula on the GPU's K vectors

Plus the error paths no captured call reaches: a crafted zero pivot (Nsng = 1, H halved, integration goes on), six in a
row (IERR = -8), and backward integration (TOUT < TIN, Direction = -1)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import MECHS, REPO, rel_diff

pytestmark = pytest.mark.gpu
GAMMA = (0.43586652150845899941601945119356, 0.24291996454816804366592249683314, 0.21851380027664058511513169485832e+01)
C21, C31, C32 = -0.10156171083877702091975600115545e+01, 0.40759956452537699824805835358067e+01, 0.92076794298330791242156818474003e+01
E = (0.5, -0.29079558716805469821718236208017e+01, 0.22354069897811569627360909276199e+00)
dp = C.POINTER(C.c_double)


def P(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def chem():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mistra_amd import chem as c
    c.init(0)
    c.lib().mistra_chem_debug_first_step.argtypes = [C.c_int, C.c_int, dp, dp, dp, C.c_double, C.c_double, dp]
    return c


@pytest.fixture(scope="module")
def emu():
    d = os.path.join(REPO, "tests", "emu")
    subprocess.run(["make", "-s", "-C", d], check=True)
    lib = C.CDLL(os.path.join(d, "libschedule_emu.so"))
    lib.emu_create.restype = C.c_void_p
    lib.emu_create.argtypes = [C.c_char_p, C.c_int]
    lib.emu_lu.argtypes = [C.c_void_p, dp, dp, dp]
    lib.emu_tail_h.argtypes = [C.c_void_p]
    lib.emu_solve_backward.argtypes = [C.c_void_p, dp, dp]
    lib.emu_solve_kernel_form.argtypes = [C.c_void_p, dp, dp]
    return lib


def first_step(chem, mech, var, fix, rconst, tin=0.0, tout=10.0):
    from mistra_amd.chem import DIMS, MECH_IDS
    nvar, _, _, nnz = DIMS[mech]
    n = var.shape[0]
    dump = np.zeros((n, 5 * nvar + 2 * nnz + 2))
    v, f, r = (np.ascontiguousarray(x, np.float64) for x in (var, fix, rconst))
    assert chem.lib().mistra_chem_debug_first_step(MECH_IDS[mech], n, P(v), P(f), P(r), tin, tout, P(dump)) == 0, chem.lib().mistra_chem_last_error()
    o = 0
    out = {}
    for name, ln in (("fcn0", nvar), ("ghimj", nnz), ("lu", nnz), ("r", nvar), ("k1", nvar), ("k2", nvar), ("k3", nvar), ("err", 1), ("h", 1)):
        out[name] = dump[:, o:o + ln]
        o += ln
    return out


@pytest.mark.parametrize("mech", MECHS)
def test_phases_of_the_first_step(chem, emu, mech, golden, oracles):
    from mistra_amd.mechtab import load
    o, g, t = oracles[mech], golden[mech], load(mech)
    nt = {"gas": 64, "aer": 256, "tot": 512}[mech]      # the workgroup sizes ros3_kernel.hpp instantiates (gas: one wavefront per cell)
    h_emu = emu.emu_create(os.path.join(REPO, "mistra_amd", "mech", mech + ".mech").encode(), nt)
    cells = [0, 7, len(g["var_in"]) - 1]
    V, F, K = g["var_in"][cells], g["fix"][cells], g["rconst"][cells]
    d = first_step(chem, mech, V, F, K)
    worst = dict(lu_vs_emu=0.0, lu_vs_oracle=0.0, k=0.0, err=0.0)
    for i in range(len(cells)):
        v, f, k = (np.ascontiguousarray(x[i]) for x in (V, F, K))
        H = d["h"][i, 0]
        assert H == 1.0e-3                                                    # Hstart (gas.f:743), first attempt
        # ---- Fun_x
        fcn0 = o.fun(v, f, k)
        assert np.array_equal(d["fcn0"][i], fcn0), "Fun_x differs from the oracle"
        # ---- ros_PrepareMatrix_x
        ghinv = 1.0 / (H * GAMMA[0])
        G = -o.jac_sp(v, f, k)
        G[t.diag] += ghinv
        assert np.array_equal(d["ghimj"][i], G), "Ghimj = 1/(H*gamma) - Jac0 differs from the oracle"
        # ---- KppDecomp_x: the kernel's programs, repeated on the host operation by operation
        lu_emu, r_emu, x = G.copy(), np.empty(o.nvar), fcn0.copy()
        assert emu.emu_lu(h_emu, P(lu_emu), P(r_emu), P(x)) == 0
        rowmax = np.array([np.abs(lu_emu[t.crow[r]:t.crow[r + 1]]).max() for r in range(o.nvar)])
        scale = np.repeat(rowmax, np.diff(t.crow))
        worst["lu_vs_emu"] = max(worst["lu_vs_emu"], (np.abs(d["lu"][i] - lu_emu) / scale).max())
        assert np.array_equal(d["lu"][i], lu_emu), "factors differ from the emulated kernel programs (max %.1e of the row maximum)" % (np.abs(d["lu"][i] - lu_emu) / scale).max()
        assert np.array_equal(d["r"][i], r_emu), "pivot reciprocals differ from the emulated kernel programs"
        # ... and KppDecomp_x itself: the reference's factors, un-scaling the rows the kernel keeps row-scaled
        lu_ref, ier = o.decomp(G)
        assert ier == 0
        lu_un = d["lu"][i].copy()
        for r in range(emu.emu_tail_h(h_emu), o.nvar):
            lu_un[t.diag[r] + 1:t.crow[r + 1]] *= lu_un[t.diag[r]]
        rowmax_ref = np.repeat(np.array([np.abs(lu_ref[t.crow[r]:t.crow[r + 1]]).max() for r in range(o.nvar)]), np.diff(t.crow))
        worst["lu_vs_oracle"] = max(worst["lu_vs_oracle"], (np.abs(lu_un - lu_ref) / rowmax_ref).max())
        # (multipliers formed as W*R instead of W/U(j,j), carried through the elimination: 1e-16 per operation times the
        #  growth of the factorisation of this particular matrix — 3e-9 on the worst of the captured aer cells)
        assert (np.abs(lu_un - lu_ref) / rowmax_ref).max() <= 1e-7
        # ---- KppSolve_x, stage by stage on the GPU's own vectors (gas.f:1236-1262)
        k1, k2, k3 = d["k1"][i], d["k2"][i], d["k3"][i]
        want1 = o.solve(lu_ref, fcn0)
        fcn = o.fun(v + k1, f, k)
        want2 = o.solve(lu_ref, fcn + (C21 / H) * k1)
        want3 = o.solve(lu_ref, (fcn + (C31 / H) * k1) + (C32 / H) * k2)
        for got, want in ((k1, want1), (k2, want2), (k3, want3)):
            e = np.abs(got - want).max() / np.abs(want).max()
            worst["k"] = max(worst["k"], e)
            assert e <= 1e-9
        # ... and bit for bit against the emulated solve programs (head sweeps, tail chain — for tot with the cross-lane chain over
        # the dense block) run on the GPU's own factors and right-hand sides: same operations, same order, same fused multiply-adds
        lu_gpu = np.ascontiguousarray(d["lu"][i])
        e1 = x.copy()                                   # (emu_lu above carried the stage-1 right-hand side through the elimination)
        assert emu.emu_solve_backward(h_emu, P(lu_gpu), P(e1)) == 0
        e2 = np.ascontiguousarray((fcn + (C21 / H) * k1) + (H * GAMMA[1]) * 0.0)
        assert emu.emu_solve_kernel_form(h_emu, P(lu_gpu), P(e2)) == 0
        e3 = np.ascontiguousarray(((fcn + (C31 / H) * k1) + (C32 / H) * k2) + (H * GAMMA[2]) * 0.0)
        assert emu.emu_solve_kernel_form(h_emu, P(lu_gpu), P(e3)) == 0
        assert np.array_equal(k1, e1), "K1 differs from the emulated solve programs"      # (the forward-swept stage-1 vector is the emulator's: its factors are the GPU's, bit for bit — asserted above)
        assert np.array_equal(k2, e2), "K2 differs from the emulated solve programs"
        assert np.array_equal(k3, e3), "K3 differs from the emulated solve programs"
        worst["k_bitwise"] = worst.get("k_bitwise", 0) + 1
        # ---- ros_ErrorNorm_x (gas.f:1341) on the GPU's K vectors
        ynew = ((v + k1) + 0.61697947043828245592553615689730e+01 * k2) + -0.42772256543218573326238373806514e+00 * k3
        yerr = ((0.0 + E[0] * k1) + E[1] * k2) + E[2] * k3
        sc = 1.0e-25 + 1.0e-3 * np.maximum(np.abs(v), np.abs(ynew))
        err = np.sqrt(((yerr / sc) ** 2).sum() / o.nvar)
        worst["err"] = max(worst["err"], abs(d["err"][i, 0] - err) / err)
        assert abs(d["err"][i, 0] - err) <= 1e-12 * err
    print("%s phases: Fun and Ghimj bit-exact; LU vs emulated kernel %.1e, vs reference factors %.1e (of row max); "
          "K vectors %.1e vs the oracle's solve, bit-identical to the emulated solve programs in %d cells; error norm %.1e"
          % (mech, worst["lu_vs_emu"], worst["lu_vs_oracle"], worst["k"], worst.get("k_bitwise", 0), worst["err"]))


def _first_order_losses(t):
    """(reaction, species) for reactions A = k*V(s) whose only effect on s is the loss -A: Jac0(s,s) gets exactly -k from it"""
    out = []
    for r in range(t.nreact):
        fac = t.a_fac[t.a_ptr[r]:t.a_ptr[r + 1]]
        if len(fac) != 1 or fac[0] >= t.nvar:
            continue
        s = int(fac[0])
        terms = [(int(t.vd_idx[p]), float(t.vd_coef[p])) for p in range(t.vd_ptr[s], t.vd_ptr[s + 1])]
        if (r, -1.0) in terms:
            out.append((r, s))
    return out


@pytest.mark.parametrize("mech", MECHS)
def test_zero_pivot_paths(chem, mech, golden, oracles):
    """ros_PrepareMatrix_x's singular branch (gas.f:1439-1467): a first-order loss with the NEGATIVE rate constant
    k = -1/(H*gamma) puts an exact zero on the diagonal of Ghimj at step size H.  One such reaction: one failed
    decomposition, H halved, the step goes on (Nsng = 1, Ndec = Nstp + 1).  Six of them, tuned to H, H/2 .. H/32: six
    failures in a row, IERR = -8.  Kernel against oracle: IERR, /Statistics/, state."""
    from mistra_amd.mechtab import load
    o, g, t = oracles[mech], golden[mech], load(mech)
    losses, seen = [], set()
    for r, s in _first_order_losses(t):
        if s not in seen:
            seen.add(s)
            losses.append((r, s))
    assert len(losses) >= 6
    V, F = g["var_in"][:1].copy(), g["fix"][:1].copy()
    tin, tout = 0.0, 1.5e-3
    for nzero, want_ierr in ((1, 1), (6, -8)):
        K = np.zeros((1, t.nreact))
        for i in range(nzero):
            K[0, losses[i][0]] = -1.0 / ((1.0e-3 / 2 ** i) * GAMMA[0])
        # the premise: the oracle's own Jacobian has the exact zero on the diagonal at the first attempt
        G = -o.jac_sp(V[0], F[0], K[0])
        G[t.diag] += 1.0 / (1.0e-3 * GAMMA[0])
        assert G[t.diag[losses[0][1]]] == 0.0
        want, ierr, st = o.integrate_batch(V, F, K, tin, tout)
        res = chem.integrate(mech, V, F, K, tin, tout)
        assert int(ierr[0]) == want_ierr and int(res.ierr[0]) == want_ierr
        assert np.array_equal(res.stats, st), (res.stats, st)
        assert int(st[0, 7]) == (1 if nzero == 1 else 6)                       # Nsng
        assert np.allclose(res.var, want, rtol=1e-9, atol=1e-300)
        # the row KppDecomp_x reports (IER = first row with an exactly zero diagonal, gas.f:6157) and ros_PrepareMatrix_x prints
        # (gas.f:1456): at H/2**i the loss of species losses[i] cancels 1/(H*gamma)
        rows = chem.singular_rows(mech, 0)
        assert list(rows[:nzero]) == [losses[i][1] + 1 for i in range(nzero)], rows
        _, ier = o.decomp(G)
        assert ier == rows[0]
    print(mech, "zero-pivot paths: Nsng = 1 continues, six in a row end with IERR = -8; statistics identical to the oracle's")


def test_backward_integration(chem, golden, oracles):
    """TOUT < TIN: Direction = -1 (gas.f:1188-1192), every H-dependent factor changes sign."""
    g, o = golden["gas"], oracles["gas"]
    V, F, K = g["var_in"][:3], g["fix"][:3], g["rconst"][:3]
    want, ierr, st = o.integrate_batch(V, F, K, 2.0e-3, 0.0)
    res = chem.integrate("gas", V, F, K, 2.0e-3, 0.0)
    assert np.array_equal(res.ierr, ierr) and np.array_equal(res.stats, st)
    assert np.allclose(res.var, want, rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("mech", MECHS)
def test_too_many_steps_exit(chem, mech, golden, oracles):
    """IERR = -6, "No of steps exceeds maximum bound" (gas.f:1199-1202): the bound Max_no_steps = 100000 that INTEGRATE_x leaves in place
    is out of a test's reach, so kernel and oracle both get 5 through their test hooks (mistra_chem_debug_set_max_steps /
    kpp_set_max_steps; the oracle's exit is pinned against the compiled reference's Rosenbrock_x called with IPAR(3) = 5,
    tests/test_oracle.py).  Code, /Statistics/, exit time, last step and the state reached, cells that finish below the bound among them."""
    from oracle import oracle as om
    g, o = golden[mech], oracles[mech]
    n = min(12, g["var_in"].shape[0])
    V, F, K = g["var_in"][:n], g["fix"][:n], g["rconst"][:n]
    touts = (10.0, 1.0e-6)      # (the second horizon is short enough for every cell to finish below the bound: the ordinary exit under the same hook)
    try:
        om.set_max_steps(5)
        chem.debug_set_max_steps(5)
        seen = set()
        for tout in touts:
            want, ierr, st = o.integrate_batch(V, F, K, 0.0, tout)
            res = chem.integrate(mech, V, F, K, 0.0, tout)
            assert np.array_equal(res.ierr, ierr), (res.ierr, ierr)
            assert np.array_equal(res.stats, st)
            assert rel_diff(res.var, want).max() <= 2e-5
            seen |= set(int(x) for x in ierr)
        assert -6 in seen and (mech == "tot" or 1 in seen), seen      # (the cloudy tot cells reject their way past the bound on any horizon)
    finally:
        om.set_max_steps(0)
        chem.debug_set_max_steps(0)
    res = chem.integrate(mech, V[:2], F[:2], K[:2], 0.0, 10.0)
    assert np.all(res.ierr == 1) and np.array_equal(res.stats[:, 2], g["stats"][:2, 2])
