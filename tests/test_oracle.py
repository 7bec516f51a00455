"""The oracle (oracle/kpp_ros3.c) against the reference: committed golden vectors everywhere, and — where the compiled
reference exists (oracle/_ref, this container) — the reference's own routines called directly."""
import numpy as np
import pytest

from conftest import EXTRA_SETS, MECHS, load_golden
from oracle.oracle import Reference


@pytest.mark.parametrize("mech", MECHS)
def test_oracle_reproduces_captured_reference_calls_bit_exactly(mech, golden, oracles):
    g, o = golden[mech], oracles[mech]
    for i in range(g["var_in"].shape[0]):
        v, ierr, st, te, he = o.integrate(g["var_in"][i], g["fix"][i], g["rconst"][i], g["tin"][i], g["tout"][i])
        assert ierr == 1
        assert np.array_equal(v, g["var_out"][i]), "record %d: concentrations differ from the reference" % i
        assert np.array_equal(st, g["stats"][i]), "record %d: /Statistics/ differ" % i
        assert te == g["tin_out"][i] and he == g["stepmin_out"][i]


@pytest.mark.parametrize("which,mech", EXTRA_SETS)
def test_oracle_reproduces_further_reference_captures_bit_exactly(which, mech, golden, oracles):
    g, o = load_golden(mech, "_" + which), oracles[mech]
    if which == "day":     # the set differs from the night one where it should: photolysis rate constants are no longer all zero
        night_off = (golden[mech]["rconst"] == 0).all(axis=0)
        assert (night_off & ~(g["rconst"] == 0).all(axis=0)).sum() >= 20
    for i in range(g["var_in"].shape[0]):
        v, ierr, st, te, he = o.integrate(g["var_in"][i], g["fix"][i], g["rconst"][i], g["tin"][i], g["tout"][i])
        assert ierr == 1
        assert np.array_equal(v, g["var_out"][i]), "record %d: concentrations differ from the reference" % i
        assert np.array_equal(st, g["stats"][i]), "record %d: /Statistics/ differ" % i
        assert te == g["tin_out"][i] and he == g["stepmin_out"][i]


@pytest.mark.parametrize("mech", MECHS)
def test_golden_files_are_sane(mech, golden):
    g = golden[mech]
    n = g["var_in"].shape[0]
    assert n >= 32 and g["var_out"].shape == g["var_in"].shape
    assert np.all(g["tin"] == 0.0) and np.all(g["tout"] == 10.0)          # dd = 10 s, str.f90:126
    assert np.all(g["stats"][:, 2] >= 7)                                  # 1e-3 * 6^k growth: 7 steps is the floor
    assert "flang" in str(g["provenance"])


@pytest.mark.skipif(not Reference.available(), reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("mech", MECHS)
def test_oracle_functions_match_compiled_reference_bit_exactly(mech, golden, oracles):
    g, o, r = golden[mech], oracles[mech], Reference(mech)
    from mistra_amd.mechtab import load
    t = load(mech)
    rng = np.random.default_rng(7)
    for i in (0, g["var_in"].shape[0] // 2, g["var_in"].shape[0] - 1):
        V, F, K = g["var_in"][i], g["fix"][i], g["rconst"][i]
        assert np.array_equal(o.fun(V, F, K), r.fun(V, F, K))                               # Fun_x
        j = o.jac_sp(V, F, K)
        assert np.array_equal(j, r.jac_sp(V, F, K))                                         # Jac_SP_x
        G = -j
        G[t.diag] += 1.0 / (1e-3 * 0.43586652150845899941601945119356)                      # ros_PrepareMatrix_x
        lu_o, ier_o = o.decomp(G)
        lu_r, ier_r = r.decomp(G)
        assert ier_o == ier_r == 0 and np.array_equal(lu_o, lu_r)                           # KppDecomp_x
        b = rng.normal(size=o.nvar)
        assert np.array_equal(o.solve(lu_o, b), r.solve(lu_r, b))                           # KppSolve_x
        out, st, te, he = r.integrate(V, F, K)                                              # INTEGRATE_x
        v, ierr, st_o, te_o, he_o = o.integrate(V, F, K)
        assert np.array_equal(v, out) and np.array_equal(st_o, st) and te == te_o and he == he_o


@pytest.mark.parametrize("mech", MECHS)
def test_oracle_zero_pivot_and_failure_codes(mech, oracles):
    o = oracles[mech]
    from mistra_amd.mechtab import load
    t = load(mech)
    G = np.ones(o.nnz)
    G[t.diag[3]] = 0.0
    _, ier = o.decomp(G)
    assert ier == 4                                    # KppDecomp_x returns the 1-based row of the zero diagonal (gas.f:6157)
    # NaN input: every error estimate is NaN -> rejected; the reject loop (gas.f:1264-1331) has no exit test of its own, so
    # H shrinks until it underflows to 0, `H <= Hmin` then "accepts" a zero-length step and the outer test (gas.f:1236)
    # ends the call with IERR = -7.  (With the compiled reference this input never returns: MAX(FacMin, NaN) is NaN there
    # and H stays NaN — the oracle and the kernel deliberately use the NaN-ignoring MIN/MAX so that every call terminates.)
    V = np.full(o.nvar, np.nan)
    v, ierr, st, te, he = o.integrate(V, np.ones(o.nfix), np.ones(o.nreact))
    assert ierr == -7 and te == 0.0 and st[3] == 1 and st[2] > 300
    # zero-length interval: the time loop is not entered, state untouched, IERR = 1
    V = np.full(o.nvar, 1e-10)
    v, ierr, st, te, he = o.integrate(V, np.ones(o.nfix), np.ones(o.nreact), 5.0, 5.0)
    assert ierr == 1 and np.array_equal(v, V) and st[2] == 0 and te == 5.0


@pytest.mark.parametrize("mech", MECHS)
def test_reference_sensitivity_bounds_the_parity_tolerance(mech, golden, oracles):
    """How far does the reference's own algorithm move under legal re-association (fma contraction, other summation
    direction in the backward sweep, pivot quotients formed with the reciprocal)?  That spread is what the GPU parity tolerance (tests/test_gpu_parity.py) is set
    against: 2e-5 for all species must be well above it, and the step bookkeeping must not change."""
    from conftest import rel_diff
    from oracle.oracle import set_variant
    g, o = golden[mech], oracles[mech]
    worst = 0.0
    try:
        for v in (1, 2, 3, 4, 5, 7):
            set_variant(v)
            out, ierr, st = o.integrate_batch(g["var_in"], g["fix"], g["rconst"])
            assert np.array_equal(st, g["stats"]) and np.all(ierr == 1)
            worst = max(worst, rel_diff(out, g["var_out"]).max())
    finally:
        set_variant(0)
    print("%s: reference spread under re-association %.3e" % (mech, worst))
    assert worst <= 5e-6


@pytest.mark.skipif(not Reference.available(), reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("mech", MECHS)
def test_too_many_steps_exit_matches_the_compiled_reference(mech, golden, oracles):
    """IERR = -6 (gas.f:1199-1202): INTEGRATE_x leaves Max_no_steps at 100000, out of a test's reach, so both sides get the bound through
    the option that exists for it — the compiled reference's Rosenbrock_x is called as INTEGRATE_x calls it but with IPAR(3) = 5, the
    oracle through kpp_set_max_steps.  The integrator returns at the head of the first step that finds Nstp > 5 (rejected attempts count) with whatever Y it reached: code,
    counters, exit time, last step and state must be the reference's bit for bit."""
    from oracle import oracle as om
    g, o, r = golden[mech], oracles[mech], Reference(mech)
    try:
        om.set_max_steps(5)
        for i in (0, g["var_in"].shape[0] - 1):
            v, ierr, st, te, he = o.integrate(g["var_in"][i], g["fix"][i], g["rconst"][i], 0.0, 10.0)
            rv, rierr, rst, rte, rhe = r.rosenbrock(g["var_in"][i], g["fix"][i], g["rconst"][i], 0.0, 10.0, max_steps=5)
            assert ierr == rierr == -6
            assert np.array_equal(st, rst) and st[2] > 5
            assert te == rte and he == rhe and te < 10.0
            assert np.array_equal(v, rv)
    finally:
        om.set_max_steps(0)
    # with the bound lifted the same call is the ordinary one again
    v, ierr, st, te, he = o.integrate(g["var_in"][0], g["fix"][0], g["rconst"][0], 0.0, 10.0)
    assert ierr == 1 and np.array_equal(v, g["var_out"][0])
