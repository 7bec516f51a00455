"""Host logic: the schedule compiler (mistra_amd/csrc/schedule.cpp) run through the TEST-ONLY emulator
(tests/emu/schedule_emu.cpp) against the oracle.  Fun, Jac and the matrix preparation keep the reference's operation
order and must be bit-exact; the LU forms its multipliers with the pivot's reciprocal and the solve applies the
backward-sweep terms in readiness order: both are checked to round-off."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO

EMU_DIR = os.path.join(REPO, "tests", "emu")
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)
GAMMA1 = 0.43586652150845899941601945119356


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["make", "-s", "-C", EMU_DIR], check=True)
    lib = C.CDLL(os.path.join(EMU_DIR, "libschedule_emu.so"))
    lib.emu_create.restype = C.c_void_p
    lib.emu_create.argtypes = [C.c_char_p, C.c_int]
    lib.emu_describe.restype = C.c_char_p
    lib.emu_describe.argtypes = [C.c_void_p]
    lib.emu_tail_h.argtypes = [C.c_void_p]
    lib.emu_dense_nd.argtypes = [C.c_void_p]
    lib.emu_lu.argtypes = [C.c_void_p, dp, dp, dp]
    lib.emu_solve_backward.argtypes = [C.c_void_p, dp, dp]
    lib.emu_solve.argtypes = [C.c_void_p, dp, dp]
    lib.emu_solve_split.argtypes = [C.c_void_p, dp, dp]
    lib.emu_fun.argtypes = [C.c_void_p, dp, dp, dp, dp]
    lib.emu_jac_prepare.argtypes = [C.c_void_p, dp, dp, dp, C.c_double, C.c_int, dp]
    lib.emu_round_profile.argtypes = [C.c_void_p, C.c_int, ip, ip, C.c_int]
    return lib


def P(a):
    return a.ctypes.data_as(dp)


# the workgroup sizes the library instantiates (capi.cpp) plus one more per mechanism
@pytest.mark.parametrize("mech,nt", [("gas", 128), ("gas", 64), ("aer", 512), ("aer", 320), ("tot", 512), ("tot", 1024)])
def test_programs_match_oracle(emu, mech, nt, golden, oracles):
    from mistra_amd.mechtab import load
    o, g, t = oracles[mech], golden[mech], load(mech)
    h = emu.emu_create(os.path.join(REPO, "mistra_amd", "mech", mech + ".mech").encode(), nt)
    assert h, "schedule compiler failed"
    assert b"LU:" in emu.emu_describe(h)
    rng = np.random.default_rng(11)
    for i in (0, g["var_in"].shape[0] - 1):
        V, F, K = (np.ascontiguousarray(g[k][i]) for k in ("var_in", "fix", "rconst"))
        f = np.empty(o.nvar)
        emu.emu_fun(h, P(V), P(F), P(K), P(f))
        assert np.array_equal(f, o.fun(V, F, K))
        J = np.empty(o.nnz)
        emu.emu_jac_prepare(h, P(V), P(F), P(K), 0.0, 0, P(J))
        j_ref = o.jac_sp(V, F, K)
        assert np.array_equal(J, j_ref)
        ghinv = 1.0 / (1e-3 * GAMMA1)
        G = np.empty(o.nnz)
        emu.emu_jac_prepare(h, P(V), P(F), P(K), ghinv, 1, P(G))
        G_ref = -j_ref
        G_ref[t.diag] += ghinv
        assert np.array_equal(G, G_ref) and np.array_equal(np.signbit(G), np.signbit(G_ref))
        lu_ref, ier = o.decomp(G_ref)
        assert ier == 0
        R = np.empty(o.nvar)
        b1 = rng.normal(size=o.nvar) * np.abs(o.fun(V, F, K)).max()
        y1 = b1.copy()             # rides through the factorisation: the forward sweep of the stage-1 right-hand side
        assert emu.emu_lu(h, P(G), P(R), P(y1)) == 0, "hazard inside an LU round"
        # multipliers are formed as W*R(j) instead of W/U(j,j) (schedule.hpp): last-bit differences that the elimination
        # carries along; entries are compared against the scale of their row
        rowmax = np.array([np.abs(lu_ref[t.crow[k]:t.crow[k + 1]]).max() for k in range(o.nvar)])
        scale = np.repeat(rowmax, np.diff(t.crow))
        # the program leaves the tail block's upper triangle row-scaled, U'(i,c) = U(i,c)*R(i) (schedule.hpp: TailSolve)
        tail_h = emu.emu_tail_h(h)
        G_un = G.copy()
        for k in range(tail_h, o.nvar):
            G_un[t.diag[k] + 1:t.crow[k + 1]] *= G[t.diag[k]]
        print(mech, nt, "LU max |diff|/rowmax %.2e" % (np.abs(G_un - lu_ref) / scale).max())
        assert (np.abs(G_un - lu_ref) / scale).max() <= 1e-10
        assert np.allclose(R, 1.0 / lu_ref[t.diag], rtol=1e-9, atol=0)
        b = rng.normal(size=o.nvar) * np.abs(o.fun(V, F, K)).max()
        x = b.copy()
        assert emu.emu_solve(h, P(lu_ref), P(x)) == 0, "hazard inside a solve round"
        x_ref = o.solve(lu_ref, b)
        assert np.abs(x - x_ref).max() <= 1e-11 * np.abs(x_ref).max()
        # stage 1 as the kernel runs it: forward sweep inside the LU program, then only the backward half
        lu_k = G.copy()
        assert emu.emu_solve_backward(h, P(lu_k), P(y1)) == 0
        x1_ref = o.solve(lu_ref, b1)
        assert np.abs(y1 - x1_ref).max() <= 1e-10 * np.abs(x1_ref).max()
        # the form the kernel runs: head rows through the VM, the tail chain by one wave in registers
        x2 = b.copy()
        assert emu.emu_solve_split(h, P(lu_ref), P(x2)) == 0, "hazard inside a head round"
        assert np.abs(x2 - x_ref).max() <= 1e-11 * np.abs(x_ref).max()


def test_round_structure_tot(emu):
    """The eager schedule keeps the critical path short.  With the last 64 rows factorised as a dense block in registers
    (schedule.hpp: DenseTail) the LU program is left with the ~25 rounds of the head pivots."""
    h = emu.emu_create(os.path.join(REPO, "mistra_amd", "mech", "tot.mech").encode(), 512)
    crit = np.zeros(512, np.int32)
    tot = np.zeros(512, np.int32)
    n_lu = emu.emu_round_profile(h, 0, crit.ctypes.data_as(ip), tot.ctypes.data_as(ip), 512)
    assert emu.emu_dense_nd(h) == 64
    assert 15 < n_lu < 40 and crit[:n_lu].sum() < 200
    n_sv = emu.emu_round_profile(h, 1, crit.ctypes.data_as(ip), tot.ctypes.data_as(ip), 512)
    assert 100 < n_sv < 200 and crit[:n_sv].sum() < 800


def test_flop_counts_come_from_the_tables():
    """bench.py's FP64 fraction uses operation counts WALKED from the mechanism tables (mistra_amd/mechtab.py: flop_counts), not scaled guesses:
    for tot they reproduce the survey's own count of the reference's loops (KppDecomp_x 214 377 multiply-adds + 6 176
    quotients, KppSolve_x 13 086 + 417, Fun_x 2 718 + 5 441, Jac_SP_x 2 297 + 9 423)."""
    from mistra_amd.mechtab import flop_counts, load
    c = flop_counts("tot")
    assert (c["lu_fma"], c["lu_div"]) == (214377, 6176)
    assert c["solves"] == 3 * (2 * 13086 + 417) and c["fun"] == 3 * (2718 + 5441) and c["jac"] == 2297 + 9423
    assert c["step"] == 569818
    assert flop_counts("gas")["step"] == 23527 and flop_counts("aer")["step"] == 213354
    for m in ("gas", "aer"):
        t, c = load(m), flop_counts(m)
        assert c["lu_div"] == int(t.diag.sum() - t.crow[:-1].sum())      # one quotient per strictly-lower entry
