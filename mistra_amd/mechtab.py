"""Loader for the committed mechanism tables `mistra_amd/mech/{gas,aer,tot}.mech` (written by tools/extract_mech.py).

The tables restate, as data, what the reference's generated mechanism files define:
`Fun_x` / `Jac_SP_x` term lists (gas.f:2043,2656 | aer.f:2741,4368 | tot.f:4145,6845) and the LU sparsity pattern
`LU_ICOL/LU_CROW/LU_DIAG` (gas.f:6718 | aer.f:23480 | tot.f:44435).  All indices are 0-based here.
"""
import os
import numpy as np

MAGIC = 0x48434D4B
VERSION = 2
MECH_IDS = {"gas": 0, "aer": 1, "tot": 2}
MECH_NAMES = ("gas", "aer", "tot")
MECH_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mech")


class MechTables:
    """Plain container; attribute names follow tools/extract_mech.py."""

    def __init__(self, name, path=None):
        self.name = name
        path = path or os.path.join(MECH_DIR, name + ".mech")
        raw = open(path, "rb").read()
        hdr = np.frombuffer(raw, np.int32, 12)
        if hdr[0] != MAGIC or hdr[1] != VERSION:
            raise ValueError("%s: not a KMCH v%d mechanism table" % (path, VERSION))
        (self.nvar, self.nfix, self.nreact, self.nnz, n_afac, self.nb, n_bfac, n_vd, n_jv, self.nconst) = \
            (int(x) for x in hdr[2:])
        off = 48
        def take(n, dt):
            nonlocal off
            a = np.frombuffer(raw, dt, n, off).copy()
            off += a.nbytes
            return a
        self.crow = take(self.nvar + 1, np.int32)
        self.icol = take(self.nnz, np.int32)
        self.diag = take(self.nvar, np.int32)
        self.a_ptr = take(self.nreact + 1, np.int32)
        self.a_fac = take(n_afac, np.int32)
        self.b_rct = take(self.nb, np.int32)
        self.b_ptr = take(self.nb + 1, np.int32)
        self.b_fac = take(n_bfac, np.int32)
        self.vd_ptr = take(self.nvar + 1, np.int32)
        self.vd_idx = take(n_vd, np.int32)
        self.jv_ptr = take(self.nnz + 1, np.int32)
        self.jv_idx = take(n_jv, np.int32)
        off += (-off) % 8
        self.vd_coef = take(n_vd, np.float64)
        self.jv_coef = take(n_jv, np.float64)
        self.consts = take(self.nconst, np.float64)
        assert off == len(raw), (off, len(raw))

    @property
    def nspec(self):
        return self.nvar + self.nfix


_cache = {}


def load(name):
    if name not in _cache:
        _cache[name] = MechTables(name)
    return _cache[name]


def flop_counts(name):
    """Floating-point operations of ONE internal Ros3 step of the reference's algorithm, COUNTED from the mechanism tables (SURVEY.md
    §8d's accounting): KppDecomp_x's multiply-adds and quotients by walking its loops over the LU pattern (gas.f:6142-6176), three
    KppSolve_x, three Fun_x (the reference also evaluates the one behind ros_FunTimeDerivative_x), one Jac_SP_x, the Ghimj build and the
    stage / error vectors.  A multiply-add counts 2.  -> dict of the parts and their sum 'step'."""
    t = load(name)
    rows = [t.icol[t.crow[k]:t.crow[k + 1]] for k in range(t.nvar)]
    upper = t.crow[1:] - t.diag - 1                    # entries right of the diagonal per row
    lu_fma = sum(int(upper[j]) for k in range(t.nvar) for j in rows[k] if j < k)
    lu_div = sum(int((rows[k] < k).sum()) for k in range(t.nvar))
    solve = 2 * (t.nnz - t.nvar) + t.nvar              # forward + backward multiply-adds, one quotient per row
    fun = len(t.a_fac) + len(t.vd_idx)                 # products RCT*V*..  +  signed sums
    jac = len(t.b_fac) + len(t.jv_idx)
    parts = {"lu": 2 * lu_fma + lu_div, "lu_fma": lu_fma, "lu_div": lu_div, "solves": 3 * solve, "fun": 3 * fun, "jac": jac,
             "prepare": t.nnz + t.nvar, "vectors": 12 * t.nvar}
    parts["step"] = parts["lu"] + parts["solves"] + parts["fun"] + parts["jac"] + parts["prepare"] + parts["vectors"]
    return parts
