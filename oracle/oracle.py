"""TEST INFRASTRUCTURE — ctypes access to the two CPU checkers.  Not product code; mistra_amd/ never imports this.

`Oracle(mech)`     the plain-C restatement, oracle/kpp_ros3.c  (libkpp_oracle.so, built by oracle/Makefile)
`Reference(mech)`  the reference itself, compiled by oracle/build_ref.sh from /root/reference/src into
                   oracle/_ref/libmistra_ref.so (Fortran symbols `integrate_t_`, `fun_t_`, ...; COMMON /GDATA_x/ as
                   `gdata_t_`, gas_Global.h:29-58).  Present only where it was built (this container, or prebuilt and
                   carried to the GPU box); `Reference.available()` says so.
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
MECH_DIR = os.path.join(REPO, "mistra_amd", "mech")
SFX = {"gas": "g", "aer": "a", "tot": "t"}
DIMS = {"gas": (102, 3, 331, 1110), "aer": (257, 5, 979, 6579), "tot": (417, 7, 1627, 13503)}  # nvar nfix nreact nnz

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def build_oracle():
    subprocess.run(["make", "-s", "-C", HERE], check=True)


_lib = None


def _oracle_lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "libkpp_oracle.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(HERE, "kpp_ros3.c")):
            build_oracle()
        lib = C.CDLL(path)
        lib.kpp_mech_load.restype = C.c_void_p
        lib.kpp_mech_load.argtypes = [C.c_char_p]
        lib.kpp_mech_dim.argtypes = [C.c_void_p, C.c_int]
        lib.kpp_work_doubles.restype = C.c_size_t
        lib.kpp_work_doubles.argtypes = [C.c_void_p]
        lib.kpp_fun.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp]
        lib.kpp_jac_sp.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp]
        lib.kpp_decomp.argtypes = [C.c_void_p, _dp, _dp]
        lib.kpp_solve.argtypes = [C.c_void_p, _dp, _dp]
        lib.kpp_integrate.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_double, C.c_double, _ip, _dp, _dp, _dp]
        lib.kpp_set_variant.argtypes = [C.c_int]
        lib.kpp_set_options.argtypes = [C.c_double, C.c_double, C.c_double]
        lib.kpp_integrate_batch.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, _ip, _ip]
        lib.kpp_integrate_batch_hstart.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, _dp, _ip, _ip]
        _lib = lib
    return _lib


def set_variant(v):
    """0 = pinned restatement; bit 0 descending backward sweep; bit 1 fma-contracted; bit 2 reciprocal instead of
    division by pivots (sensitivity studies only)"""
    _oracle_lib().kpp_set_variant(int(v))


def set_options(rtol=0.0, atol=0.0, hstart=0.0):
    """Study knobs of the oracle (0 = INTEGRATE_x's fixed value): tolerances and first step size (tools/hstart_study.py)."""
    _oracle_lib().kpp_set_options(float(rtol), float(atol), float(hstart))


def set_max_steps(n=0):
    """IPAR(3) of Rosenbrock_x (0 = INTEGRATE_x's 100000): makes the IERR = -6 exit testable."""
    _oracle_lib().kpp_set_max_steps(int(n))


class Oracle:
    def __init__(self, mech):
        self.mech = mech
        self.lib = _oracle_lib()
        self.h = self.lib.kpp_mech_load(os.path.join(MECH_DIR, mech + ".mech").encode())
        if not self.h:
            raise RuntimeError("cannot load mechanism table for " + mech)
        self.nvar, self.nfix, self.nreact, self.nnz = (self.lib.kpp_mech_dim(self.h, k) for k in range(4))
        self.work = np.zeros(self.lib.kpp_work_doubles(self.h))

    def fun(self, V, F, RCT):
        out = np.empty(self.nvar)
        self.lib.kpp_fun(self.h, _d(np.ascontiguousarray(V, np.float64)), _d(np.ascontiguousarray(F, np.float64)),
                         _d(np.ascontiguousarray(RCT, np.float64)), _d(out), _d(self.work))
        return out

    def jac_sp(self, V, F, RCT):
        out = np.empty(self.nnz)
        self.lib.kpp_jac_sp(self.h, _d(np.ascontiguousarray(V, np.float64)), _d(np.ascontiguousarray(F, np.float64)),
                            _d(np.ascontiguousarray(RCT, np.float64)), _d(out), _d(self.work))
        return out

    def decomp(self, JVS):
        a = np.array(JVS, np.float64)
        ier = self.lib.kpp_decomp(self.h, _d(a), _d(self.work))
        return a, ier

    def solve(self, LU, X):
        x = np.array(X, np.float64)
        self.lib.kpp_solve(self.h, _d(np.ascontiguousarray(LU, np.float64)), _d(x))
        return x

    def integrate(self, var, fix, rconst, tin=0.0, tout=10.0):
        """-> (var_out, ierr, stats[8], texit, hexit)"""
        v = np.array(var, np.float64)
        st = np.zeros(8, np.int32)
        te, he = C.c_double(), C.c_double()
        ierr = self.lib.kpp_integrate(self.h, _d(v), _d(np.ascontiguousarray(fix, np.float64)),
                                      _d(np.ascontiguousarray(rconst, np.float64)), tin, tout, _i(st),
                                      C.byref(te), C.byref(he), _d(self.work))
        return v, ierr, st, te.value, he.value

    def integrate_batch(self, var, fix, rconst, tin=0.0, tout=10.0, hstart=None):
        """cell-major arrays [ncell, n*] -> (var_out, ierr[ncell], stats[ncell,8]).  hstart [ncell]: a first step size per cell
        instead of INTEGRATE_x's 1e-3 (entries <= 0: the reference's value) — the kernel's opt-in mode, not the reference's."""
        v = np.array(var, np.float64, order="C")
        ncell = v.shape[0]
        ierr = np.zeros(ncell, np.int32)
        st = np.zeros((ncell, 8), np.int32)
        if hstart is not None:
            h = np.ascontiguousarray(hstart, np.float64).reshape(ncell)
            self.lib.kpp_integrate_batch_hstart(self.h, ncell, _d(v), _d(np.ascontiguousarray(fix, np.float64)),
                                                _d(np.ascontiguousarray(rconst, np.float64)), tin, tout, _d(h), _i(ierr), _i(st))
            return v, ierr, st
        self.lib.kpp_integrate_batch(self.h, ncell, _d(v), _d(np.ascontiguousarray(fix, np.float64)),
                                     _d(np.ascontiguousarray(rconst, np.float64)), tin, tout, _i(ierr), _i(st))
        return v, ierr, st


class Reference:
    """The compiled reference (oracle/_ref/libmistra_ref.so).  Non-reentrant like the Fortran it wraps."""
    PATH = os.path.join(HERE, "_ref", "libmistra_ref.so")
    _lib = None

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self, mech):
        if Reference._lib is None:
            Reference._lib = C.CDLL(self.PATH, mode=os.RTLD_LAZY)
        self.lib = Reference._lib
        self.mech, s = mech, SFX[mech]
        self.nvar, self.nfix, self.nreact, self.nnz = DIMS[mech]
        nvar, nfix, nreact = self.nvar, self.nfix, self.nreact

        class GData(C.Structure):      # COMMON /GDATA_x/  (gas_Global.h:29-58)
            _fields_ = [("c", C.c_double * (nvar + nfix)), ("rconst", C.c_double * nreact), ("time", C.c_double),
                        ("dt", C.c_double), ("atol", C.c_double * nvar), ("rtol", C.c_double * nvar),
                        ("stepmin", C.c_double), ("stepmax", C.c_double)]
        self.gdata = GData.in_dll(self.lib, "gdata_%s_" % s)
        self.stats = (C.c_int32 * 8).in_dll(self.lib, "statistics_")       # COMMON /Statistics/ (gas.f:913)
        self._fun = getattr(self.lib, "fun_%s_" % s)
        self._jac = getattr(self.lib, "jac_sp_%s_" % s)
        self._dec = getattr(self.lib, "kppdecomp_%s_" % s)
        self._sol = getattr(self.lib, "kppsolve_%s_" % s)
        self._int = getattr(self.lib, "integrate_%s_" % s)

    def fun(self, V, F, RCT):
        out = np.empty(self.nvar)
        self._fun(_d(np.ascontiguousarray(V, np.float64)), _d(np.ascontiguousarray(F, np.float64)),
                  _d(np.ascontiguousarray(RCT, np.float64)), _d(out))
        return out

    def jac_sp(self, V, F, RCT):
        out = np.empty(self.nnz)
        self._jac(_d(np.ascontiguousarray(V, np.float64)), _d(np.ascontiguousarray(F, np.float64)),
                  _d(np.ascontiguousarray(RCT, np.float64)), _d(out))
        return out

    def decomp(self, JVS):
        a = np.array(JVS, np.float64)
        ier = C.c_int32(0)
        self._dec(_d(a), C.byref(ier))
        return a, ier.value

    def solve(self, LU, X):
        x = np.array(X, np.float64)
        self._sol(_d(np.ascontiguousarray(LU, np.float64)), _d(x))
        return x

    # COMMON /kpp_rate_x/ member order (gas_Global.h:82-87 | aer_Global.h:82-88 | tot_Global.h:91-95); NSPEC = nvar + nfix
    RATE_LAYOUT = {
        "gas": [("yxkmtd", 2), ("yhenry", 0), ("yxeq", 0), ("ycwd", -2), ("conv1", -1), ("xhal", -1), ("xiod", -1), ("xhet1", -1), ("xhet2", -1)],
        "aer": [("yhenry", 0), ("yxkmt", 2), ("ykef", 2), ("ykeb", 2), ("yxkmtd", 2), ("yxeq", 0), ("ycw", -2), ("ycwd", -2), ("conv1", -1),
                ("cvv1", -1), ("cvv2", -1), ("xhal", -1), ("xiod", -1), ("xliq1", -1), ("xliq2", -1), ("xhet1", -1), ("xhet2", -1)],
        "tot": [("conv1", -1), ("cvv1", -1), ("cvv2", -1), ("cvv3", -1), ("cvv4", -1), ("xhal", -1), ("xiod", -1), ("xliq1", -1), ("xliq2", -1),
                ("xliq3", -1), ("xliq4", -1), ("ycw", -4), ("yhenry", 0), ("yxkmt", 4), ("ykef", 4), ("ykeb", 4), ("xhet1", -1), ("xhet2", -1),
                ("yxkmtd", 2), ("yxeq", 0), ("ycwd", -2)],
    }       # (name, k): k > 0 array (NSPEC, k) column-major; 0 array (NSPEC); -1 scalar; -n small array of n

    def update_rconst(self, names, env):
        """Update_RCONST_x (gas.f:275 | aer.f:304 | tot.f:1040) on one cell's inputs: `names` says what each entry of `env` is
        (mistra_amd/mech/<mech>.rates_env.json, written by tools/extract_rates.py).  Fills COMMON /cb_1/ (kpp.f90:7140),
        /kpp_rate_x/, /ph_r_x/ and C, calls the compiled reference routine, returns RCONST."""
        import re
        sfx = SFX[self.mech]
        nspec = self.nvar + self.nfix
        e = np.asarray(env, np.float64)
        off, total = {}, 0
        for nm, k in self.RATE_LAYOUT[self.mech]:
            off[nm] = (total, k)
            total += nspec * k if k > 0 else nspec if k == 0 else -k
        cb1 = (C.c_double * 4).in_dll(self.lib, "cb_1_")
        rate = np.ctypeslib.as_array((C.c_double * total).in_dll(self.lib, "kpp_rate_%s_" % sfx))
        ph = np.ctypeslib.as_array((C.c_double * 47).in_dll(self.lib, "ph_r_%s_" % sfx))
        c = np.ctypeslib.as_array(self.gdata.c)
        rate[:] = 0.0
        ph[:] = 0.0
        c[:] = 0.0
        for nm, v in zip(names, e):
            m = re.match(r"(\w+)\((\d+)(?:,(\d+))?\)$", nm)
            if nm in ("aircc", "te", "h2oppm", "pk"):
                cb1[("aircc", "te", "h2oppm", "pk").index(nm)] = v
            elif m is None:
                rate[off[nm][0]] = v
            else:
                arr, i, j = m.group(1), int(m.group(2)), int(m.group(3) or 1)
                if arr == "ph_rat":
                    ph[i - 1] = v
                elif arr == "c":
                    c[i - 1] = v
                elif arr == "fix":
                    c[self.nvar + i - 1] = v
                else:
                    o, k = off[arr]
                    rate[o + (j - 1) * nspec + (i - 1) if k > 0 else o + i - 1] = v
        getattr(self.lib, "update_rconst_%s_" % sfx)()
        return np.ctypeslib.as_array(self.gdata.rconst).copy()

    def integrate(self, var, fix, rconst, tin=0.0, tout=10.0):
        g = self.gdata
        np.ctypeslib.as_array(g.c)[:self.nvar] = var
        np.ctypeslib.as_array(g.c)[self.nvar:] = fix
        np.ctypeslib.as_array(g.rconst)[:] = rconst
        t0, t1 = C.c_double(tin), C.c_double(tout)
        self._int(C.byref(t0), C.byref(t1))
        out = np.ctypeslib.as_array(g.c)[:self.nvar].copy()
        st = np.array(list(self.stats), np.int32)
        return out, st, t0.value, g.stepmin


    def rosenbrock(self, var, fix, rconst, tin=0.0, tout=10.0, max_steps=0):
        """Rosenbrock_x (gas.f:777) called as INTEGRATE_x calls it (gas.f:729-750) but with IPAR(3) = max_steps, the one option that makes
        the "too many steps" exit (IERR = -6, gas.f:1199-1202) reachable -> (VAR, IERR, IPAR(11:18), Texit, Hexit)."""
        g, s = self.gdata, SFX[self.mech]
        np.ctypeslib.as_array(g.c)[:self.nvar] = var
        np.ctypeslib.as_array(g.c)[self.nvar:] = fix
        np.ctypeslib.as_array(g.rconst)[:] = rconst
        y = np.array(var, np.float64)
        atol, rtol = np.full(self.nvar, 1.0e-25), np.full(self.nvar, 1.0e-3)
        ipar, rpar = np.zeros(20, np.int32), np.zeros(20, np.float64)
        ipar[1], ipar[3], ipar[2] = 1, 2, max_steps      # IPAR(2) vector tolerances, IPAR(4) Ros3, IPAR(3) step limit
        rpar[2] = 1.0e-3                                 # RPAR(3) starting step
        t0, t1, ierr = C.c_double(tin), C.c_double(tout), C.c_int32(0)
        getattr(self.lib, "rosenbrock_%s_" % s)(_d(y), C.byref(t0), C.byref(t1), _d(atol), _d(rtol), getattr(self.lib, "funtemplate_%s_" % s),
                                                getattr(self.lib, "jactemplate_%s_" % s), _d(rpar), _i(ipar), C.byref(ierr))
        return y, ierr.value, ipar[10:18].copy(), rpar[10], rpar[11]


def read_capture(path):
    """Records written by oracle/capture_wrap.c -> list of dicts."""
    recs = []
    raw = open(path, "rb").read()
    off = 0
    while off < len(raw):
        h = np.frombuffer(raw, np.int32, 14, off)
        off += 56
        assert h[0] == 0x4d495354
        mech, nvar, nfix, nreact, callno = (int(x) for x in h[1:6])
        nd = 2 + nvar + nfix + nreact + nvar + 2
        d = np.frombuffer(raw, np.float64, nd, off)
        off += 8 * nd
        p = 2
        c_in = d[p:p + nvar + nfix]; p += nvar + nfix
        rconst = d[p:p + nreact]; p += nreact
        var_out = d[p:p + nvar]; p += nvar
        recs.append(dict(mech=("gas", "aer", "tot")[mech], callno=callno, stats=h[6:14].copy(), tin=d[0], tout=d[1],
                         var_in=c_in[:nvar].copy(), fix=c_in[nvar:].copy(), rconst=rconst.copy(),
                         var_out=var_out.copy(), tin_out=d[p], stepmin_out=d[p + 1]))
    return recs
