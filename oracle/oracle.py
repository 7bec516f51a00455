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
        _lib = lib
    return _lib


def set_variant(v):
    """0 = pinned restatement; bit 0 descending backward sweep; bit 1 fma-contracted; bit 2 reciprocal instead of
    division by pivots (sensitivity studies only)"""
    _oracle_lib().kpp_set_variant(int(v))


def set_options(rtol=0.0, atol=0.0, hstart=0.0):
    """Study knobs of the oracle (0 = INTEGRATE_x's fixed value): tolerances and first step size (tools/hstart_study.py)."""
    _oracle_lib().kpp_set_options(float(rtol), float(atol), float(hstart))


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

    def integrate_batch(self, var, fix, rconst, tin=0.0, tout=10.0):
        """cell-major arrays [ncell, n*] -> (var_out, ierr[ncell], stats[ncell,8])"""
        v = np.array(var, np.float64, order="C")
        ncell = v.shape[0]
        ierr = np.zeros(ncell, np.int32)
        st = np.zeros((ncell, 8), np.int32)
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

    def update_rconst_gas(self, env):
        """Update_RCONST_g (gas.f:275) on one cell's inputs, given as the 74-double vector of tools/extract_rates.py ENV["gas"]:
        fills COMMON /cb_1/ (kpp.f90:7140), /kpp_rate_g/ and /ph_r_g/ (gas_Global.h:76-96) and C, calls the compiled
        reference routine, returns RCONST(331)."""
        assert self.mech == "gas"
        e = np.asarray(env, np.float64)
        nspec = self.nvar + self.nfix
        cb1 = (C.c_double * 4).in_dll(self.lib, "cb_1_")
        rate = (C.c_double * (2 * nspec + nspec + nspec + 2 + 5)).in_dll(self.lib, "kpp_rate_g_")      # yxkmtd(NSPEC,2) yhenry yxeq ycwd(2) conv1 xhal xiod xhet1 xhet2
        ph = (C.c_double * 47).in_dll(self.lib, "ph_r_g_")
        cb1[:] = list(e[0:4])
        r = np.ctypeslib.as_array(rate)
        r[:] = 0.0
        # species numbers of gas_Parameters.h:79-203 (1-based)
        ind = {"ind_hno3l1": 13, "ind_hno3l2": 16, "ind_h2so4": 19, "ind_nh3": 24, "ind_n2o5": 32, "ind_hno3": 75}
        for k, sp in enumerate(("hno3", "n2o5", "nh3", "h2so4")):
            for b in range(2):
                r[b * nspec + ind["ind_" + sp] - 1] = e[61 + 2 * k + b]                 # yxkmtd(sp, bin), column-major
        r[2 * nspec + ind["ind_hno3"] - 1] = e[69]                                      # yhenry
        r[3 * nspec + ind["ind_hno3"] - 1] = e[70]                                      # yxeq
        r[4 * nspec:4 * nspec + 2] = e[9:11]                                            # ycwd
        r[4 * nspec + 2:4 * nspec + 7] = e[4:9]                                         # conv1 xhal xiod xhet1 xhet2
        ph[:] = list(e[11:58])
        c = np.ctypeslib.as_array(self.gdata.c)
        c[:] = 0.0
        c[self.nvar:] = e[58:61]                                                        # FIX
        c[ind["ind_hno3"] - 1], c[ind["ind_hno3l1"] - 1], c[ind["ind_hno3l2"] - 1] = e[71], e[72], e[73]
        self.lib.update_rconst_g_()
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
