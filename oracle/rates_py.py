"""TEST INFRASTRUCTURE — plain-Python restatement of the reference's rate laws that the gas mechanism calls
(kpp.f90:7127-7561, 8198-8376) and an evaluator for the postfix programs of mistra_amd/mech/<mech>.rates.json
(tools/extract_rates.py).  Checks the extracted TABLE against the compiled reference on the CPU (tests/test_rates.py); the
device evaluator (mistra_amd/csrc/rates.hip) is then checked against the same fixture on the GPU.  Float32-literal semantics
as in the reference: 300. exact, 8.314 -> float32, 0.21 -> float32, 10**(-6.16) in single precision."""
import math

import numpy as np

R8314 = float(np.float32(8.314))
O21 = float(np.float32(0.21))
TEN_POW = float(np.float32(10.0) ** np.float32(-6.16))


def _troe(e, a1, a2, b1, b2, fc, tref):
    aircc, te = e[0], e[1]
    a0 = (a1 * aircc) * math.pow(te / tref, a2)
    b0 = b1 * math.pow(te / tref, b2)
    l = math.log10(a0 / b0)
    return (a0 / (1.0 + a0 / b0)) * math.pow(fc, 1.0 / (1.0 + l * l))


K5555 = float(np.float32(55.55))
DCLIM = 1.0e10
# positions in the slot list of what the rate laws read from COMMON themselves (tools/extract_rates.py: fslot_names)
FS = dict(H2OL=0, CLM=4, BRM=8, YXKMT_N2O5=12, YXKMT_CLNO3=16, YXKMT_BRNO3=20, YCW=24, YXKMTD_N2O5=28, YXKMTD_BRNO3=30,
          YXKMTD_CLNO3=32, YXKMTD_HNO3=34, YXKMTD_NH3=36, YXKMTD_H2SO4=38, YCWD=40, YHENRY_HNO3=42, YXEQ_HNO3=43, C_HNO3=44,
          C_HNO3L=45, C_NO3ML=47, XHAL=49)


class E(list):
    """env vector that also knows the slot list: e.at("YCWD", 1)"""
    fslot = None

    def at(self, key, k=0):
        return self[self.fslot[FS[key] + k]]

    def has(self, key):
        return self.fslot[FS[key]] >= 0


def _fmax(a, b):
    return a if (a > b or b != b) else b


def _fdhet(e, na, nb):      # fdhetg | fdheta | fdhett (kpp.f90:8198, 8269, 8311)
    na, nb = int(na), int(nb)
    ycwd = e.at("YCWD", na - 1)
    if nb == 1:
        yx = e.at("YXKMTD_HNO3", na - 1)
        x1 = yx * ycwd
        if e.has("C_NO3ML"):
            caq = 0.0
            if (e.at("YXEQ_HNO3") + 1.0e-2) != 0.0:
                caq = ((e.at("C_HNO3L", na - 1) + e.at("C_NO3ML", na - 1)) * 1.0e-2) / (e.at("YXEQ_HNO3") + 1.0e-2)
        else:
            caq = ((e.at("C_HNO3L", na - 1) * 1.5e3) * 1.0e-2) / (e.at("YXEQ_HNO3") + 1.0e-2)
        x2 = 0.0
        if e.at("C_HNO3") != 0.0 and e.at("YHENRY_HNO3") != 0.0:
            x2 = ((-yx) / (e.at("C_HNO3") * e.at("YHENRY_HNO3"))) * caq
        return _fmax(0.0, x1 + x2)
    return e.at({2: "YXKMTD_N2O5", 3: "YXKMTD_NH3", 4: "YXKMTD_H2SO4"}[nb], na - 1) * ycwd


def _fhet_t(e, a0, b0, c0):      # kpp.f90:7582
    a0, b0, c0 = int(a0), int(b0), int(c0)
    h2oa = e.at("H2OL", a0 - 1)
    het = (h2oa + 5.0e2 * e.at("CLM", a0 - 1)) + 3.0e5 * e.at("BRM", a0 - 1)
    xbr = h2oa if b0 == 1 else 5.0e2 if b0 == 2 else 3.0e5
    xtr = e.at({1: "YXKMT_N2O5", 2: "YXKMT_CLNO3", 3: "YXKMT_BRNO3"}[c0], a0 - 1)
    return ((xtr * e.at("YCW", a0 - 1)) * xbr) / het if het > 0.0 else 0.0


def _fhet_d(e, xliq, xhet, a0, b0, c0):      # fhet_da | fhet_dt (kpp.f90:8023, 8111)
    a0, b0, c0 = int(a0), int(b0), int(c0)
    xhal = e.at("XHAL")
    if xhet == 0.0:
        xtr = e.at({1: "YXKMT_N2O5", 2: "YXKMT_CLNO3", 3: "YXKMT_BRNO3"}[c0], a0 - 1)
        h2oa = e.at("H2OL", a0 - 1)
        het = (h2oa + 5.0e2 * e.at("CLM", a0 - 1)) + 3.0e5 * e.at("BRM", a0 - 1)
        yw = e.at("YCW", a0 - 1)
        if xhal == 0.0:
            if c0 in (2, 3):
                xtr = 0.0
            het = e.at("H2OL", a0 - 1)
    else:
        xtr = e.at({1: "YXKMTD_N2O5", 2: "YXKMTD_BRNO3", 3: "YXKMTD_CLNO3"}[c0], a0 - 1)
        h2oa = (K5555 * e.at("YCWD", a0 - 1)) * 1.0e3
        het = (h2oa + 5.0e2 * e.at("CLM", a0 - 1)) + 3.0e5 * e.at("BRM", a0 - 1)
        yw = e.at("YCWD", a0 - 1)
        if xhal == 0.0:
            if c0 in (2, 3):
                xtr = 0.0
            het = (K5555 * e.at("YCWD", a0 - 1)) * 1.0e3
    xbr = h2oa if b0 == 1 else 5.0e2 if b0 == 2 else 3.0e5
    r = ((xtr * yw) * xbr) / het if het > 0.0 else 0.0
    if (c0 in (2, 3) or b0 in (2, 3)) and xhal == 0.0:
        r = 0.0
    if xliq == 0.0:
        r = 0.0
    return r


def _arr2(e, a0, b0):
    return a0 * math.exp(b0 * (1.0 / e[1] - 3.3557e-3))


def _sp23(e, a1, b1, a2, b2, a3, b3):
    aircc, te, h2oppm = e[0], e[1], e[2]
    tte = 1.0 / te
    f1 = a1 * math.exp(b1 * tte)
    f2 = (a2 * aircc) * math.exp(b2 * tte)
    f3 = (((a3 * aircc) * h2oppm) * 1.0e-6) * math.exp(b3 * tte)
    return (f1 + f2) * (1.0 + f3)


def _shno3(e, a1, b1, a2, b2, a3, b3):
    aircc, tte = e[0], 1.0 / e[1]
    f1, f2, f3 = a1 * math.exp(b1 * tte), a2 * math.exp(b2 * tte), a3 * math.exp(b3 * tte)
    return f1 + ((f3 * aircc) / (1.0 + (f3 * aircc) / f2))


def _fbck2(e, a1, a2, b1, b2, fc, ck):
    te = e[1]
    x1 = _troe(e, a1, a2, b1, b2, fc, 300.0)
    return x1 / (((((5.44e-9 * math.exp(14192.0 / te)) * R8314) / 101325.0) * te) / ck) if ck != 0.0 else 0.0


def _dms(e):
    o2, tte = O21 * e[0], 1.0 / e[1]
    return ((9.5e-39 * math.exp(5270.0 * tte)) * o2) / (1.0 + (7.5e-29 * math.exp(5610.0 * tte)) * o2)


def _fcn(e, x1):
    x2 = R8314 * e[1]
    return ((TEN_POW * math.exp(-90.7e3 / x2)) * (e[3] / x2)) * x1


FUNCS = {
    "farr": lambda e, a, b: a * math.exp(b / e[1]),
    "farr_sp": lambda e, a, b, c, d: (a * math.pow(e[1] / b, c)) * math.exp(d / e[1]),
    "atk_3": lambda e, *a: _troe(e, *a, 300.0),
    "atk_3f": lambda e, *a: _troe(e, *a, 298.0),
    "shno3": _shno3,
    "fbck": lambda e, a1, a2, b1, b2, fc, ak, bk: _troe(e, a1, a2, b1, b2, fc, 300.0) / (ak * math.exp(bk / e[1])),
    "fbckj": lambda e, a1, a2, b1, b2, ak, bk: _troe(e, a1, a2, b1, b2, 0.6, 300.0) / (ak * math.exp(bk / e[1])),
    "fbck2": _fbck2,
    "sp_17": lambda e, a, b: a * (1.0 + e[0] / b),
    "sp_23": _sp23,
    "fcn": _fcn,
    "dms_add": _dms,
    "fdhetg": _fdhet, "fdheta": _fdhet, "fdhett": _fdhet,
    "farr2": _arr2,
    "fhet_t": _fhet_t,
    "fhet_da": _fhet_d, "fhet_dt": _fhet_d,
    "fliq_60": lambda e, a1, b1, c, d: ((_arr2(e, a1, b1)) * c) / (c + 0.1 / d) if d > 0.0 else 0.0,
    "dmin2": lambda e, a: a if a < DCLIM else DCLIM,
    "dmin3": lambda e, a: a if a < DCLIM * 2.0 else DCLIM * 2.0,
    "flsc4": lambda e, a, b, c: (a * b) * ((c * c) * c) if c > 0.0 else 0.0,
    "flsc5": lambda e, a, b, c: (a * (b * b)) * (((c * c) * c) * c) if c > 0.0 else 0.0,      # flang expands c**4 as ((c*c)*c)*c
    "flsc6": lambda e, a, b: a / b if b > 1.0e-15 else 0.0,
    "uplim": lambda e, a, b, c, d: a / (1.0 + ((b / DCLIM) * _fmax(c, 0.0)) * d) if d > 0.0 else 0.0,
    "uparm": lambda e, a0, b0, c, d, ee: _arr2(e, a0, b0) / (1.0 + ((c / DCLIM) * d) * ee) if d > 0.0 else 0.0,
    "uplip": lambda e, a, b, c: (a / (1.0 + ((a / DCLIM) * _fmax(b, 0.0)) * c)) * (c * c) if c > 0.0 else 0.0,
    "uparp": lambda e, a0, b0, c, d: (_arr2(e, a0, b0) / (1.0 + ((_arr2(e, a0, b0) / DCLIM) * c) * d)) * (d * d) if d > 0.0 else 0.0,
}


def _a_n2o5(e, cw1, cm1, s13, s14):      # kpp.f90:8377 with kc = 1
    xno3m = xclm = xh2o = 0.0
    if cw1 > 0.0:
        xno3m = (s13 / cw1) * 1.0e-3
        xclm = (s14 / cw1) * 1.0e-3
    if cm1 > 0.0 and cw1 > 0.0:
        xh2o = 55.55 * (cm1 / cw1)
    xk2f = 1.15e6 - 1.15e6 * math.exp(-0.13 * xh2o)
    denom = 1.0
    if xno3m > 0.0:
        denom = (1.0 + (6.0e-2 * xh2o) / xno3m) + (29.0 * xclm) / xno3m
    return (3.2e-8 * xk2f) * (1.0 - (1.0 / denom))


# st_coeff_a / st_coeff_t (kpp.f90:857-1038 | 664-851) use the same program format (tools/extract_stcoeff.py)
FUNCS.update({"exp": lambda e, x: math.exp(x), "a_n2o5": _a_n2o5, "min": lambda e, a, b: a if (a < b or b != b) else b})


def st_coeff_layer(table, lp_joyce14bc, lp_buxmann15alph, env):
    """alpha(:,k) of one layer [NSPEC] from env = [t(k), cw(1,k), cm(1,k), sion1(13,1,k), sion1(14,1,k)]; table = <mech>.stcoeff.json"""
    var = table["variants"][int(bool(lp_joyce14bc)) + 2 * int(bool(lp_buxmann15alph))]
    return evaluate({"nreact": table["nspec"], "programs": var["programs"]}, {n: i for i, n in enumerate(table["env"])}, env)


def evaluate(table, slot, env, fslot=None):
    """rconst[nreact] for one env vector; table = the .rates.json dict, slot = {name: env index}, fslot = the slot list of the
    mechanism (both from mistra_amd/mech/<mech>.rates_env.json)"""
    env = E(float(x) for x in env)
    env.fslot = fslot
    out = np.empty(table["nreact"])
    for r, prog in enumerate(table["programs"]):
        st = []
        for t in prog:
            k = t[0]
            if k == "num":
                st.append(float(t[1]))
            elif k == "var":
                st.append(float(env[slot[t[1]]]))
            elif k == "arr":
                st.append(float(env[slot["%s(%s)" % (t[1], ",".join(str(i) for i in t[2:]))]]))
            elif k == "neg":
                st[-1] = -st[-1]
            elif k == "call":
                n = t[2]
                args = st[len(st) - n:] if n else []
                del st[len(st) - n:]
                st.append(FUNCS[t[1]](env, *args))
            else:
                b, a = st.pop(), st.pop()
                st.append(a + b if k == "+" else a - b if k == "-" else a * b if k == "*" else a / b)
        assert len(st) == 1
        out[r] = st[0]
    return out
