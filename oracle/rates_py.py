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


def _fdhetg(e, na, nb):
    na, nb = int(na), int(nb)
    ycwd = e[9 + na - 1]
    if nb == 1:
        yx = e[61 + na - 1]
        x1 = yx * ycwd
        caq = ((e[72 + na - 1] * 1.5e3) * 1.0e-2) / (e[70] + 1.0e-2)
        x2 = 0.0
        if e[71] != 0.0 and e[69] != 0.0:
            x2 = ((-yx) / (e[71] * e[69])) * caq
        s = x1 + x2
        return 0.0 if (0.0 > s or s != s) else s
    return e[61 + 2 * (nb - 1) + na - 1] * ycwd


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
    "fdhetg": _fdhetg,
}


def evaluate(table, slot, env):
    """rconst[nreact] for one env vector; table = the .rates.json dict, slot = {name: env index} (tools/extract_rates.py ENV)"""
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
