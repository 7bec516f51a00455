"""TEST INFRASTRUCTURE — numpy restatement of the reference's fast_k_mt_a / fast_k_mt_t (kpp.f90:2683-2947 |
2421-2676) for ONE layer, in the reference's summation order (ia outer, jt inner, one rounding per operation).  Pins the species list
(mistra_amd/mech/<mech>.kmt.json) and the formula on the CPU against layers captured from the running reference model
(tests/golden/kmt_<mech>.npz, tests/test_pack.py); the device kernel is then checked against the same fixtures."""
import json
import os

import numpy as np

MECH_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mistra_amd", "mech")
Z4PI3 = 4.0 * 3.1415926535897932 / 3.0      # z4pi3 = 4._dp * pi / 3._dp


def load(mech):
    return json.load(open(os.path.join(MECH_DIR, mech + ".kmt.json")))


def vterm(a, t, p):
    """str.f90:2793-2863, one rounding per operation, the PARAMETER constants folded left to right as the compiler does; a**3 = (a*a)*a"""
    import math
    g, r0, rhow = 9.80665, 8.3144743 / 28.96546e-3, 1000.0
    b = (-.318657e+1, .992696e+0, -.153193e-2, -.987059e-3, -.578878e-3, +.855176e-4, -.327815e-5)
    c1, c3, c4 = 2.0 * g / 9.0, 1.26 * 6.6e-8 * 101325.0 / 293.15, 32.0 * g / 3.0
    rho_a = p / (r0 * t)
    eta = 3.7957e-06 + 4.9e-08 * t
    if a <= 1.0e-5:
        return c1 * a * a * (rhow - rho_a) / eta * (1.0 + c3 * t / (a * p))
    best = c4 * ((a * a) * a) * (rhow - rho_a) * rho_a / (eta * eta)
    x = math.log(best)
    y = b[6] * x + b[5]
    for k in (4, 3, 2, 1, 0):
        y = y * x + b[k]
    return eta * math.exp(y) / (2.0 * rho_a * a)


def vt_layer(tab, ff, rq, kw, ka, ifeed, nkc_l, cw, t, p, vt):
    """the LWC-weighted sedimentation velocity vt(1:nkc) of one layer (the routine's l = 1 pass, every bin with cw > 0) -> updated copy"""
    out = np.array(vt, np.float64)
    nka, nkt = tab["nka"], tab["nkt"]
    rqm = rq * 1.0e-6
    for kc in range(1, nkc_l + 1):
        ia0, ia1 = ((2 if ifeed == 2 else 1), ka) if kc in (1, 3) else (ka + 1, nka)
        xx1 = 0.0
        for ia in range(ia0, ia1 + 1):
            jt0, jt1 = (1, int(kw[ia - 1])) if kc in (1, 2) else (int(kw[ia - 1]) + 1, nkt)
            for jt in range(jt0, jt1 + 1):
                r = rqm[ia - 1, jt - 1]
                xx1 = xx1 + ((((r * r) * r) * vterm(r, t, p)) * ff[ia - 1, jt - 1]) * 1.0e6
        if cw[kc - 1] > 0.0:
            out[kc - 1] = Z4PI3 / cw[kc - 1] * xx1
    return out


def fast_k_mt_layer(tab, ff, rq, kw, ka, ifeed, nkc_l, cw, cm, freep, alpha, vmean, xkmt):
    """ff, rq: [nka][nkt]; xkmt [nkc][NSPEC] -> updated copy"""
    out = np.array(xkmt, np.float64)
    nka, nkt = tab["nka"], tab["nkt"]
    rqm = rq * 1.0e-6
    for kc in range(1, nkc_l + 1):
        if not cm[kc - 1] > 0.0:
            continue
        ia0, ia1 = ((2 if ifeed == 2 else 1), ka) if kc in (1, 3) else (ka + 1, nka)
        for c in tab["lex"]:
            x1 = 4.0 / (3.0 * alpha[c - 1]) if alpha[c - 1] > 0.0 else 0.0
            xk1 = 0.0
            for ia in range(ia0, ia1 + 1):
                jt0, jt1 = (1, int(kw[ia - 1])) if kc in (1, 2) else (int(kw[ia - 1]) + 1, nkt)
                for jt in range(jt0, jt1 + 1):
                    r = rqm[ia - 1, jt - 1]
                    x2 = vmean[c - 1] / (r / freep + x1)
                    xk1 = xk1 + (((x2 * r) * r) * ff[ia - 1, jt - 1]) * 1.0e6
            if cw[kc - 1] > 0.0:
                out[kc - 1, c - 1] = Z4PI3 / cw[kc - 1] * xk1
    return out
