"""TEST INFRASTRUCTURE — numpy restatement of the xkmt part of the reference's fast_k_mt_a / fast_k_mt_t (kpp.f90:2683-2947 |
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
