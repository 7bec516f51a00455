"""TEST INFRASTRUCTURE — plain-Python restatement of the reference's henry_a / henry_t, equil_co_a / equil_co_t and v_mean_a / v_mean_t
(kpp.f90:1676-2145 | 2954-3363 | 1268-1670) for ONE layer from the tables tools/extract_liq.py cuts out of them (mistra_amd/mech/<mech>.liq.json): the reference's
operation order, one rounding per operation, the host libm's exp.  Pins tables and formulas on the CPU against layers captured from the
running reference model (tests/golden/liq_<mech>.npz, tests/test_pack.py); the device kernels are then checked against the same fixtures."""
import json
import math
import os

import numpy as np

MECH_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mistra_amd", "mech")


def load(mech):
    return json.load(open(os.path.join(MECH_DIR, mech + ".liq.json")))


def load_vmean(mech):
    return json.load(open(os.path.join(MECH_DIR, mech + ".vmean.json")))


def v_mean_layer(tab, tt):
    """kpp.f90:1319-1463 (v_mean_t) | 1524-1668 (v_mean_a): vmean(:,k) of one layer, [NSPEC], from the table tools/extract_vmean.py writes"""
    out = np.zeros(tab["nspec"])                         # vmean(:,:) = 0._dp
    for j, a in tab["entries"]:
        out[j - 1] = math.sqrt(tt / a) * tab["coef"]     # func(a,k) = sqrt(tt(k)/a)*4.60138
    return out


def henry_layer(tab, tt):
    """kpp.f90:1742-1905: henry(:,k) of one layer, [NSPEC]"""
    t = tab["henry"]
    out = np.zeros(tab["nspec"])                         # henry(:,:) = 0._dp
    tfact = 1.0 / tt - t["tref"]                         # Tfact = 1.d0/tt(k) - 3.3540d-3
    for j, a0, b0 in t["entries"]:
        out[j - 1] = a0 if b0 is None else a0 * math.exp(b0 * tfact)      # func3(a0,b0) = a0*exp(b0*Tfact)
    fct = t["fct"] * tt                                  # FCT = 0.0820577_dp * tt(k)
    for j in range(tab["nspec"]):
        if out[j] > 0.0:
            out[j] = 1.0 / (out[j] * fct)
    return out


def _product(prog, tt, tref, cv2, xg):
    v = None
    for f in prog:
        if f[0] == "num":
            x = f[1]
        elif f[0] == "funa":                             # funa(a0,b0,k) = a0*exp(b0*(1/tt(k)-3.354d-3))
            x = f[1] * math.exp(f[2] * (1.0 / tt - tref))
        elif f[0] == "cv2":
            x = cv2
        else:
            x = xg[f[1] - 1]                             # xgamma(i,kc,k)
        v = x if v is None else v * x
    return v


def equil_co_layer(tab, tt, conv2, xgamma, xkef, xkeb):
    """kpp.f90:3032-3149: xkef(:,:,k), xkeb(:,:,k) of one layer, [nkc][NSPEC] each, from what they held before (species the
    routine does not set keep their values; a bin with conv2 <= 0 is zeroed); conv2 [nkc], xgamma [nkc][j6]"""
    t = tab["equil"]
    ef, eb = np.array(xkef, np.float64), np.array(xkeb, np.float64)
    for kc in range(t["nkc"]):
        cv2 = conv2[kc]
        if cv2 > 0.0:
            for j, fprog, bprog in t["entries"]:
                ef[kc, j - 1] = _product(fprog, tt, t["tref"], cv2, xgamma[kc])
                eb[kc, j - 1] = _product(bprog, tt, t["tref"], cv2, xgamma[kc])
        else:
            ef[kc, :] = 0.0
            eb[kc, :] = 0.0
    return ef, eb
