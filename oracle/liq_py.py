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


def cw_rc_layer(ff, rq, e, kw, ka, ifeed, feu=None, cloud=None, crys4=None, dry=False):
    """cw_rc (kpp.f90:2152-2414) | dry_cw_rc (kpp.f90:4580-4690) for ONE layer: ff [nka][nkt] = ff(1:nkt,1:nka,k) transposed as the model holds it,
    rq [nka][nkt], e [nkt], kw [nka] (1-based jt limits), ka.  -> rc, cw, cm, conv2 [4] and the 'below both crystallisation points' flag; dry: rcd, cwd [2].
    The reference's loops and running sums, one rounding per operation."""
    nka, nkt = ff.shape
    pi = 3.1415926535897932
    xpi = float(np.float32(4.0) / np.float32(3.0)) * pi if dry else 4.0 / 3.0 * pi       # dry_cw_rc: 4./3.*pi, a single-precision quotient
    ial = 2 if ifeed == 2 else 1
    nb = 2 if dry else 4
    cws, rcs, cms = [0.0] * 4, [0.0] * 4, [0.0] * 4
    for kc in range(nb):
        small = kc in (0, 2)
        for ia in range(ial if small else ka + 1, (ka if small else nka) + 1):
            kwa = int(kw[ia - 1])
            for jt in (range(1, kwa + 1) if kc < 2 else range(kwa + 1, nkt + 1)):
                f, r = float(ff[ia - 1, jt - 1]), float(rq[ia - 1, jt - 1])
                x0 = (f * xpi) * ((r * r) * r)
                cws[kc] = cws[kc] + x0
                rcs[kc] = rcs[kc] + x0 * r
                if not dry:
                    cms[kc] = cms[kc] + f * float(e[jt - 1])
    rc = np.array([(rcs[kc] / cws[kc]) * 1.0e-6 if cws[kc] > 0.0 else 0.0 for kc in range(nb)])
    cw = np.array([cws[kc] * 1.0e-12 for kc in range(nb)])
    if dry:
        return rc, cw
    cwm, cwmd = 1.0e-1, 1.0e2
    xcryssulf, xcrysss, xdelisulf, xdeliss = (float(x) for x in crys4)
    cm, conv2 = np.zeros(4), np.zeros(4)
    below = feu < min(xcryssulf, xcrysss)
    if not below:
        on = [cws[0] >= cwm and ((bool(cloud[0]) and feu >= xcryssulf) or feu >= xdelisulf),
              cws[1] >= cwm and ((bool(cloud[1]) and feu >= xcrysss) or feu >= xdeliss), cws[2] >= cwmd, cws[3] >= cwmd]
        for kc in range(4):
            if on[kc]:
                cm[kc] = cms[kc] * 1.0e-3
                conv2[kc] = 1.0e9 / cws[kc]
    return rc, cw, cm, conv2, int(below)


def dry_rates_layer(tt, freep, rcd, vmean4=None, henry4=None):
    """dry_rates_g (vmean4 None: kpp.f90:4697-4853) | dry_rates_a / dry_rates_t (kpp.f90:4860-5073 | 5079-5198) for ONE layer and the four species of the
    routines' idr list (HNO3, N2O5, NH3, H2SO4): -> xkmtd [2][4], xeq(HNO3) (, henry [4] for gas, from the entries it held before)."""
    zgamma, mass = (0.02, 0.02, 0.05, 0.1), (6.3e-2, 1.08e-1, 1.7e-2, 9.8e-2)
    xeq = 1.54e+1 * math.exp(8700.0 * (1.0 / tt - 3.354e-3))                     # funa(1.54d+1,8700.d0,k)
    h = None
    if vmean4 is None:
        h = [float(x) for x in henry4]
        h[0] = (2.5e6 / xeq) * math.exp(8694.0 * ((1.0 / tt) - 3.3557e-3))          # func3(2.5d6/xeq,8694.d0,k)
        fct = float(np.float32(0.0820577)) * tt                                      # FCT=0.0820577*tt(k): a default-real literal
        h = [1.0 / (x * fct) if x > 0.0 else x for x in h]
        vm = [math.sqrt(tt / m) * float(np.float32(4.60138)) for m in mass]          # func(a,k) = sqrt(tt(k)/a)*4.60138
    else:
        vm = [float(x) for x in vmean4]
    xk = np.zeros((2, 4))
    for kc in range(2):
        r = float(rcd[kc])
        for l in range(4):
            x1 = 1.0 / (r * (r / freep + 4.0 / (3.0 * zgamma[l]))) if (zgamma[l] > 0.0 and r > 0.0) else 0.0
            xk[kc, l] = vm[l] * x1
    return (xk, xeq, np.array(h)) if h is not None else (xk, xeq)
