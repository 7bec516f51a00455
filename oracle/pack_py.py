"""TEST INFRASTRUCTURE — numpy restatement of the hand-over halves of the reference's per-layer drivers (gas_drive / aer_drive /
tot_drive: gas.f:60-217 | aer.f:59-246 | tot.f:59-982 with aer_mk.dat / aer_km.dat; budgets bud_x.f, bud_s_x.f) from the tables
tools/extract_pack.py writes (mistra_amd/mech/<mech>.pack.json).  Pins the TABLES on the CPU against driver calls captured from the
running reference model (tests/golden/drive_<mech>.npz, tests/test_pack.py); the device kernels (mistra_amd/csrc/pack.hip) are then
checked against the same fixtures on the GPU.  One layer at a time; arrays as the fixtures hold them (sl1 / sion1: the layer's
Fortran slab (j, nkc) flattened column-major)."""
import json
import os

import numpy as np

MECH_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mistra_amd", "mech")
O21, O79, K5555 = float(np.float32(0.21)), float(np.float32(0.79)), float(np.float32(55.55))      # default-REAL literals


def fmax0(v):
    """MAX(0.d0, x) as flang compiles it (0 > x ? 0 : x): -0.0 and NaN pass through"""
    return np.where(0.0 > np.asarray(v, np.float64), 0.0, v)


def load(mech):
    return json.load(open(os.path.join(MECH_DIR, mech + ".pack.json")))


def flat(tab, arr, i, kc):
    return (i - 1) + (kc - 1) * (tab["j2"] if arr == "sl1" else tab["j6"])


def pack(tab, c_prev, s1, s3, sl1, sion1, air, h2o, cvv, gas_m2k, rad_m2k):
    """-> (C, sl1, sion1): x_drive up to Update_RCONST_x.  c_prev: what COMMON /GDATA_x/ holds when the driver is entered (entries the
    driver does not set keep it)."""
    C, L, I = np.array(c_prev, np.float64), np.array(sl1, np.float64), np.array(sion1, np.float64)
    if tab["preclamp"]:
        L, I = fmax0(L), fmax0(I)
    for c, src in gas_m2k:
        C[c - 1] = s1[src - 1]
    for c, src in rad_m2k:
        C[c - 1] = s3[src - 1]
    for c, kind, kc in tab["fix"]:
        C[c - 1] = O21 * air if kind == "O2" else O79 * air if kind == "N2" else h2o if kind == "H2O" else (K5555 / cvv[kc - 1] if cvv[kc - 1] > 0 else 0.0)
    src = {"sl1": L, "sion1": I}
    for c, arr, i, kc, clamp in tab["pack"]:
        v = src[arr][flat(tab, arr, i, kc)]
        C[c - 1] = fmax0(v) if clamp else v
    return C, L, I


def unpack(tab, C, s1, s3, sl1, sion1, gas_k2m, rad_k2m):
    """the hand-over after the integration -> (s1, s3, sl1, sion1)"""
    s1, s3, L, I = (np.array(x, np.float64) for x in (s1, s3, sl1, sion1))
    for j, c in enumerate(gas_k2m):
        s1[j] = C[c - 1]
    for j, c in enumerate(rad_k2m):
        s3[j] = C[c - 1]
    dst = {"sl1": L, "sion1": I}
    for arr, i, kc, c, clamp in tab["unpack"]:
        dst[arr][flat(tab, arr, i, kc)] = fmax0(C[c - 1]) if clamp else C[c - 1]
    return s1, s3, L, I


def budgets(tab, mech_tables, C, rconst, dt, bg, bgs):
    """bud_x (every reaction: RCONST(i) * reactants, left to right) and bud_s_x on the state the integration left -> (bg, bgs), both [n][2]"""
    t = mech_tables
    X = np.concatenate([C, t.consts])
    bg, bgs = np.array(bg, np.float64).reshape(-1, 2), np.array(bgs, np.float64).reshape(-1, 2)
    for r in range(t.nreact):
        p = rconst[r]
        for f in t.a_fac[t.a_ptr[r]:t.a_ptr[r + 1]]:
            p = p * X[f]
        bg[r, 0] = p
        bg[r, 1] = bg[r, 1] + dt * p
    for slot, terms in tab["bud_s"]:
        acc = None
        for sign, r, cs in terms:
            p = rconst[r - 1]
            for c in cs:
                p = p * C[c - 1]
            acc = (-p if sign < 0 else p) if acc is None else (acc - p if sign < 0 else acc + p)
        bgs[slot - 1, 0] = acc
    for lo, hi in tab["bud_s_acc"]:
        for i in range(lo - 1, hi):
            bgs[i, 1] = bgs[i, 1] + dt * bgs[i, 0]
    return bg, bgs
