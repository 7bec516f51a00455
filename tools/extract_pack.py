#!/usr/bin/env python3
"""Extracts the hand-over tables of the reference's per-layer mechanism drivers into data:
mistra_amd/mech/<mech>.pack.json — what gas_drive / aer_drive / tot_drive (gas.f:60-217 | aer.f:59-246 | tot.f:59-982, with the
include files aer_mk.dat / aer_km.dat) do around Update_RCONST_x + INTEGRATE_x, as tables:

  "pack"    [[c, "sl1"|"sion1", i, kc, clamp], ...]   C(c) = [max(0,] arr(i,kc,k) [)]      (the explicit assignments before the
            integration; c 1-based index into C = VAR | FIX, i and kc 1-based as in the Fortran)
  "fix"     [[c, kind, kc], ...]   kind "O2": 0.21*air, "N2": 0.79*air, "H2O": h2o, "H2Ol": 55.55/cvv<kc> if cvv<kc> > 0 else 0
            (0.21, 0.79, 55.55 are DEFAULT-REAL literals: the double nearest the float32)
  "preclamp"  true where the driver first clamps the layer's whole sl1(:,:,k) and sion1(:,:,k) to >= 0 IN PLACE (aer, tot)
  "unpack"  [["sl1"|"sion1", i, kc, c, clamp], ...]   arr(i,kc,k) = [max(0,] C(c) [)]      (after the integration)
  "gas_maps"  true: s1 / s3 travel through the run-time index maps gas_m2k_x / rad_m2k_x (pack) and gas_k2m_x / rad_k2m_x (unpack)
            of module gas_common (utils.f90:82-140): they depend on the user's species lists and are handed to the library at run time
  "bud_s"   [[slot, [[sign, reaction, [c, ...]], ...]], ...]   bgs(1,slot,k) = +-RCONST(reaction)*C(c)*... +- ...  left to right
            (bud_s_g.f | bud_s_a.f | bud_s_t.f);
            "bud_s_acc": the slot ranges whose bgs(2,.,k) accumulate dt * bgs(1,.,k)
  "bud"     "A": bg(1,i,kl) = the rate product A(i) of Fun_x for every reaction i (bud_g.f | bud_a.f | bud_t.f state the same factor lists:
            checked here against mistra_amd/mech/<mech>.mech), bg(2,i,kl) += dt * bg(1,i,kl)

Needs the reference tree (/root/reference/src); the output is committed.  Run: python tools/extract_pack.py [mech ...]"""
import json
import os
import re
import sys

REF = os.environ.get("MISTRA_REFERENCE_SRC", "/root/reference/src")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "mistra_amd", "mech")
sys.path.insert(0, os.path.join(HERE, ".."))
FILES = {"gas": ("gas.f", "g", "gas_drive"), "aer": ("aer.f", "a", "aer_drive"), "tot": ("tot.f", "t", "tot_drive")}


def parameters(mech):
    p = {}
    for m in re.finditer(r"PARAMETER\s*\(\s*(\w+)\s*=\s*(\d+)\s*\)", open(os.path.join(REF, mech + "_Parameters.h"), errors="replace").read()):
        p[m.group(1).lower()] = int(m.group(2))
    return p


def code_lines(path):
    """(line number, text) of the non-comment lines of a fixed-form file"""
    out = []
    for n, l in enumerate(open(path, errors="replace").read().split("\n"), 1):
        if not l.strip() or l[0] in "cC*!" or l.lstrip().startswith("!"):
            continue
        out.append((n, l.split("!")[0].rstrip()))
    return out


def statements(path):
    """code_lines with fixed-form continuation lines (a character in column 6) joined to their statement"""
    out = []
    for n, l in code_lines(path):
        if len(l) > 5 and l[:5].strip() == "" and l[5] not in " 0" and out:
            out[-1] = (out[-1][0], out[-1][1] + l[6:].strip())
        else:
            out.append((n, l))
    return out


def driver_body(mech):
    """statements of x_drive with the include files spliced in, split at the INTEGRATE_x call"""
    fname, sfx, sub = FILES[mech]
    lines = code_lines(os.path.join(REF, fname))
    start = next(i for i, (_, l) in enumerate(lines) if re.match(r"\s+subroutine\s+%s\b" % sub, l, re.I))
    end = next(i for i, (_, l) in enumerate(lines) if i > start and re.match(r"\s+end subroutine\s+%s\b" % sub, l, re.I))
    body = []
    for n, l in lines[start:end]:
        m = re.match(r"\s+include\s+'(aer_mk\.dat|aer_km\.dat)'", l, re.I)
        if m:
            body.extend(("%s:%d" % (m.group(1), k), t) for k, t in code_lines(os.path.join(REF, m.group(1))))
        else:
            body.append(("%s:%d" % (fname, n), l))
    cut = next(i for i, (_, l) in enumerate(body) if re.search(r"call\s+INTEGRATE_%s" % sfx, l, re.I))
    return body[:cut], body[cut + 1:]


def extract(mech):
    par = parameters(mech)
    nvar = par["nvar"]

    def cidx(name):
        name = name.lower()
        if name.startswith("indf_"):
            return nvar + par[name]
        return par[name]

    before, after = driver_body(mech)
    pack, fix, unpack = [], [], []
    j2_model, _, _ = model_dims()
    j3_model = int(re.search(r"integer,\s*parameter\s*::\s*j3\s*=\s*(\d+)", open(os.path.join(REF, "global_params.f90"), errors="replace").read()).group(1))
    IDX = r"(\d+|j2-j3\+\d+)"      # first index of sl1: a literal, or j2-j3+N (the radicals' liquid-phase counterparts behind the j1_fake gases)

    def index(txt):
        return int(txt) if txt.isdigit() else j2_model - j3_model + int(txt.split("+")[1])

    preclamp = any(re.search(r"sl1\(:,:,k\)\s*=\s*max\(0\.d0,\s*sl1\(:,:,k\)\)", l) for _, l in before)
    assert preclamp == any(re.search(r"sion1\(:,:,k\)\s*=\s*max\(0\.d0,\s*sion1\(:,:,k\)\)", l) for _, l in before)
    gas_maps = any("gas_m2k_" in l for _, l in before) and any("gas_k2m_" in l for _, l in after)
    assert gas_maps
    pending_if = None
    for where, l in before:
        m = re.match(r"\s+C\((ind_\w+)\)\s*=\s*(max\(0\.d0,\s*)?(sl1|sion1)\(" + IDX + r",(\d+),k\)\)?\s*$", l.replace(" ", "").replace("C(", " C(", 1))
        if m:
            pack.append([cidx(m.group(1)), m.group(3), index(m.group(4)), int(m.group(5)), bool(m.group(2))])
            continue
        if re.search(r"\b(sl1|sion1)\(", l) and not re.search(r"(sl1|sion1)\(:,:,k\)|common|double precision", l):
            raise ValueError("%s: unrecognised statement with sl1 / sion1: %r" % (where, l))
        m = re.match(r"\s+FIX\((indf_\w+)\)\s*=\s*(.+?)\s*$", l)
        if m:
            name, rhs = m.group(1), m.group(2).replace(" ", "")
            if rhs == "0.21*air":
                fix.append([cidx(name), "O2", 0])
            elif rhs == "0.79*air":
                fix.append([cidx(name), "N2", 0])
            elif rhs == "h2o":
                fix.append([cidx(name), "H2O", 0])
            elif re.fullmatch(r"55\.55/cvv(\d)", rhs):
                kc = int(rhs[-1])
                assert pending_if == kc, (where, l)
                fix.append([cidx(name), "H2Ol", kc])
            elif rhs == "0.":
                assert fix and fix[-1][0] == cidx(name) and fix[-1][1] == "H2Ol", (where, l)      # the else branch of the same test
            else:
                raise ValueError("%s: unknown FIX expression %r" % (where, l))
            continue
        m = re.match(r"\s+if\s*\(cvv(\d)\.gt\.0\)\s*then", l)
        if m:
            pending_if = int(m.group(1))
    for where, l in after:
        m = re.match(r"\s+(sl1|sion1)\(" + IDX + r",(\d+),k\)\s*=\s*(max\(0\.d0,\s*)?C\((ind_\w+)\)\)?\s*$", " " + l.replace(" ", ""))
        if m:
            unpack.append([m.group(1), index(m.group(2)), int(m.group(3)), cidx(m.group(5)), bool(m.group(4))])
        elif re.search(r"\b(sl1|sion1)\(", l):
            raise ValueError("%s: unrecognised statement with sl1 / sion1: %r" % (where, l))
    # every C index is written at most once by the explicit assignments, every array element at most once by the hand-over
    assert len({p[0] for p in pack}) == len(pack), "a C entry is packed twice"
    assert len({(u[0], u[1], u[2]) for u in unpack}) == len(unpack)
    # ---- budgets
    sfx = FILES[mech][1]
    buds, acc = [], []
    for where, l in statements(os.path.join(REF, "bud_s_%s.f" % sfx)):
        m = re.match(r"\s+bgs\(1,\s*(\d+),k\)\s*=\s*(.+?)\s*$", l)
        if m:
            terms, rhs = [], m.group(2).replace(" ", "")
            for tm in re.finditer(r"([+-]?)RCONST\((\d+)\)((?:\*C\(ind_\w+\))*)", rhs):
                terms.append([-1 if tm.group(1) == "-" else 1, int(tm.group(2)), [cidx(x) for x in re.findall(r"C\((ind_\w+)\)", tm.group(3))]])
            assert terms and "".join(("-" if t[0] < 0 else "+" if i else "") + "RCONST(%d)" % t[1] + "".join("*C(x)" for _ in t[2]) for i, t in enumerate(terms)) == \
                re.sub(r"C\(ind_\w+\)", "C(x)", rhs), (where, l)
            buds.append([int(m.group(1)), terms])
        m = re.match(r"\s+do\s+i\s*=\s*(\d+)(\+0)?\s*,\s*(\d+)", l)
        if m:
            acc.append([int(m.group(1)), int(m.group(3))])
    assert buds and acc
    # bud_x: every reaction, the factor list of Fun_x's A(i)
    from mistra_amd.mechtab import load
    t = load(mech)
    text = " ".join(l.strip() for _, l in code_lines(os.path.join(REF, "bud_%s.f" % sfx)))
    text = re.sub(r"\s*&\s*", "", text)
    seen = 0
    for m in re.finditer(r"bg\(1,(\d+),kl\)\s*=\s*RCONST\((\d+)\)((?:\*(?:VAR|FIX)\(\d+\))*)", text):
        i, r = int(m.group(1)), int(m.group(2))
        assert i == r
        fac = [int(x[1]) - 1 + (t.nvar if x[0] == "FIX" else 0) for x in re.findall(r"(VAR|FIX)\((\d+)\)", m.group(3))]
        want = [int(f) for f in t.a_fac[t.a_ptr[i - 1]:t.a_ptr[i]]]
        assert fac == want, ("bud_%s reaction %d: factors differ from Fun_x's A" % (sfx, i), fac, want)
        seen += 1
    assert seen == t.nreact, (seen, t.nreact)
    return {"mech": mech, "nvar": nvar, "nfix": par["nfix"], "preclamp": preclamp, "gas_maps": True, "pack": pack, "fix": fix,
            "unpack": unpack, "bud": "A", "bud_s": buds, "bud_s_acc": acc,
            "source": "%s (%s) with aer_mk.dat / aer_km.dat, bud_%s.f, bud_s_%s.f" % (FILES[mech][2], FILES[mech][0], sfx, sfx)}


def model_dims():
    """j2, j6, nkc of global_params.f90 (compile-time dimensions of sl1(j2,nkc,n), sion1(j6,nkc,n))"""
    text = open(os.path.join(REF, "global_params.f90"), errors="replace").read()
    val = {}
    for name in ("j1_fake", "j3", "j6", "nkc"):
        val[name] = int(re.search(r"integer,\s*parameter\s*::\s*%s\s*=\s*(\d+)" % name, text).group(1))
    assert re.search(r"integer,\s*parameter\s*::\s*j2\s*=\s*j1_fake\s*\+\s*j3", text)
    return val["j1_fake"] + val["j3"], val["j6"], val["nkc"]


def write_binary(tab, path):
    """<mech>.pack for the library (mistra_amd/csrc/pack.cpp: PackTable::load): int32 header
    {'KPAK', 1, nvar, nfix, j2, j6, nkc, preclamp, n_pack, n_fix, n_unpack, n_slots, n_terms, n_term_words, n_acc, n_envc}, then int32 arrays
    pack[n][4] = {c0, array (0 sl1 | 1 sion1), flat index into the layer's (j, nkc) slab, clamp}; fix[n][3] = {c0, kind (0 O2 | 1 N2 | 2 H2O |
    3 H2Ol), kc0}; unpack[n][4] = {array, flat index, c0, clamp}; slot_id[n_slots]; slot_first_term[n_slots + 1]; term[n_terms][3] = {sign,
    reaction0, first word}; term_words (C indices, 0-based; a term's run ends where the next one's begins); acc[n_acc][2] (1-based slot ranges);
    envc[n_envc][2] = {env slot, c0}: the entries of the rate evaluator's input vector that are concentrations (mistra_amd/mech/<mech>.rates_env.json)"""
    import struct
    import numpy as np
    j2, j6, nkc = tab["j2"], tab["j6"], tab["nkc"]
    dim = {"sl1": j2, "sion1": j6}
    arr = {"sl1": 0, "sion1": 1}
    pack = [[c - 1, arr[a], (i - 1) + (kc - 1) * dim[a], int(cl)] for c, a, i, kc, cl in tab["pack"]]
    kinds = {"O2": 0, "N2": 1, "H2O": 2, "H2Ol": 3}
    fix = [[c - 1, kinds[k], max(kc - 1, 0)] for c, k, kc in tab["fix"]]
    unpack = [[arr[a], (i - 1) + (kc - 1) * dim[a], c - 1, int(cl)] for a, i, kc, c, cl in tab["unpack"]]
    slot_id, first, terms, words = [], [0], [], []
    for slot, tl in tab["bud_s"]:
        slot_id.append(slot)
        for sign, r, cs in tl:
            terms.append([sign, r - 1, len(words)])
            words.extend(c - 1 for c in cs)
        first.append(len(terms))
    envn = json.load(open(os.path.join(OUT, tab["mech"] + ".rates_env.json")))["env"]
    envc = []
    for e, nm in enumerate(envn):
        m = re.match(r"(c|fix)\((\d+)\)$", nm)
        if m:
            envc.append([e, int(m.group(2)) - 1 + (tab["nvar"] if m.group(1) == "fix" else 0)])
    with open(path, "wb") as f:
        f.write(struct.pack("<16i", 0x4B41504B, 1, tab["nvar"], tab["nfix"], j2, j6, nkc, int(tab["preclamp"]), len(pack), len(fix), len(unpack),
                            len(slot_id), len(terms), len(words), len(tab["bud_s_acc"]), len(envc)))
        for a in (pack, fix, unpack, slot_id, first, terms, words, tab["bud_s_acc"], envc):
            f.write(np.asarray(a, np.int32).tobytes())


def main():
    j2, j6, nkc = model_dims()
    for mech in (sys.argv[1:] or ["gas", "aer", "tot"]):
        tab = extract(mech)
        tab.update(j2=j2, j6=j6, nkc=nkc, nbgs=122)
        path = os.path.join(OUT, mech + ".pack.json")
        json.dump(tab, open(path, "w"), separators=(",", ":"))
        write_binary(tab, os.path.join(OUT, mech + ".pack"))
        print("%s: %d packed, %d FIX, %d handed over, preclamp %s, bud_s %d slots -> %s" % (mech, len(tab["pack"]), len(tab["fix"]),
              len(tab["unpack"]), tab["preclamp"], len(tab["bud_s"]), os.path.normpath(path)))


if __name__ == "__main__":
    main()
