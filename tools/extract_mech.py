#!/usr/bin/env python3
"""Build-time table extractor: KPP-generated mechanism file (gas.f | aer.f | tot.f) -> compact mechanism tables.

Runs only where the reference tree is present (this container); the tables it writes
(`mistra_amd/mech/{gas,aer,tot}.mech`) are committed *data* — integers and stoichiometric coefficients — that
the C oracle, the HIP library and the tests all load.  Nothing of the reference's code is emitted.

What is read from `<mech>.f` (all citations for gas.f; aer.f/tot.f have the same generated sections):
  * `Fun_x`       (gas.f:2043)  `A(i) = RCT(i)*<factors>`  and  `Vdot(j) = <signed sum of [coef*]A(i)>`
  * `Jac_SP_x`    (gas.f:2656)  `B(m) = RCT(i)*<factors>`  and  `JVS(k) = <signed sum of [coef*]B(m)>` | `JVS(k) = 0`
  * `BLOCK DATA JACOBIAN_SPARSE_DATA_x` (gas.f:6718)  LU_ICOL / LU_CROW / LU_DIAG  (1-based CSR incl. fill-in)
  * `<mech>_Parameters.h:28-49`  NVAR NFIX NREACT LU_NONZERO

Literal semantics: the reference is built without -r8, so a literal such as `0.05` is a default-REAL
(float32) constant promoted to double; integer literals (`2*`) are exact.  Coefficients are therefore stored as
float64(float32(literal)).

File format (little-endian), see `mistra_amd/mechtab.py` / `mistra_amd/csrc/mech_tables.h` for the loaders:
  int32  magic 'KMCH', version, nvar, nfix, nreact, nnz, nA_fac, nB, nB_fac, nvd_terms, njv_terms, nconst
  then arrays in the order written by `write_mech` below.
"""
import re
import sys
import os
import numpy as np

MAGIC = 0x48434D4B  # 'KMCH'
VERSION = 2


def logical_lines(path):
    """Fixed-form Fortran -> list of (first_line_no, statement) with continuations joined and comments dropped."""
    out = []
    cur, cur_no = None, 0
    with open(path, "r", errors="replace") as f:
        for no, raw in enumerate(f, 1):
            line = raw.rstrip("\n")
            if not line.strip():
                continue
            if line[0] in "Cc*!" or line.lstrip().startswith("!"):
                continue
            if len(line) > 5 and line[:5].strip() == "" and line[5] not in " 0":
                cur += line[6:].strip()           # continuation
                continue
            if cur is not None:
                out.append((cur_no, cur))
            cur, cur_no = line.strip(), no
    if cur is not None:
        out.append((cur_no, cur))
    return out


def f32lit(tok):
    """Value of a Fortran numeric literal as the reference's compilers see it (default REAL -> float32)."""
    t = tok.lower()
    if "d" in t:
        return float(t.replace("d", "e"))
    if re.fullmatch(r"\d+", t):
        return float(int(t))
    return float(np.float32(float(t)))


_term = re.compile(r"([+-]?)(?:([0-9.]+(?:[eEdD][+-]?\d+)?)\*)?([AB])\((\d+)\)")


def parse_sum(rhs, which):
    """`-A(3)+0.05*A(7)-2*A(9)` -> [(signed coef, index0)]; `0` -> []"""
    rhs = rhs.replace(" ", "")
    if rhs in ("0", "0.0", "0.0d0"):
        return []
    pos, terms = 0, []
    while pos < len(rhs):
        m = _term.match(rhs, pos)
        if not m or m.group(3) != which or (pos > 0 and m.group(1) == ""):
            raise ValueError("cannot parse sum: %r at %d" % (rhs, pos))
        coef = 1.0 if m.group(2) is None else f32lit(m.group(2))
        if m.group(1) == "-":
            coef = -coef
        terms.append((coef, int(m.group(4)) - 1))
        pos = m.end()
    return terms


_fac = re.compile(r"(RCT|V|F)\((\d+)\)|([0-9.]+(?:[eEdD][+-]?\d+)?)")


def parse_product(rhs):
    """`RCT(7)*V(90)*F(1)*F(1)` -> (rct0, [factor codes]).  Factor code: ('V',i0) | ('F',i0) | ('C',value)."""
    toks = rhs.replace(" ", "").split("*")
    m = _fac.fullmatch(toks[0])
    if not m or m.group(1) != "RCT":
        raise ValueError("product does not start with RCT: %r" % rhs)
    rct = int(m.group(2)) - 1
    facs = []
    for t in toks[1:]:
        m = _fac.fullmatch(t)
        if not m or m.group(1) == "RCT":
            raise ValueError("bad factor %r in %r" % (t, rhs))
        if m.group(1):
            facs.append((m.group(1), int(m.group(2)) - 1))
        else:
            facs.append(("C", f32lit(m.group(3))))
    return rct, facs


def parse_params(path):
    vals = {}
    for _, st in logical_lines(path):
        m = re.match(r"PARAMETER\s*\(\s*(\w+)\s*=\s*(\d+)\s*\)", st, re.I)
        if m:
            vals[m.group(1).upper()] = int(m.group(2))
    return vals


def extract(src_dir, mech):
    sfx = {"gas": "g", "aer": "a", "tot": "t"}[mech]
    par = parse_params(os.path.join(src_dir, "%s_Parameters.h" % mech))
    nvar, nfix, nreact, nnz = par["NVAR"], par["NFIX"], par["NREACT"], par["LU_NONZERO"]
    lines = logical_lines(os.path.join(src_dir, "%s.f" % mech))

    def section(start_re):
        it = iter(lines)
        for _, st in it:
            if re.match(start_re, st, re.I):
                break
        else:
            raise ValueError("section %s not found" % start_re)
        body = []
        for no, st in it:
            if re.fullmatch(r"END", st.strip(), re.I):
                return body
            body.append((no, st))
        raise ValueError("unterminated section")

    asg = re.compile(r"(\w+)\((\d+)\)\s*=\s*(.+)")

    # ---- Fun
    A = [None] * nreact
    vd = [None] * nvar
    for no, st in section(r"SUBROUTINE\s+Fun_%s\b" % sfx):
        m = asg.match(st)
        if not m:
            continue
        name, idx, rhs = m.group(1), int(m.group(2)) - 1, m.group(3)
        if name == "A":
            rct, facs = parse_product(rhs)
            assert rct == idx, (no, st)
            A[idx] = facs
        elif name == "Vdot":
            vd[idx] = parse_sum(rhs, "A")
        else:
            raise ValueError("unexpected statement in Fun: %r" % st)
    assert all(a is not None for a in A) and all(v is not None for v in vd)

    # ---- Jac_SP
    Bmap = {}      # source B index (0-based) -> (rct, facs)
    jv = [None] * nnz
    for no, st in section(r"SUBROUTINE\s+Jac_SP_%s\b" % sfx):
        m = asg.match(st)
        if not m:
            continue
        name, idx, rhs = m.group(1), int(m.group(2)) - 1, m.group(3)
        if name == "B":
            if re.fullmatch(r"RCT\(\d+\)", rhs.strip()):
                Bmap[idx] = (int(rhs.strip()[4:-1]) - 1, [])
            else:
                Bmap[idx] = parse_product(rhs)
        elif name == "JVS":
            jv[idx] = parse_sum(rhs, "B")
        else:
            raise ValueError("unexpected statement in Jac_SP: %r" % st)
    assert all(j is not None for j in jv)
    b_order = sorted(Bmap)                     # compact the B numbering (gaps = derivatives w.r.t. fixed species)
    b_new = {old: new for new, old in enumerate(b_order)}

    # ---- sparsity
    data = {"LU_ICOL": {}, "LU_CROW": {}, "LU_DIAG": {}}
    for no, st in section(r"BLOCK\s*DATA\s+JACOBIAN_SPARSE_DATA_%s\b" % sfx):
        m = re.match(r"DATA\s*\(\s*(LU_\w+?)_%s\s*\(i\)\s*,\s*i\s*=\s*(\d+)\s*,\s*(\d+)\s*\)\s*/(.*)/" % sfx, st, re.I)
        if not m:
            m2 = re.match(r"DATA\s+(LU_\w+?)_%s\s*/(.*)/" % sfx, st, re.I)   # whole-array form
            if m2:
                vals = [int(x) for x in m2.group(2).replace(" ", "").split(",") if x]
                for k, v in enumerate(vals):
                    data[m2.group(1).upper()][1 + k] = v
            continue
        vals = [int(x) for x in m.group(4).replace(" ", "").split(",") if x]
        lo, hi = int(m.group(2)), int(m.group(3))
        assert len(vals) == hi - lo + 1, (no, len(vals), lo, hi)
        for k, v in enumerate(vals):
            data[m.group(1).upper()][lo + k] = v
    icol = np.array([data["LU_ICOL"][i] for i in range(1, nnz + 1)], np.int32) - 1
    crow = np.array([data["LU_CROW"][i] for i in range(1, nvar + 2)], np.int32) - 1
    diag = np.array([data["LU_DIAG"][i] for i in range(1, nvar + 2)], np.int32) - 1
    assert crow[0] == 0 and crow[-1] == nnz and diag[-1] == nnz
    for k in range(nvar):
        row = icol[crow[k]:crow[k + 1]]
        assert np.all(np.diff(row) > 0) and icol[diag[k]] == k and crow[k] <= diag[k] < crow[k + 1]

    # ---- constants used as factors (all are exact small numbers; slot 0 is the padding factor 1.0)
    consts = [1.0]

    def code(f):
        kind, v = f
        if kind == "V":
            assert 0 <= v < nvar
            return v
        if kind == "F":
            assert 0 <= v < nfix
            return nvar + v
        if v not in consts:
            consts.append(v)
        return nvar + nfix + consts.index(v)

    def pack_products(items):   # items: list of (rct, facs) -> rct[], ptr[], fac[]
        rct = np.array([r for r, _ in items], np.int32)
        ptr = np.zeros(len(items) + 1, np.int32)
        fac = []
        for i, (_, fs) in enumerate(items):
            fac += [code(f) for f in fs]
            ptr[i + 1] = len(fac)
        return rct, ptr, np.array(fac, np.int32)

    a_rct, a_ptr, a_fac = pack_products([(i, A[i]) for i in range(nreact)])
    b_rct, b_ptr, b_fac = pack_products([Bmap[o] for o in b_order])

    def pack_sums(rows, remap=None):
        ptr = np.zeros(len(rows) + 1, np.int32)
        idx, coef = [], []
        for i, terms in enumerate(rows):
            for c, j in terms:
                idx.append(remap[j] if remap else j)
                coef.append(c)
            ptr[i + 1] = len(idx)
        return ptr, np.array(idx, np.int32), np.array(coef, np.float64)

    vd_ptr, vd_idx, vd_coef = pack_sums(vd)
    jv_ptr, jv_idx, jv_coef = pack_sums(jv, b_new)

    return dict(nvar=nvar, nfix=nfix, nreact=nreact, nnz=nnz, crow=crow, icol=icol, diag=diag[:nvar].copy(),
                a_ptr=a_ptr, a_fac=a_fac, b_rct=b_rct, b_ptr=b_ptr, b_fac=b_fac,
                vd_ptr=vd_ptr, vd_idx=vd_idx, vd_coef=vd_coef, jv_ptr=jv_ptr, jv_idx=jv_idx, jv_coef=jv_coef,
                consts=np.array(consts, np.float64))


def write_mech(path, t):
    hdr = np.array([MAGIC, VERSION, t["nvar"], t["nfix"], t["nreact"], t["nnz"], len(t["a_fac"]), len(t["b_rct"]),
                    len(t["b_fac"]), len(t["vd_idx"]), len(t["jv_idx"]), len(t["consts"])], np.int32)
    with open(path, "wb") as f:
        f.write(hdr.tobytes())
        for k in ("crow", "icol", "diag", "a_ptr", "a_fac", "b_rct", "b_ptr", "b_fac",
                  "vd_ptr", "vd_idx", "jv_ptr", "jv_idx"):
            f.write(np.ascontiguousarray(t[k], np.int32).tobytes())
        if f.tell() % 8:
            f.write(b"\0" * (8 - f.tell() % 8))
        for k in ("vd_coef", "jv_coef", "consts"):
            f.write(np.ascontiguousarray(t[k], np.float64).tobytes())


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src"
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(__file__), "..", "mistra_amd", "mech")
    os.makedirs(out, exist_ok=True)
    for mech in ("gas", "aer", "tot"):
        t = extract(src, mech)
        write_mech(os.path.join(out, mech + ".mech"), t)
        print("%s: nvar=%d nfix=%d nreact=%d nnz=%d  A-factors=%d  B=%d (factors %d)  Vdot terms=%d  JVS terms=%d  "
              "nonzero JVS=%d  consts=%s" % (mech, t["nvar"], t["nfix"], t["nreact"], t["nnz"], len(t["a_fac"]),
                                             len(t["b_rct"]), len(t["b_fac"]), len(t["vd_idx"]), len(t["jv_idx"]),
                                             int(np.sum(np.diff(t["jv_ptr"]) > 0)), t["consts"].tolist()))


if __name__ == "__main__":
    main()
