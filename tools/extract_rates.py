#!/usr/bin/env python3
"""Extracts the rate-constant assignments of the reference's generated Update_RCONST_x (gas.f:275-666 | aer.f:304-1364 |
tot.f:1040-2768) into a data table: mistra_amd/mech/<mech>.rates (JSON).

Every `RCONST(i) = <expression>` becomes a postfix program over
    ["num", value]            a literal, already folded to the double the reference's compiler makes of it: `3.2d-11` and `1._dp`
                              are doubles, `300.` and `0.21` are DEFAULT-REAL literals (the double nearest the float32, SURVEY.md
                              §2.1), integers stay integers until an operation promotes them; literal-only subexpressions are
                              folded in the type Fortran evaluates them in (`8.314/101325.` is a float32 division)
    ["var", name]             a scalar of COMMON /kpp_rate_x/ or a dummy of x_drive (conv1, xhal, xliq1, cvv1, ...)
    ["arr", name, i, j]       an array element with constant (1-based) indices, ind_X parameters resolved: ph_rat(3), FIX(2),
                              yxkmt(ind_HNO3,1), C(ind_Hplz), ...
    ["call", fname, nargs]    a rate-law function of kpp.f90:7127-8601 on the nargs values below it
    ["+"] ["-"] ["*"] ["/"] ["neg"]
in the reference's evaluation order (left to right within a precedence level).  The script needs the reference tree
(/root/reference/src); its output is committed.  Run: python tools/extract_rates.py [mech ...]"""
import json
import os
import re
import sys

import numpy as np

REF = os.environ.get("MISTRA_REFERENCE_SRC", "/root/reference/src")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mistra_amd", "mech")
SPAN = {"gas": ("gas.f", "g"), "aer": ("aer.f", "a"), "tot": ("tot.f", "t")}

TOKEN = re.compile(r"\s*(?:(\d+\.?\d*(?:[dDeE][+-]?\d+)?(?:_dp)?|\.\d+(?:[dDeE][+-]?\d+)?(?:_dp)?)|([A-Za-z_][A-Za-z_0-9]*)|(\*\*|[-+*/(),]))")


def statements(path, first_marker, last_marker):
    """fixed-form statements of one subroutine, continuation lines joined"""
    lines = open(path, errors="replace").read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(first_marker, l))
    out = []
    for l in lines[start:]:
        if re.match(last_marker, l) and out:
            break
        if not l.strip() or l[0] in "cC*!" or l.lstrip().startswith("!"):
            continue
        if len(l) > 5 and l[:5].strip() == "" and l[5] not in " 0":
            out[-1] += l[6:].rstrip()
        else:
            out.append(l.rstrip())
    return out


def parameters(path):
    p = {}
    for m in re.finditer(r"PARAMETER\s*\(\s*(\w+)\s*=\s*(\d+)\s*\)", open(path, errors="replace").read()):
        p[m.group(1).lower()] = int(m.group(2))
    return p


class Parser:
    def __init__(self, text, params):
        self.toks = []
        pos = 0
        text = text.strip()
        while pos < len(text):
            m = TOKEN.match(text, pos)
            if not m:
                raise ValueError("cannot tokenise %r at %d" % (text, pos))
            self.toks.append(m.group(1) or m.group(2) or m.group(3))
            pos = m.end()
        self.i = 0
        self.params = params

    def peek(self):
        return self.toks[self.i] if self.i < len(self.toks) else None

    def take(self, want=None):
        t = self.peek()
        if want is not None and t != want:
            raise ValueError("expected %r, got %r in %r" % (want, t, self.toks))
        self.i += 1
        return t

    # nodes: ("num", kind, value) kind in int|real|double ; ("var", name) ; ("arr", name, [idx]) ; ("call", name, [args]) ;
    #        ("bin", op, a, b) ; ("neg", a)
    def expr(self):
        sign = None
        if self.peek() in ("+", "-"):
            sign = self.take()
        node = self.term()
        if sign == "-":
            node = ("neg", node)
        while self.peek() in ("+", "-"):
            op = self.take()
            node = ("bin", op, node, self.term())
        return node

    def term(self):
        node = self.factor()
        while self.peek() in ("*", "/"):
            op = self.take()
            node = ("bin", op, node, self.factor())
        return node

    def factor(self):
        node = self.primary()
        if self.peek() == "**":
            raise ValueError("** is not used in Update_RCONST_x and not supported here")
        return node

    def primary(self):
        t = self.take()
        if t == "(":
            node = self.expr()
            self.take(")")
            return node
        if re.match(r"[\d.]", t):
            return number(t)
        name = t.lower()
        if self.peek() == "(":
            self.take("(")
            args = []
            if self.peek() != ")":
                args.append(self.expr())
                while self.peek() == ",":
                    self.take(",")
                    args.append(self.expr())
            self.take(")")
            return ("ref", name, args)
        if name in self.params:
            return ("num", "int", self.params[name])
        return ("var", name)


def number(t):
    low = t.lower()
    if low.endswith("_dp"):
        return ("num", "double", float(low[:-3].replace("d", "e")))
    if "d" in low:
        return ("num", "double", float(low.replace("d", "e")))
    if "." in low or "e" in low:
        return ("num", "real", float(np.float32(float(low))))
    return ("num", "int", int(low))


RANK = {"int": 0, "real": 1, "double": 2}


def fold(node):
    """constant folding of literal-only subtrees in the type Fortran gives them"""
    if node[0] == "neg":
        a = fold(node[1])
        if a[0] == "num":
            return ("num", a[1], -a[2])
        return ("neg", a)
    if node[0] == "bin":
        a, b = fold(node[2]), fold(node[3])
        if a[0] == "num" and b[0] == "num":
            kind = a[1] if RANK[a[1]] >= RANK[b[1]] else b[1]
            x, y = a[2], b[2]
            if kind == "int":
                v = {"+": x + y, "-": x - y, "*": x * y, "/": int(x / y) if y else 0}[node[1]]      # Fortran integer division truncates
            elif kind == "real":
                fx, fy = np.float32(x), np.float32(y)
                v = float({"+": fx + fy, "-": fx - fy, "*": fx * fy, "/": fx / fy}[node[1]])
            else:
                v = {"+": x + y, "-": x - y, "*": x * y, "/": x / y}[node[1]]
            return ("num", kind, v)
        return ("bin", node[1], a, b)
    if node[0] == "ref":
        return ("ref", node[1], [fold(x) for x in node[2]])
    return node


ARRAYS = {"ph_rat", "fix", "c", "yxkmt", "yxkmtd", "ycw", "ycwd", "yhenry", "yxeq", "ykef", "ykeb"}


def emit(node, prog, funcs):
    if node[0] == "num":
        prog.append(["num", float(node[2]) if node[1] != "int" else int(node[2])])
    elif node[0] == "var":
        prog.append(["var", node[1]])
    elif node[0] == "neg":
        emit(node[1], prog, funcs)
        prog.append(["neg"])
    elif node[0] == "bin":
        emit(node[2], prog, funcs)
        emit(node[3], prog, funcs)
        prog.append([node[1]])
    elif node[0] == "ref":
        name, args = node[1], node[2]
        if name in ARRAYS:
            idx = []
            for a in args:
                if a[0] != "num" or a[1] != "int":
                    raise ValueError("array index is not a constant: %r" % (node,))
                idx.append(a[2])
            prog.append(["arr", name] + idx)
        else:
            for a in args:
                emit(a, prog, funcs)
            prog.append(["call", name, len(args)])
            funcs.add(name)
    else:
        raise ValueError(node)


def extract(mech):
    fname, sfx = SPAN[mech]
    params = parameters(os.path.join(REF, "%s_Parameters.h" % mech))
    stm = statements(os.path.join(REF, fname), r"\s+SUBROUTINE Update_RCONST_%s" % sfx, r"\s+END\b")
    progs, funcs, names = {}, set(), set()
    for s in stm:
        m = re.match(r"\s+RCONST\((\d+)\)\s*=\s*(.*)$", s)
        if not m:
            continue
        prog = []
        emit(fold(Parser(m.group(2), params).expr()), prog, funcs)
        progs[int(m.group(1))] = prog
        names.update(t[1] for t in prog if t[0] == "var")
    n = params["nreact"]
    assert sorted(progs) == list(range(1, n + 1)), "not every RCONST is assigned exactly once"
    table = {"mech": mech, "nreact": n, "source": "%s: SUBROUTINE Update_RCONST_%s" % (fname, sfx),
             "functions": sorted(funcs), "scalars": sorted(names),
             "arrays": sorted({t[1] for p in progs.values() for t in p if t[0] == "arr"}),
             "programs": [progs[i] for i in range(1, n + 1)]}
    path = os.path.join(OUT, mech + ".rates.json")
    json.dump(table, open(path, "w"), separators=(",", ":"))
    print(mech, n, "reactions; functions", table["functions"], "; scalars", table["scalars"], "; arrays", table["arrays"], "->", path, os.path.getsize(path), "bytes")
    write_binary(mech, table, params)


# ---- binary form for the device evaluator (mistra_amd/csrc/rates.hip).
# A cell's inputs are ONE vector of doubles ("env"): what Update_RCONST_x and its rate laws read from COMMON /cb_1/,
# /kpp_rate_x/, /ph_r_x/ and C (kpp.f90:7140, gas_Global.h:76-96 | aer_Global.h:76-88 | tot_Global.h:76-98), and nothing else:
#   aircc te h2oppm pk | the scalars the assignments name | the array elements they name | what the rate-law FUNCTIONS read
# from COMMON themselves (FSLOT_NAMES, in the fixed order the device code indexes them by; -1 where a mechanism has no such
# entry).  The gas layout is spelled out (its fixture, tests/golden/rates_gas.npz, is in this order); aer and tot are built.
GAS_ENV = (["aircc", "te", "h2oppm", "pk", "conv1", "xhal", "xiod", "xhet1", "xhet2", "ycwd(1)", "ycwd(2)"] +
           ["ph_rat(%d)" % i for i in range(1, 48)] + ["fix(1)", "fix(2)", "fix(3)"] +
           ["yxkmtd(ind_hno3,1)", "yxkmtd(ind_hno3,2)", "yxkmtd(ind_n2o5,1)", "yxkmtd(ind_n2o5,2)", "yxkmtd(ind_nh3,1)",
            "yxkmtd(ind_nh3,2)", "yxkmtd(ind_h2so4,1)", "yxkmtd(ind_h2so4,2)", "yhenry(ind_hno3)", "yxeq(ind_hno3)",
            "c(ind_hno3)", "c(ind_hno3l1)", "c(ind_hno3l2)"])


def fslot_names():
    """what the rate-law functions read from COMMON, symbolic; index = position the device code uses (rates.hip: FS_*)"""
    n = []
    n += ["fix(indf_h2ol%d)" % a for a in (1, 2, 3, 4)]                        # 0   FS_H2OL
    n += ["c(ind_clml%d)" % a for a in (1, 2, 3, 4)]                           # 4   FS_CLM
    n += ["c(ind_brml%d)" % a for a in (1, 2, 3, 4)]                           # 8   FS_BRM
    n += ["yxkmt(ind_n2o5,%d)" % a for a in (1, 2, 3, 4)]                      # 12  FS_YXKMT_N2O5
    n += ["yxkmt(ind_clno3,%d)" % a for a in (1, 2, 3, 4)]                     # 16  FS_YXKMT_CLNO3
    n += ["yxkmt(ind_brno3,%d)" % a for a in (1, 2, 3, 4)]                     # 20  FS_YXKMT_BRNO3
    n += ["ycw(%d)" % a for a in (1, 2, 3, 4)]                                 # 24  FS_YCW
    for sp in ("n2o5", "brno3", "clno3", "hno3", "nh3", "h2so4"):              # 28  FS_YXKMTD (species-major, 2 bins)
        n += ["yxkmtd(ind_%s,%d)" % (sp, a) for a in (1, 2)]
    n += ["ycwd(1)", "ycwd(2)"]                                                # 40  FS_YCWD
    n += ["yhenry(ind_hno3)", "yxeq(ind_hno3)", "c(ind_hno3)"]                 # 42 43 44
    n += ["c(ind_hno3l1)", "c(ind_hno3l2)", "c(ind_no3ml1)", "c(ind_no3ml2)"]  # 45 46 47 48
    n += ["xhal"]                                                              # 49
    return n


def resolve(name, params):
    """symbolic array element -> numeric key 'arr(i,j)'; None if the mechanism does not have the species / the bin"""
    m = re.match(r"(\w+)\((.*)\)$", name)
    if not m:
        return name
    idx = []
    for t in m.group(2).split(","):
        t = t.strip()
        if t.isdigit():
            idx.append(int(t))
        elif t in params:
            idx.append(params[t])
        else:
            return None
    return "%s(%s)" % (m.group(1), ",".join(str(i) for i in idx))


NSPEC_BINS = {"gas": 2, "aer": 2, "tot": 4}


def env_names(mech, table, params):
    if mech == "gas":
        names = [resolve(n, params) for n in GAS_ENV]
        assert None not in names
        return names
    names = ["aircc", "te", "h2oppm", "pk"] + table["scalars"]
    arr = sorted({(t[1],) + tuple(t[2:]) for p in table["programs"] for t in p if t[0] == "arr"})
    names += ["%s(%s)" % (a[0], ",".join(str(i) for i in a[1:])) for a in arr]
    have = set(names)
    for n in fslot_names():
        r = resolve(n, params)
        if r is None or r in have:
            continue
        m = re.match(r"(\w+)\((?:\d+,)?(\d+)\)$", r)
        if m and m.group(1) in ("yxkmt", "ycw") and int(m.group(2)) > NSPEC_BINS[mech]:
            continue                                                            # a bin the mechanism does not have
        names.append(r)
        have.add(r)
    return names


FUNC_ID = {"farr": (0, 2), "farr_sp": (1, 4), "atk_3": (2, 5), "atk_3f": (3, 5), "shno3": (4, 6), "fbck": (5, 7), "fbckj": (6, 6),
           "fbck2": (7, 6), "sp_17": (8, 2), "sp_23": (9, 6), "fcn": (10, 1), "dms_add": (11, 0), "fdhetg": (12, 2),
           "fdheta": (12, 2), "fdhett": (12, 2),      # one routine: the three differ in the caq line, told apart by the NO3- slots
           "farr2": (13, 2), "fhet_t": (14, 3), "fhet_da": (15, 5), "fhet_dt": (15, 5), "fliq_60": (16, 4), "dmin2": (17, 1),
           "dmin3": (18, 1), "flsc4": (19, 3), "flsc5": (20, 3), "flsc6": (21, 2), "uplim": (22, 4), "uparm": (23, 5),
           "uplip": (24, 3), "uparp": (25, 4)}      # id, nargs
OP = {"const": 0, "env": 1, "+": 2, "-": 3, "*": 4, "/": 5, "neg": 6, "call": 7}


def write_binary(mech, table, params):
    import struct
    env = env_names(mech, table, params)
    slot = {name: i for i, name in enumerate(env)}
    fslot = []
    for n in fslot_names():
        r = resolve(n, params)
        fslot.append(slot.get(r, -1) if r is not None else -1)
    consts, words, offs = [], [], [0]

    def const(v):
        consts.append(float(v))
        return len(consts) - 1
    for prog in table["programs"]:
        for t in prog:
            if t[0] == "num":
                words.append(OP["const"] | (const(t[1]) << 8))
            elif t[0] == "var":
                words.append(OP["env"] | (slot[t[1]] << 8))
            elif t[0] == "arr":
                words.append(OP["env"] | (slot["%s(%s)" % (t[1], ",".join(str(i) for i in t[2:]))] << 8))
            elif t[0] == "call":
                fid, nargs = FUNC_ID[t[1]]
                assert nargs == t[2], (t, nargs)
                words.append(OP["call"] | (fid << 8))
            else:
                words.append(OP[t[0]])
        offs.append(len(words))
    path = os.path.join(OUT, mech + ".rates")
    with open(path, "wb") as f:
        f.write(struct.pack("<8i", 0x5441524B, 2, table["nreact"], len(env), len(consts), len(words), len(fslot), 0))      # 'KRAT' v2
        f.write(np.asarray(consts, np.float64).tobytes())
        f.write(np.asarray(offs, np.int32).tobytes())
        f.write(np.asarray(words, np.int32).tobytes())
        f.write(np.asarray(fslot, np.int32).tobytes())
    json.dump({"env": env, "fslot": fslot}, open(os.path.join(OUT, mech + ".rates_env.json"), "w"), separators=(",", ":"))
    print("   binary:", path, os.path.getsize(path), "bytes;", len(env), "env doubles per cell,", len(consts), "constants,", len(words), "words")


if __name__ == "__main__":
    for mech in (sys.argv[1:] or ["gas", "aer", "tot"]):
        extract(mech)
