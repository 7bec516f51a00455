"""In-tree build of libmistra_chem.so (HIP kernels for gfx950 + the C ABI of include/mistra_chem.h).

`python -m mistra_amd.build` or `build_lib()`.  hipcc cross-compiles for gfx950 without a GPU present.
The library is written to mistra_amd/lib/ (git-ignored, shipped to the GPU box with the working tree).
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(PKG, "build")
LIB = os.path.join(LIBDIR, "libmistra_chem.so")

ARCH = "gfx950"
# -ffp-contract=off: one rounding per multiply and per add, as in the reference built without FMA contraction
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]
SOURCES = ["mech_tables.cpp", "schedule.cpp", "capi.cpp", "ros3_kernel.hip", "rates.hip", "pack.hip"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    cc = hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))]
    headers.append(os.path.join(PKG, "..", "include", "mistra_chem.h"))
    headers.append(os.path.abspath(__file__))
    objs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [path] + headers):
            cmd = [cc, "--offload-arch=" + ARCH] + COMMON + ["-c", path, "-o", obj]
            if src.endswith(".hip"):
                cmd += ["-Rpass-analysis=kernel-resource-usage"] if verbose else []
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
    if force or _stale(LIB, objs):
        # the look-ahead rings of ros3_kernel.hip sit in registers the compiler does not know are busy: no library is linked
        # from a kernel object in which a ring-using function's own registers reach its ring (raises)
        isa = ring_register_report()
        hits = isa_hazard_report(isa["__isa_text__"])
        if hits:
            raise RuntimeError("hazards the hardware does not interlock inside hand-written assembly:\n  " + "\n  ".join(hits))
        cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


def ring_register_report(isa_path=None):
    """Checks the one assumption the table look-ahead ring of ros3_kernel.hip rests on (see the comment there): in the
    non-inlined device functions, every register the COMPILER allocates stays below the ring's blocks (v192.. or, in the
    low placement, v64..), so a
    table load landing in the ring can never hit a compiler value.  Compiles the kernel source to gfx950 assembly and
    scans it.  Returns {function: highest VGPR named outside inline asm}; raises if a ring-using function reaches its ring."""
    import re
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        if isa_path is None:
            isa_path = os.path.join(tmp, "ros3_kernel.s")
            cmd = [hipcc(), "--offload-arch=" + ARCH] + [f for f in COMMON if f != "-fPIC"] + \
                  ["-S", "--offload-device-only", os.path.join(CSRC, "ros3_kernel.hip"), "-o", isa_path]
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        lines = open(isa_path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_ZN6mistra.*:", l)]
    starts.append((len(lines), "end"))
    report = {}
    ring_users = set()
    for (i, name), (j, _) in zip(starts, starts[1:]):
        in_asm, hi = False, 0
        for l in lines[i:j]:
            if "ASMSTART" in l:
                in_asm = True
            elif "ASMEND" in l:
                in_asm = False
            elif in_asm and re.search(r"global_load_dwordx4 v\[(64|192):\d+\], v\[\d+:\d+\], off", l):
                ring_users.add(name)         # the function issues ring loads (vm_ring_load) itself; (vm_run's own ring is
                                             # loaded and consumed inside ONE asm statement that lists it as clobbered)
            elif "Folded Spill" in l or "Folded Reload" in l:
                pass    # prologue / epilogue saves of callee-saved ring blocks: before the first ring load, after the drain
            elif not in_asm and not l.strip().startswith((";", ".")):
                for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", l):
                    hi = max(hi, int(m.group(1) or m.group(3)))
        report[name] = hi
        low = re.search(r"(gsum_run|tail_solve|tail_solve_columns|scale_run)I.*Lb([01])E+[A-Z]", name)       # last template argument: ring placement LOW
        if low:
            limit = 64 if low.group(2) == "1" else 192
            if hi >= limit:
                raise RuntimeError("%s: the compiler allocates v%d, inside the look-ahead ring's register blocks (v%d..)" % (name, hi, limit))
        elif name in ring_users and "ros3_integrate_kernel" not in name:
            raise RuntimeError("%s issues look-ahead ring loads but is not covered by the register check" % name)
    report["__isa_text__"] = "\n".join(lines)
    return report


# ---- hazards inside hand-written assembly.  The compiler's hazard recogniser and its s_waitcnt insertion do not look inside asm statements,
#      and ros3_kernel.hip carries ~4 000 lines of hand-written / generated instructions (vm_exec_asm.inc, gsum_exec_asm.inc, the ring
#      helpers, the DPP chains).  Three classes are checked on the ISA the compiler emits, each the cause of a wrong result or a GPU
#      memory fault met while those were written:
#   (i)   an SGPR written by v_readfirstlane / v_readlane and used as the scalar base of a vector-memory instruction less than 5 wait
#         states later (round 3: a table base moved to scalar registers in front of a global_load -> memory access fault);
#   (ii)  a DPP instruction whose DPP-read source (src0) was written by a VALU instruction less than 2 wait states earlier (the hardware
#         does not interlock the read: tools/ubench/dpp.hip returns wrong sums with none);
#   (iii) an `s_waitcnt vmcnt(N)` inside an asm block with N not below the number of distinct registers-in-flight slots the function's own
#         asm loads fill (such a wait can never guarantee that the oldest slot has landed).
# An instruction is one wait state, `s_nop N` is N + 1.  The scan is linear per function (labels are ignored: conservative for the
# straight-line blocks these sequences are).
def isa_hazard_report(isa_text, sgpr_vmem_wait=5, dpp_wait=2, seen=None):
    """-> list of findings (empty = clean).  seen: optional dict that receives how many instructions of each class were examined."""
    import re
    hits = []
    seen = seen if seen is not None else {}
    for k in ("vmem_scalar_base", "dpp", "vmcnt", "functions"):
        seen.setdefault(k, 0)
    func, insts = None, []

    def regs(tok, kind):
        """register numbers named by an operand token of class `kind` ('s' or 'v'): s3, s[2:3], -v[4:5], |v7| ..."""
        tok = tok.strip().lstrip("-").strip("|")
        m = re.match(r"^%s(\d+)$" % kind, tok)
        if m:
            return {int(m.group(1))}
        m = re.match(r"^%s\[(\d+):(\d+)\]$" % kind, tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        return set()

    def flush():
        if func is None:
            return
        seen["functions"] += 1
        depth = len({i["ops"][0] for i in insts if i["asm"] and i["op"].startswith(("global_load", "buffer_load")) and i["ops"]})
        for n, it in enumerate(insts):
            if not it["asm"]:
                continue
            op, ops = it["op"], it["ops"]
            if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
                base = set()
                for o in ops:
                    if re.match(r"^s\[\d+:\d+\]$", o.strip()):
                        base |= regs(o, "s")
                if base:
                    seen["vmem_scalar_base"] += 1
                    waited, k = 0, n - 1
                    while k >= 0 and waited < sgpr_vmem_wait:
                        p = insts[k]
                        if p["op"] in ("v_readfirstlane_b32", "v_readlane_b32") and p["ops"] and regs(p["ops"][0], "s") & base:
                            hits.append("%s: line %d `%s` reads a scalar base written by `%s` %d wait state(s) earlier (needs %d)" %
                                        (func, it["line"], it["text"], p["text"], waited, sgpr_vmem_wait))
                            break
                        waited += p["wait"]
                        k -= 1
            if "_dpp" in op or re.search(r"\b(row_newbcast|row_shr|row_shl|row_ror|row_bcast|quad_perm|row_mirror|row_half_mirror|row_share|row_xmask):?", it["text"]):
                src0 = regs(ops[1], "v") if len(ops) > 1 else set()
                seen["dpp"] += 1
                waited, k = 0, n - 1
                while src0 and k >= 0 and waited < dpp_wait:
                    p = insts[k]
                    if p["op"].startswith("v_") and p["ops"] and regs(p["ops"][0], "v") & src0:
                        hits.append("%s: line %d `%s` DPP-reads a register written by `%s` %d wait state(s) earlier (needs %d)" %
                                    (func, it["line"], it["text"], p["text"], waited, dpp_wait))
                        break
                    waited += p["wait"]
                    k -= 1
            if op == "s_waitcnt" and depth:
                m = re.search(r"vmcnt\((\d+)\)", it["text"])
                seen["vmcnt"] += 1 if m else 0
                if m and int(m.group(1)) >= depth:
                    hits.append("%s: line %d `%s` waits for at most %s loads in flight, the function's asm loads fill only %d slots" %
                                (func, it["line"], it["text"], m.group(1), depth))

    in_asm = False
    for ln, raw in enumerate(isa_text.split("\n"), 1):
        if re.match(r"^[A-Za-z_.$][\w.$]*:", raw) and not raw.startswith(".L"):
            flush()
            func, insts, in_asm = raw.split(":")[0], [], False
            continue
        if "ASMSTART" in raw:
            in_asm = True
            continue
        if "ASMEND" in raw:
            in_asm = False
            continue
        t = raw.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        parts = t.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in re.split(r",(?![^\[]*\])", parts[1].split(" offset")[0])] if len(parts) > 1 else []
        # modifiers that follow the last operand without a comma (row_newbcast:3 row_mask:0x1 ...) stay glued to it: cut them off
        ops = [o.split()[0] if o.split() else o for o in ops]
        wait = 1
        if op == "s_nop" and ops:
            wait = int(ops[0], 0) + 1
        insts.append({"op": op, "ops": ops, "asm": in_asm, "line": ln, "text": t, "wait": wait})
    flush()
    return hits


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
