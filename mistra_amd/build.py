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
        ring_register_report()
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
    return report


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
