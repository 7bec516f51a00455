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
SOURCES = ["mech_tables.cpp", "schedule.cpp", "capi.cpp", "ros3_kernel.hip"]


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
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
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
        cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
