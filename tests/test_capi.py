"""The C-ABI library: loads on a machine without a GPU, exports every symbol include/mistra_chem.h declares, and
refuses to compute without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import REPO


@pytest.fixture(scope="module")
def lib():
    from mistra_amd.build import build_lib
    build_lib()
    from mistra_amd import chem
    return chem.lib()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(REPO, "include", "mistra_chem.h")).read()
    names = set(re.findall(r"\b(mistra_chem_\w+)\s*\(", hdr))
    assert {"mistra_chem_init", "mistra_chem_integrate", "mistra_chem_integrate_device", "mistra_chem_integrate_common",
            "mistra_chem_finalize", "mistra_chem_dims", "mistra_chem_last_error", "mistra_chem_describe"} <= names
    for n in names:
        assert hasattr(lib, n), "symbol %s declared in the header is not exported" % n


def test_dims(lib):
    from mistra_amd.chem import DIMS
    for mech, name in enumerate(("gas", "aer", "tot")):
        v = [C.c_int32() for _ in range(4)]
        assert lib.mistra_chem_dims(mech, *[C.byref(x) for x in v]) == 0
        assert tuple(x.value for x in v) == DIMS[name]
    assert lib.mistra_chem_dims(7, None, None, None, None) != 0


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mistra_amd import chem
    assert lib.mistra_chem_init(0) != 0
    assert b"no HIP device" in lib.mistra_chem_last_error()
    with pytest.raises(chem.MistraChemError):
        chem.integrate("gas", np.zeros((1, 102)), np.zeros((1, 3)), np.zeros((1, 331)))
    # (the Fortran-facing entry points bring the library up themselves; without a device that fails loudly too)
    rc = lib.mistra_chem_integrate_ex(0, 1, None, None, None, 0.0, 10.0, None, None, None, None)
    assert rc != 0 and b"no HIP device" in lib.mistra_chem_last_error()
    # compute entry point without init: error, not a silent result
    out = np.zeros(102)
    dp = C.POINTER(C.c_double)
    rc = lib.mistra_chem_integrate(0, 1, out.ctypes.data_as(dp), out.ctypes.data_as(dp), out.ctypes.data_as(dp), 0.0, 10.0,
                                   out.ctypes.data_as(dp), None, None)
    assert rc != 0


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under mistra_amd/ may reference it."""
    for root, _, files in os.walk(os.path.join(REPO, "mistra_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert "oracle." not in text and "oracle/" not in text and "kpp_ros3" not in text and "libmistra_ref" not in text, f


def test_lookahead_ring_registers_are_out_of_the_compilers_reach():
    """ros3_kernel.hip streams its tables through a ring of fixed VGPRs (v192..v247, or four blocks from v64 up for the mechanisms run at three or
    four waves per SIMD) inside two non-inlined device functions; that is only sound while the compiler's own values in those
    functions stay below the ring (build.py scans the
    generated gfx950 assembly).  Cross-compiles, no GPU needed."""
    from mistra_amd.build import ring_register_report
    rep = ring_register_report()          # raises if a function's own registers reach its ring
    dev = {k: v for k, v in rep.items() if "gsum_run" in k or "tail_solve" in k or "scale_run" in k}      # (tail_solve and tail_solve_columns)
    assert len(dev) >= 9, rep
    low = {k: v for k, v in dev.items() if "Lb1E" in k}
    assert low and max(low.values()) < 64, low      # (ring_register_report has raised already if not)


@pytest.mark.parametrize("tool", ["gen_vm_asm.py", "gen_gsum_asm.py", "gen_rates_shim.py"])
def test_generated_sources_are_up_to_date(tool):
    """mistra_amd/csrc/vm_exec_asm.inc, gsum_exec_asm.inc and shim/mistra_kpp_rates.f90 are generator output kept in the tree: what is committed is what
    the generator writes today."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", tool), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
