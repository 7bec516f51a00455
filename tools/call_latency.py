"""Latency of the one-cell-per-call drop-in entry point (what the Fortran shim's INTEGRATE_x pays per call):
mistra_chem_integrate_common on a /GDATA_x/-shaped block, repeated.  GPU box: python tools/call_latency.py"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
from mistra_amd import chem
L = chem.lib(); chem.init(0)
for mech, mid in (('gas', 0), ('aer', 1), ('tot', 2)):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'integrate_%s.npz' % mech))
    nv, nf, nr, _ = chem.DIMS[mech]
    block = np.zeros(nv + nf + nr + 2 + 2 * nv + 2)           # C | RCONST | TIME DT | ATOL | RTOL | STEPMIN STEPMAX
    n = 300 if mech != 'gas' else 2000
    t_tot = 0.0
    for k in range(n):
        i = k % g['var_in'].shape[0]
        block[:nv] = g['var_in'][i]; block[nv:nv + nf] = g['fix'][i]; block[nv + nf:nv + nf + nr] = g['rconst'][i]
        tin, tout = C.c_double(0.0), C.c_double(10.0)
        t0 = time.perf_counter()
        rc = L.mistra_chem_integrate_common(mid, block.ctypes.data_as(C.c_void_p), C.byref(tin), C.byref(tout))
        t_tot += time.perf_counter() - t0
        assert rc == 0
        if k < g['var_in'].shape[0]:
            ref = g['var_out'][i]
            big = np.abs(ref) >= 1e-4 * np.abs(ref).max()
            assert np.allclose(block[:nv][big], ref[big], rtol=1e-10), 'wrong result'
    print('%s: %.1f us per INTEGRATE_%s call (mean of %d, one cell per call)' % (mech, 1e6 * t_tot / n, mech[0], n), flush=True)
