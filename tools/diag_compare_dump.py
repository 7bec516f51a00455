"""Diagnostic: first-step dumps (mistra_chem_debug_first_step) of two builds of the library, compared bit for bit per section:
   python tools/diag_compare_dump.py libA.so libB.so [mech] [ncell]       (libmistra_chem.so = the product, other names in tools/diaglib/; one process per library)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag import REPO, diag_env
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, REPO)
    import numpy as np
    from mistra_amd import chem
    from mistra_amd.workload import make_batch
    mech, n, out = sys.argv[2], int(sys.argv[3]), sys.argv[4]
    chem.init(0)
    var, fix, rconst = make_batch(mech, 0, n, 'cpu')
    import ctypes as C
    dp = C.POINTER(C.c_double)
    from mistra_amd.mechtab import load, MECH_IDS
    m = load(mech)
    chem.lib().mistra_chem_debug_first_step.argtypes = [C.c_int, C.c_int, dp, dp, dp, C.c_double, C.c_double, dp]
    d = np.empty((n, 5 * m.nvar + 2 * m.nnz + 2))
    v, f, r = (np.ascontiguousarray(x.numpy()) for x in (var, fix, rconst))
    P = lambda a: a.ctypes.data_as(dp)
    assert chem.lib().mistra_chem_debug_first_step(MECH_IDS[mech], n, P(v), P(f), P(r), 0.0, 10.0, P(d)) == 0
    res = chem.integrate(mech, var.numpy(), fix.numpy(), rconst.numpy())
    np.savez(out, dump=d, var=res.var, stats=res.stats)
    sys.exit(0)
import numpy as np
a, b = sys.argv[1], sys.argv[2]
mech = sys.argv[3] if len(sys.argv) > 3 else 'tot'
n = int(sys.argv[4]) if len(sys.argv) > 4 else 64
outs = []
for lib in (a, b):
    out = '/tmp/dump_%s.npz' % lib
    env = diag_env(lib)
    subprocess.run([sys.executable, os.path.abspath(__file__), '--child', mech, str(n), out], check=True, env=env)
    outs.append(np.load(out))
sys.path.insert(0, REPO)
from mistra_amd.mechtab import load
m = load(mech)
nv, nz = m.nvar, m.nnz
sec = [('Fcn0', 0, nv), ('Ghimj prepared', nv, nv + nz), ('Ghimj factorised', nv + nz, nv + 2 * nz), ('R', nv + 2 * nz, 2 * nv + 2 * nz),
       ('K1', 2 * nv + 2 * nz, 3 * nv + 2 * nz), ('K2', 3 * nv + 2 * nz, 4 * nv + 2 * nz), ('K3', 4 * nv + 2 * nz, 5 * nv + 2 * nz), ('Err,H', 5 * nv + 2 * nz, 5 * nv + 2 * nz + 2)]
da, db = outs[0]['dump'], outs[1]['dump']
for name, i, j in sec:
    x, y = da[:, i:j], db[:, i:j]
    same = (x == y) | (np.isnan(x) & np.isnan(y))
    print('%-18s %s' % (name, 'bit-identical' if same.all() else '%d of %d values differ, max rel %.2e' % ((~same).sum(), same.size, np.nanmax(np.abs(x - y) / (np.abs(y) + 1e-300)))))
print('integrate(): VAR %s, statistics %s' % ('bit-identical' if np.array_equal(outs[0]['var'], outs[1]['var']) else 'DIFFER', 'identical' if np.array_equal(outs[0]['stats'], outs[1]['stats']) else 'DIFFER'))
