"""Phase profile: cycles of wave 0 per phase of the step loop, mean over the cells of a call (kernel VARIANT 1), printed by
the DIAGNOSTIC build of the library (tools/diag_dense.sh env -> tools/diaglib/libdiag_env.so; the product library has no
such switch).  GPU box: python tools/profile_phases.py [tot]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag import use_diag_lib
use_diag_lib('libdiag_env.so', MISTRA_CHEM_PROFILE='1')
from mistra_amd import chem
from mistra_amd.workload import make_batch
chem.init(0)
for mech, n in ((('tot', 512),) if len(sys.argv) > 1 else (('tot', 512), ('aer', 512), ('gas', 4096))):
    var, fix, rconst = make_batch(mech, 0, n, 'cpu')
    t0 = time.time(); res = chem.integrate(mech, var.numpy(), fix.numpy(), rconst.numpy()); dt = time.time() - t0
    print(mech, n, 'cells wall %.3fs' % dt, 'steps/cell %.1f' % res.stats[:, 2].mean(), chem.describe(mech), flush=True)
