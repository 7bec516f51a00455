"""Per-round cycles of the LDS VM programs (cell 0, wave 0) next to the schedule's rows per round.
Run on the GPU box: python tools/profile_rounds.py [mech]   (MISTRA_CHEM_PROFILE=2 diagnostics of capi.cpp)"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
os.environ['MISTRA_CHEM_PROFILE'] = '2'
from mistra_amd import chem
from mistra_amd.workload import make_batch
mech = sys.argv[1] if len(sys.argv) > 1 else 'tot'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
chem.init(0)
var, fix, rconst = make_batch(mech, 0, n, 'cpu')
res = chem.integrate(mech, var.numpy(), fix.numpy(), rconst.numpy())
print(mech, 'cell 0: Nstp', res.stats[0, 2], 'Ndec', res.stats[0, 5], 'Nsol', res.stats[0, 6], flush=True)
