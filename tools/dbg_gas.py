import os, sys, numpy as np
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path.insert(0, R)
from mistra_amd import chem
for mech in ('gas','aer','tot'):
    g=np.load(os.path.join(R,'tests/golden/integrate_%s.npz'%mech))
    res=chem.integrate(mech, g['var_in'][:2], g['fix'][:2], g['rconst'][:2])
    print(os.path.basename(os.environ.get('MISTRA_CHEM_LIB','default')), mech, res.ierr, res.stats[0], g['stats'][0], np.nanmax(np.abs(res.var-g['var_out'][:2])))
