import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mistra_amd import chem
from oracle.oracle import Oracle
chem.init(0)
mech = sys.argv[1] if len(sys.argv) > 1 else "aer"
g = np.load("tests/golden/integrate_%s.npz" % mech)
res = chem.integrate(mech, g["var_in"], g["fix"], g["rconst"])
want, ierr, st = Oracle(mech).integrate_batch(g["var_in"], g["fix"], g["rconst"])
floor = 1e-12*np.abs(want).max(axis=1, keepdims=True)
print(mech, "ierr", np.unique(res.ierr), "stats equal", np.array_equal(res.stats, st), "max rel", (np.abs(res.var-want)/(np.abs(want)+floor)).max())
