"""Diagnostic: per-cell phase ticks of the tot kernel with 1, 32 and 512 cells in flight (is the per-cell time set by
the cell's own dependency chains, or by contention for the shared table stream?)."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mistra_amd import chem
from mistra_amd.workload import make_batch
chem.init(0)
os.environ['MISTRA_CHEM_PROFILE'] = '1'
var, fix, rconst = (x.numpy() for x in make_batch('tot', 0, 512, 'cpu'))
for n in (1, 32, 256, 512):
    print('cells', n, flush=True)
    chem.integrate('tot', var[:n], fix[:n], rconst[:n])
