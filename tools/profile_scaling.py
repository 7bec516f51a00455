"""Diagnostic: per-cell phase ticks of the tot kernel with 1, 32 and 512 cells in flight (is the per-cell time set by
the cell's own dependency chains, or by contention for the shared table stream?)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag import use_diag_lib
use_diag_lib('libdiag_env.so', MISTRA_CHEM_PROFILE='1')      # the product library has no profiling switch
from mistra_amd import chem
from mistra_amd.workload import make_batch
chem.init(0)
var, fix, rconst = (x.numpy() for x in make_batch('tot', 0, 512, 'cpu'))
for n in (1, 32, 256, 512):
    print('cells', n, flush=True)
    chem.integrate('tot', var[:n], fix[:n], rconst[:n])
