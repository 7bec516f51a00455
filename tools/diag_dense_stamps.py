"""Diagnostic: where wave 0 spends its cycles inside dense_lu (library built by `tools/diag_dense.sh stamps`)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag import use_diag_lib
use_diag_lib(os.environ.get('DIAG_LIB', 'libdiag_stamps.so'))
import numpy as np
from mistra_amd import chem
from mistra_amd.workload import make_batch
chem.init(0)
var, fix, rconst = make_batch('tot', 0, 512, 'cpu')
res = chem.integrate('tot', var.numpy(), fix.numpy(), rconst.numpy())
out = (C.c_ulonglong * 16)()
assert chem.lib().mistra_diag_dense_stamps(out, 1) == 0
calls, panels = out[9], out[8]
names = {6: 'table loads land', 7: 'LU program (VM)', 10: 'scaling pass', 0: 'load+schur', 1: 'scale L', 2: 'publish+barrier', 3: 'chain (wave 0)', 4: 'barrier 2', 5: 'mfma update', 12: 'panel loop', 11: 'whole function (+ a final barrier)'}
print('dense_lu calls', calls, 'panels', panels)
for k, n in names.items():
    print('%-18s %8.0f cycles per call' % (n, out[k] / calls), '' if k in (0, 1, 6, 7, 10, 11, 12) else '(%.0f per panel)' % (out[k] / panels))
