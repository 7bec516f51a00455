"""Where the LU program's time goes, by rounds: runs the profiling kernel with the LU program cut after n rounds
(MISTRA_DIAG_LU_ROUNDS, numerically meaningless) and prints LU cycles per LU call.  GPU box: python tools/profile_lu_rounds.py"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag import REPO as ROOT, diag_env
code = r'''
import os, sys
sys.path.insert(0, %r)
os.environ['MISTRA_CHEM_PROFILE'] = '1'
from mistra_amd import chem
from mistra_amd.workload import make_batch
chem.init(0)
var, fix, rconst = make_batch('tot', 0, 256, 'cpu')
res = chem.integrate('tot', var.numpy(), fix.numpy(), rconst.numpy())
print('NDEC', res.stats[:, 5].mean())
''' % ROOT
for n in [int(x) for x in sys.argv[1:]] or [1, 5, 10, 16, 17, 18, 19, 30, 45, 57, 70, 88, 89]:
    env = diag_env('libdiag_env.so', MISTRA_DIAG_LU_ROUNDS=str(n))   # tools/diag_dense.sh env
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=120)
    lu = re.search(r' lu=(\d+)', r.stderr)
    nd = re.search(r'NDEC ([\d.]+)', r.stdout)
    if lu and nd:
        print('rounds %3d  LU cycles per call %9.0f   (decompositions per cell %.1f)' % (n, int(lu.group(1)) / float(nd.group(1)), float(nd.group(1))), flush=True)
    else:
        print('rounds', n, 'failed', r.stderr[-300:])
