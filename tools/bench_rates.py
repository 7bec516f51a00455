"""Update_RCONST_x on the device, timed: 1e5 cells per mechanism with inputs resident in HBM, the compiled
reference's update_rconst_x_ (oracle/_ref, one host core) beside it on a sample.  GPU box: python tools/bench_rates.py"""
import json, os, sys, time
import numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
import torch
from mistra_amd import chem
from mistra_amd.chem import DIMS

chem.init(0)
dev = torch.device('cuda', 0)
n = 100000
for mech in ('gas', 'aer', 'tot'):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'rates_%s.npz' % mech))
    base = torch.tensor(g['env'], device=dev)
    env = base[torch.arange(n, device=dev) % base.shape[0]].contiguous()
    out = chem.update_rconst(mech, env)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    ev0.record()
    for _ in range(reps):
        out = chem.update_rconst(mech, env)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    nreact = DIMS[mech][2]
    gb = n * 8 * (env.shape[1] + nreact) / 1e9
    line = '%s: %d cells in %.3f ms = %.1f M cells/s; %.2f GB in + out = %.0f GB/s (%.1f %% of 8 TB/s)' % (
        mech, n, ms, n / ms / 1e3, gb, gb / (ms / 1e3), 100 * gb / (ms / 1e3) / 8000)
    try:
        from oracle.oracle import Reference
        names = json.load(open(os.path.join(ROOT, 'mistra_amd', 'mech', mech + '.rates_env.json')))['env']
        ref = Reference(mech)
        t0 = time.time(); k = 0
        while time.time() - t0 < 2.0:
            ref.update_rconst(names, g['env'][k % g['env'].shape[0]]); k += 1
        line += '; compiled reference incl. the ctypes staging of its COMMON blocks: %.0f cells/s on one host core' % (k / (time.time() - t0))
    except Exception as e:       # no compiled reference on this box
        line += '; (no compiled reference here: %s)' % type(e).__name__
    print(line, flush=True)
