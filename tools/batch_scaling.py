"""Throughput against batch size (device-resident inputs, one launch): how many cells it takes to fill one MI355X.
GPU box: python tools/batch_scaling.py [mech ...]"""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from mistra_amd import chem
from mistra_amd.workload import make_batch

chem.init(0)
dev = torch.device('cuda', 0)
for mech in (sys.argv[1:] or ['tot', 'aer', 'gas']):
    big = {'tot': 16384, 'aer': 32768, 'gas': 262144}[mech]
    var, fix, rconst = make_batch(mech, 0, big, dev)
    sizes = [s for s in (1, 16, 64, 148, 256, 512, 1024, 2048, 4096, 16384, 65536, 262144) if s <= big]
    for n in sizes:
        v, f, r = var[:n].contiguous(), fix[:n].contiguous(), rconst[:n].contiguous()
        chem.integrate(mech, v, f, r); torch.cuda.synchronize()
        reps = 3 if n >= 4096 else 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            chem.integrate(mech, v, f, r)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print('%s %7d cells: %9.3f ms per launch  %10.0f timesteps/s' % (mech, n, ms, n / ms * 1e3), flush=True)
