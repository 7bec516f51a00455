"""Tolerance study of the OPT-IN Hstart-reuse mode (include/mistra_chem.h: mistra_chem_integrate_device_hstart).

INTEGRATE_x restarts every 10-s call at H = 1e-3 s (gas.f:743) and climbs back to the step size the chemistry allows.
Here NCALLS consecutive chemistry timesteps are run on a batch of synthetic tot cells (rate constants frozen) twice:
  A  as the reference does it: every call from Hstart = 1e-3
  B  every call from the cell's last step size of the previous call (texit_hexit[:, 1])
and both are measured against a tight-tolerance solution of the same 10*NCALLS seconds (the oracle at RTOL 1e-7, ONE call,
on a sample of the cells — test infrastructure standing in as the yardstick, CPU).

    python tools/hstart_study.py [ncells=2048] [ncalls=30] [nsample=12]         (GPU box)
"""
import os
import sys
import time

import numpy as np

REPO = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
nsample = int(sys.argv[3]) if len(sys.argv) > 3 else 12

# the yardstick first, in worker processes forked before this process touches the GPU
import multiprocessing as mp
from mistra_amd.workload import make_batch


def truth_worker(job):
    from oracle.oracle import Oracle, set_options
    var, fix, rconst, tend = job
    set_options(rtol=1e-7)
    out, ierr, st = Oracle("tot").integrate_batch(var, fix, rconst, 0.0, tend)
    return out, st[:, 2]


v, f, r = (x.numpy() for x in make_batch("tot", 0, nsample, "cpu"))
t0 = time.time()
with mp.get_context("fork").Pool(min(nsample, len(os.sched_getaffinity(0)))) as pool:
    res = pool.map(truth_worker, [(v[i:i + 1], f[i:i + 1], r[i:i + 1], 10.0 * ncalls) for i in range(nsample)])
truth = np.concatenate([x[0] for x in res])
print("yardstick: oracle at RTOL 1e-7, %d cells, one call over %.0f s: %.0f steps per cell, %.0f s of CPU wall" %
      (nsample, 10.0 * ncalls, np.mean([x[1][0] for x in res]), time.time() - t0), flush=True)

import torch
from mistra_amd import chem

dev = torch.device("cuda", 0)
chem.init(0)
var0, fix, rconst = make_batch("tot", 0, ncell, dev)


def run(reuse):
    var = var0.clone()
    out = torch.empty_like(var)
    ierr = torch.empty(ncell, dtype=torch.int32, device=dev)
    stats = torch.empty((ncell, 8), dtype=torch.int32, device=dev)
    th = torch.zeros((ncell, 2), dtype=torch.float64, device=dev)
    hstart = None
    steps, fails = [], 0
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(ncalls):
        chem.integrate_into("tot", var, fix, rconst, out, ierr, stats, 0.0, 10.0, texit_hexit=th, hstart=hstart)
        var, out = out, var
        steps.append(float(stats[:, 2].double().mean().item()))
        fails += int((ierr != 1).sum().item())
        if reuse:
            hstart = th[:, 1].contiguous().clone()
    torch.cuda.synchronize()
    return var.cpu().numpy(), steps, time.time() - t0, fails


def rel(a, b):
    floor = 1e-12 * np.abs(b).max(axis=1, keepdims=True)
    return np.abs(a - b) / (np.abs(b) + floor)


a, sa, ta, fa = run(False)
b, sb, tb, fb = run(True)
print("tot, %d cells, %d consecutive 10-s calls, rate constants frozen" % (ncell, ncalls))
print("  A  Hstart = 1e-3 every call (the reference): %6.1f steps per call and cell (first call %.1f), %.2f s GPU wall, %d failed" % (np.mean(sa), sa[0], ta, fa))
print("  B  Hstart = last step of the previous call : %6.1f steps per call and cell (first call %.1f), %.2f s GPU wall, %d failed" % (np.mean(sb), sb[0], tb, fb))
print("  steps %.2fx fewer, wall %.2fx shorter" % (np.mean(sa) / np.mean(sb), ta / tb))
d = rel(b, a)
major = np.abs(a) >= 1e-4 * np.abs(a).max(axis=1, keepdims=True)
print("  B against A after %d calls: max rel diff %.2e over all species, %.2e over major species (>= 1e-4 of the cell maximum), median of per-cell max %.2e"
      % (ncalls, d.max(), np.where(major, d, 0).max(), np.median(d.max(axis=1))))
ea, eb = rel(a[:nsample], truth), rel(b[:nsample], truth)
mj = np.abs(truth) >= 1e-4 * np.abs(truth).max(axis=1, keepdims=True)
print("  against the RTOL 1e-7 solution (%d cells): A max %.2e (major species %.2e), B max %.2e (major species %.2e); RTOL of the integrator is 1e-3"
      % (nsample, ea.max(), np.where(mj, ea, 0).max(), eb.max(), np.where(mj, eb, 0).max()))
