"""Synthetic cell batches for the benchmark and the large-size tests.

Base states are real chemistry states captured from the running reference model (mistra_amd/data/base_<mech>.npz: VAR, FIX,
RCONST of INTEGRATE_x calls in the cloudy layers of namelist.BTZ96 — the inputs of tests/golden/integrate_<mech>.npz, written
next to them by tests/golden/make_golden.py).  Cell c of a batch takes base state
c mod nbase and perturbs it with a counter-based generator keyed by (seed, c, index):

    VAR_i    <- VAR_i    * exp(sigma * g),   g ~ N(0,1)       sigma   = 0.10
    RCONST_j <- RCONST_j * (1 + eps * u),    u ~ U(-1,1)      eps     = 0.05
    FIX unchanged

Everything is torch integer/float ops, so a batch is generated directly in HBM on the rank's GPU (inputs never cross
PCIe) and the same code runs on CPU tensors in the gloo tests.  Arbitrary concentrations make the stiff system
explode, hence perturbations of spun-up states.
"""
import os

import numpy as np
import torch

SEED = 20260101
SIGMA = 0.10
EPS = 0.05
DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

_M1 = -7046029254386353131      # 0x9E3779B97F4A7C15 as int64
_M2 = -4658895280553007687      # 0xBF58476D1CE4E5B9
_M3 = -7723592293110705685      # 0x94D049BB133111EB


def _lsr(x, k):
    """logical shift right on int64 tensors"""
    return (x >> k) & ((1 << (64 - k)) - 1)


def _mix(x):
    """splitmix64 finaliser (wrapping int64 arithmetic)"""
    x = (x ^ _lsr(x, 30)) * _M2
    x = (x ^ _lsr(x, 27)) * _M3
    return x ^ _lsr(x, 31)


def _uniform(seed, cell, index, stream):
    """U(0,1) double, strictly inside the interval, keyed by (seed, cell, index, stream)"""
    key = _mix(_mix(cell * _M1 + seed) + index * _M1 + stream)
    return (_lsr(key, 11).to(torch.float64) + 0.5) * (1.0 / 9007199254740992.0)


def load_base(mech):
    """(var, fix, rconst) numpy arrays [nbase, .] of the captured reference states"""
    z = np.load(os.path.join(DATA_DIR, "base_%s.npz" % mech))
    return z["var"], z["fix"], z["rconst"]


def make_batch(mech, cell_start, ncell, device, seed=SEED, sigma=SIGMA, eps=EPS, base=None):
    """Cells [cell_start, cell_start+ncell) of the synthetic workload as float64 tensors on `device`:
    (var [ncell,NVAR], fix [ncell,NFIX], rconst [ncell,NREACT])."""
    bv, bf, br = base if base is not None else load_base(mech)
    bv = torch.as_tensor(bv, dtype=torch.float64, device=device)
    bf = torch.as_tensor(bf, dtype=torch.float64, device=device)
    br = torch.as_tensor(br, dtype=torch.float64, device=device)
    nbase, nvar = bv.shape
    nreact = br.shape[1]
    cell = torch.arange(cell_start, cell_start + ncell, dtype=torch.int64, device=device)
    which = cell % nbase
    var = torch.empty((ncell, nvar), dtype=torch.float64, device=device)
    rconst = torch.empty((ncell, nreact), dtype=torch.float64, device=device)
    chunk = max(1, (1 << 24) // max(nvar, nreact))      # bound temporaries to ~128 MiB per tensor
    iv = torch.arange(nvar, dtype=torch.int64, device=device)[None, :]
    ir = torch.arange(nreact, dtype=torch.int64, device=device)[None, :]
    for lo in range(0, ncell, chunk):
        hi = min(ncell, lo + chunk)
        c = cell[lo:hi, None]
        u1 = _uniform(seed, c, iv, 1)
        u2 = _uniform(seed, c, iv, 2)
        g = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(6.283185307179586 * u2)          # Box-Muller
        var[lo:hi] = bv[which[lo:hi]] * torch.exp(sigma * g)
        u = 2.0 * _uniform(seed, c, ir, 3) - 1.0
        rconst[lo:hi] = br[which[lo:hi]] * (1.0 + eps * u)
    fix = bf[which].contiguous()
    return var, fix, rconst


def shard(ncell_total, rank, world):
    """Contiguous block partition of cells over ranks (cells are independent: kpp.f90:4310-4470)."""
    per = ncell_total // world
    rem = ncell_total % world
    start = rank * per + min(rank, rem)
    return start, per + (1 if rank < rem else 0)
