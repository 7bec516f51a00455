import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

MECHS = ("gas", "aer", "tot")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Captured reference INTEGRATE_x calls (tests/golden/make_golden.py)."""
    return {m: dict(np.load(os.path.join(REPO, "tests", "golden", "integrate_%s.npz" % m))) for m in MECHS}


# further captured sets, (set, mechanism): "day" = the BTZ96 run at model hours 6.5-8 (after sunrise: photolysis reactions
# switched on), "base1" = the reference's cloud-free namelist.base1 (no tot calls in it), "buys13" = the box-model run of
# namelist.Buys13_0D (BASELINE.json configs[0]: aerosol mechanism only)
EXTRA_SETS = [("day", "gas"), ("day", "aer"), ("day", "tot"), ("base1", "gas"), ("base1", "aer"), ("buys13", "aer")]


def load_golden(mech, suffix=""):
    return dict(np.load(os.path.join(REPO, "tests", "golden", "integrate_%s%s.npz" % (mech, suffix))))


@pytest.fixture(scope="session")
def oracles():
    from oracle.oracle import Oracle
    return {m: Oracle(m) for m in MECHS}


def rel_diff(got, want, floor_rel=1e-12):
    """max over species of |got-want| / (|want| + floor), floor = floor_rel * largest concentration of the cell"""
    got, want = np.atleast_2d(got), np.atleast_2d(want)
    floor = floor_rel * np.abs(want).max(axis=1, keepdims=True)
    return np.abs(got - want) / (np.abs(want) + floor)
