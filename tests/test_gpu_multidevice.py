"""Every multi-GPU code path on the ONE GPU a test box has (-m gpu; BASELINE.json configs[3] is 1e6 tot cells on 8 GPUs):

  * the in-library split of a single-process caller (mistra_chem_init_devices + mistra_chem_integrate_ex: contiguous blocks of
    cells, one host thread per device slot, capi.cpp) — device 0 listed two and three times, so the block arithmetic, the
    per-block offsets of ierr / stats / t_h and the threads all run — must give bit for bit what one device gives;
  * bench.py's rank code with the real engine: two gloo ranks sharing the GPU (`--backend gloo --share-device`), shard,
    timed loop, reductions and the broadcast / integrate / gather leg; the union of the ranks' results at rank 0 must be
    bit for bit what one process computes for the whole batch.

Cells are independent (kpp.f90:4310-4470: the layer loop carries nothing from one k to the next), so both are plain equalities."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


@pytest.fixture()
def chem():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mistra_amd import chem as c
    yield c
    c.finalize()          # whatever a test initialised: the other modules' fixtures start from init(0)
    c.init(0)


def _column_cells(mech):
    col = np.load(os.path.join(REPO, "tests", "golden", "column_BTZ96.npz"))
    return col[mech + "_var_in"], col[mech + "_fix"], col[mech + "_rconst"], col[mech + "_stats"]


@pytest.mark.parametrize("mech", ["tot", "aer", "gas"])
def test_in_library_device_split_matches_one_device(chem, mech):
    """A captured column step (namelist.BTZ96, chem=T: 49 tot, 31 aer, 68 gas layers) through mistra_chem_integrate_ex."""
    var, fix, rconst, ref_stats = _column_cells(mech)
    n = var.shape[0]
    assert n >= 7
    chem.finalize()
    chem.init_devices([0])
    assert chem.device_count() == 1
    one, th_one = chem.integrate_ex(mech, var, fix, rconst, 0.0, 10.0)
    assert np.all(one.ierr == 1) and np.array_equal(one.stats, ref_stats)          # the reference's /Statistics/ of these calls
    for listing in ([0, 0], [0, 0, 0]):
        chem.finalize()
        chem.init_devices(listing)
        assert chem.device_count() == len(listing)
        got, th = chem.integrate_ex(mech, var, fix, rconst, 0.0, 10.0)
        assert np.array_equal(got.var, one.var), "VAR differs between the split and the one-device call"
        assert np.array_equal(got.ierr, one.ierr) and np.array_equal(got.stats, one.stats) and np.array_equal(th, th_one)
        # ragged split: a cell count that is not a multiple of the slot count, and one below 2 per slot (falls back to one slot)
        for m in (n - 1, len(listing) * 2 - 1):
            sub, th_sub = chem.integrate_ex(mech, var[:m], fix[:m], rconst[:m], 0.0, 10.0)
            assert np.array_equal(sub.var, one.var[:m]) and np.array_equal(sub.stats, one.stats[:m]) and np.array_equal(th_sub, th_one[:m])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_share_the_gpu_union_matches_single_rank(chem, tmp_path):
    """bench.rank_body with GpuEngine on two gloo ranks (one GPU box: both on cuda:0), root_io_path included."""
    import torch
    cells = 96          # per rank
    dump = tmp_path / "root_io.npz"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
           "--cells-per-gpu", str(cells), "--steps", "1", "--warmup", "0", "--mech", "tot", "--dump-root-io", str(dump)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["failed_cells"] == 0
    io = line["config"]["root_io_path"]
    assert io["cells_ok_at_root"] == 2 * cells
    z = np.load(dump)
    # the same 192 cells in ONE process
    from mistra_amd.workload import make_batch
    chem.init(0)
    var, fix, rconst = make_batch("tot", 0, 2 * cells, torch.device("cuda", 0))
    res = chem.integrate("tot", var, fix, rconst)
    torch.cuda.synchronize()
    assert np.array_equal(z["var_out"], res.var.cpu().numpy()), "union of the two ranks' results differs from the single-rank batch"
    assert np.array_equal(z["ierr"], res.ierr.cpu().numpy()) and np.array_equal(z["stats"], res.stats.cpu().numpy())
    assert abs(line["config"]["mean_internal_steps_per_cell"] - float(res.stats[:, 2].double().mean())) < 1e-9
