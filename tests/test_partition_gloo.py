"""N>1 path on CPU: two gloo ranks shard the synthetic workload exactly as bench.py does, integrate their shards
(with the oracle standing in for the GPU, this being the CPU suite) and the union equals the single-rank batch."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, ncell, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mistra_amd.workload import make_batch, shard
    from oracle.oracle import Oracle
    start, n = shard(ncell, rank, world)
    var, fix, rconst = make_batch("gas", start, n, "cpu")
    out, ierr, st = Oracle("gas").integrate_batch(var.numpy(), fix.numpy(), rconst.numpy())
    agg = torch.tensor([float(st[:, 2].sum()), float((ierr != 1).sum()), float(n)], dtype=torch.float64)
    dist.all_reduce(agg)                                   # the only collective bench.py uses, outside the timed region
    gathered = [None] * world
    dist.all_gather_object(gathered, (start, out))
    if rank == 0:
        q.put((agg.tolist(), gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partition_matches_single_rank():
    from mistra_amd.workload import make_batch, shard
    from oracle.oracle import Oracle
    ncell, world = 37, 2
    assert shard(ncell, 0, world) == (0, 19) and shard(ncell, 1, world) == (19, 18)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, ncell, q)) for r in range(world)]
    for p in procs:
        p.start()
    agg, gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    var, fix, rconst = make_batch("gas", 0, ncell, "cpu")
    want, ierr, st = Oracle("gas").integrate_batch(var.numpy(), fix.numpy(), rconst.numpy())
    got = np.concatenate([o for _, o in sorted(gathered, key=lambda x: x[0])])
    assert np.array_equal(got, want)
    assert agg == [float(st[:, 2].sum()), 0.0, float(ncell)]


class _OracleEngine:
    """Stand-in for bench.GpuEngine in the CPU suite: same interface, the oracle instead of the HIP library (test-only; the
    product has no CPU path and bench.py itself only ever builds GpuEngine)."""

    def __init__(self):
        from oracle.oracle import Oracle
        self.oracles = {m: Oracle(m) for m in ("gas",)}

    def integrate_into(self, mech, var, fix, rconst, out, ierr, stats):
        o, e, st = self.oracles[mech].integrate_batch(var.numpy(), fix.numpy(), rconst.numpy())
        out.copy_(torch.from_numpy(o))
        ierr.copy_(torch.from_numpy(e.astype(np.int32)))
        stats.copy_(torch.from_numpy(st.astype(np.int32)))

    def synchronize(self):
        pass

    def event(self):
        import time
        return time.perf_counter()

    @staticmethod
    def elapsed_ms(a, b):
        return 1e3 * (b - a)


def _bench_rank_main(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import argparse
    import bench
    args = argparse.Namespace(gpus=world, steps=2, warmup=1, cells_per_gpu=9, mech="gas", backend="gloo", share_device=False,
                              no_cpu_baseline=True)
    line = bench.rank_body(args, rank, world, torch.device("cpu"), _OracleEngine())
    if rank == 0:
        q.put(line)
    dist.barrier()
    dist.destroy_process_group()


def test_bench_rank_logic_on_two_gloo_ranks():
    """bench.py's own rank code — shard, timed loop between barriers, max/sum reductions, the broadcast-integrate-gather leg —
    on two CPU ranks over gloo, with the oracle standing in for the GPU."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    line = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2
    assert line["config"]["cells_per_gpu"] == 9 and line["config"]["failed_cells"] == 0
    assert line["value"] > 0 and abs(line["value"] - 18 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
    io = line["config"]["root_io_path"]
    assert io["cells_ok_at_root"] == 18 and io["value"] > 0          # every shard's results arrived at rank 0
    for key in ("roofline",):
        assert key in line


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts torch.distributed.run itself, as a child process."""
    import bench
    seen = {}

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        class R:
            returncode = 0
        return R()
    import subprocess
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    assert bench.self_launch(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_shard_covers_everything():
    from mistra_amd.workload import shard
    for total in (0, 1, 7, 100000, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(n for _, n in spans) == total
            for (s0, n0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + n0 == s1
