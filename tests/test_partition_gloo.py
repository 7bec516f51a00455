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


def test_shard_covers_everything():
    from mistra_amd.workload import shard
    for total in (0, 1, 7, 100000, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(n for _, n in spans) == total
            for (s0, n0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + n0 == s1
