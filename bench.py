#!/usr/bin/env python3
"""Headline benchmark: chemistry-timesteps/sec of the `tot` mechanism (BASELINE.json), one MI355X per rank.

A "step" is one pass of the hot path — INTEGRATE_t(0, 10 s) (tot.f:2812) — over one synthetic batch of
100 000 cells per GPU (BASELINE.json configs[2]; mistra_amd/workload.py builds it from captured reference states,
directly in HBM).  `value` = cells integrated by all ranks / wall time, inputs resident before the clock starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--cells-per-gpu C] [--mech tot]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started plainly with --gpus N > 1 (no WORLD_SIZE in the environment), this process starts that torch.distributed.run
command itself as a child — before it has touched the GPU — relays rank 0's JSON line and exits with the child's code.

Cells are independent, so ranks share nothing on the data path (no collective inside the timed region; weak scaling).
The data path BASELINE.json's north_star names for a single-process caller — base states broadcast from rank 0, every rank
integrating its shard, results gathered to rank 0 — is timed as a second leg at N > 1 and reported beside the headline
value (`config.root_io_path`), never as `value`.
Rank 0 at N=1 also times the CPU path on the host cores on a bounded sample of the same workload (`cpu_baseline`):
the compiled reference itself (oracle/_ref/libmistra_ref.so, kind "reference") when it was built, else the plain-C
restatement (oracle/kpp_ros3.c, kind "port").  The oracle is only ever the thing compared against / timed beside.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6
ALG_BYTES = {"gas": 4304, "aer": 11984, "tot": 19744}        # 8*(2*NVAR+NFIX+NREACT), SURVEY.md §8d


def flop_per_step(mech):
    """per internal Ros3 step, counted from the mechanism tables with SURVEY.md §8d's accounting (mistra_amd/mechtab.py: flop_counts):
    gas 23 527, aer 213 354, tot 569 818"""
    from mistra_amd.mechtab import flop_counts
    return float(flop_counts(mech)["step"])


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_worker(job):
    """Runs in a forked worker BEFORE the parent touches the GPU.  Integrates its share of the sample on one core."""
    kind, mech, var, fix, rconst = job
    from oracle.oracle import Oracle, Reference
    t0 = time.perf_counter()
    nstp = 0
    if kind == "reference":
        ref = Reference(mech)
        for c in range(var.shape[0]):
            _, st, _, _ = ref.integrate(var[c], fix[c], rconst[c], 0.0, 10.0)
            nstp += int(st[2])
    else:
        _, _, st = Oracle(mech).integrate_batch(var, fix, rconst, 0.0, 10.0)
        nstp = int(st[:, 2].sum())
    return time.perf_counter() - t0, nstp


def _parity_worker(job):
    """Forked before the parent touched the GPU: the CPU checker's answer for a few cells (inputs exactly as the GPU saw them)."""
    kind, mech, var, fix, rconst = job
    import numpy as np
    from oracle.oracle import Oracle, Reference
    if kind == "reference":
        ref = Reference(mech)
        out, stats = np.empty_like(var), np.zeros((var.shape[0], 8), np.int32)
        for c in range(var.shape[0]):
            out[c], stats[c], _, _ = ref.integrate(var[c], fix[c], rconst[c], 0.0, 10.0)
        return out, stats
    out, _, stats = Oracle(mech).integrate_batch(var, fix, rconst, 0.0, 10.0)
    return out, stats


class ParityChecker:
    """The second half of BASELINE.json's metric ("... ; max |dc|/|c| vs ref") for the workload that was just timed: a fixed
    strided sample of the batch, integrated by the CPU checker (the compiled reference where oracle/_ref holds it, else the
    plain-C restatement) in worker processes that were forked BEFORE this process initialised the GPU, compared with what the
    kernel left in `out`.  The checker is only ever the thing compared against."""
    CELLS = 256

    def __init__(self, mech):
        import multiprocessing as mp
        from oracle.oracle import Reference, build_oracle
        self.mech = mech
        self.kind = "reference" if Reference.available() else "port"
        if self.kind == "port":
            build_oracle()
        self.workers = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("MISTRA_BENCH_CPU_CORES", "16"))))
        self.pool = mp.get_context("fork").Pool(self.workers)

    def check(self, var, fix, rconst, out, stats):
        import numpy as np
        import torch
        n = var.shape[0]
        k = min(self.CELLS, n)
        sel = torch.arange(k, device=var.device, dtype=torch.int64) * (n // k)
        v, f, r = (x[sel].cpu().numpy() for x in (var, fix, rconst))
        got, got_stats = out[sel].cpu().numpy(), stats[sel].cpu().numpy()
        per = -(-k // self.workers)
        jobs = [(self.kind, self.mech, v[i:i + per], f[i:i + per], r[i:i + per]) for i in range(0, k, per)]
        res = self.pool.map(_parity_worker, jobs)
        self.pool.close()
        want = np.concatenate([x[0] for x in res])
        want_stats = np.concatenate([x[1] for x in res])
        floor = 1e-12 * np.abs(want).max(axis=1, keepdims=True)
        rel = np.abs(got - want) / (np.abs(want) + floor)
        major = np.abs(want) >= 1e-4 * np.abs(want).max(axis=1, keepdims=True)
        return {"max_rel_all": float(rel.max()), "max_rel_major": float(np.where(major, rel, 0.0).max()),
                "stats_identical": bool(np.array_equal(got_stats, want_stats)), "cells": int(k), "against": self.kind,
                "what": "every %d-th cell of the timed batch against the CPU %s on the inputs the GPU saw: max over cells and species of "
                        "|dc| / (|c| + 1e-12 max|c| of the cell); major = species above 1e-4 of the cell maximum; stats = COMMON /Statistics/ "
                        "(Nfun Njac Nstp Nacc Nrej Ndec Nsol Nsng) of every sampled cell" % (n // k, "reference (oracle/_ref)" if self.kind == "reference" else "restatement (oracle/kpp_ros3.c)")}


def kernel_source_hash():
    """Identifies the kernel a PMC pass was taken on: sha256 over the device code and the schedule compiler."""
    import hashlib
    h = hashlib.sha256()
    for f in ("ros3_kernel.hip", "ros3_kernel.hpp", "kernel_args.hpp", "schedule.cpp", "schedule.hpp"):
        h.update(open(os.path.join(REPO, "mistra_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(mech, ncell):
    """HBM bytes per launch from the most recent committed PMC pass (profiles/rNN_traffic.json), scaled to this launch's
    cell count.  bench.py cannot collect PMC counters itself (they need their own rocprofv3 pass), so a pass is only used
    while it was taken on THIS kernel: the file records the kernel source hash, and a stale or missing pass gives null."""
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_traffic*.json")), reverse=True):
        d = json.load(open(path))
        if d.get("mech") == mech:
            if d.get("kernel_source_hash") != kernel_source_hash():
                return None, os.path.basename(path) + " is stale (taken on another kernel): not used"
            return d["bytes_per_launch"] * ncell / d["cells"], os.path.basename(path)
    return None, None


def cpu_baseline(mech, budget_s=15.0):
    import multiprocessing as mp
    import numpy as np
    import torch
    from oracle.oracle import Reference, build_oracle
    from mistra_amd.workload import make_batch
    kind = "reference" if Reference.available() else "port"
    if kind == "port":
        build_oracle()
    cores_available = len(os.sched_getaffinity(0))
    # One process per core (the reference is serial).  A one-GPU box of the pool shows the whole host in its affinity mask
    # (256 logical CPUs) but grants a job the CPU share of ONE GPU, 16 cores: forking 256 workers there stalls the run, so the
    # baseline uses min(affinity, MISTRA_BENCH_CPU_CORES or 16) and reports both numbers.
    cores = max(1, min(cores_available, int(os.environ.get("MISTRA_BENCH_CPU_CORES", "16"))))
    # calibrate on a few cells, then size the sample for ~budget_s seconds on all cores
    var, fix, rconst = (x.numpy() for x in make_batch(mech, 0, 4, "cpu"))
    dt, _ = _cpu_worker((kind, mech, var, fix, rconst))
    per_cell = dt / 4
    ncell = int(max(cores, min(262144, budget_s * cores / max(per_cell, 1e-6))))
    ncell -= ncell % cores
    var, fix, rconst = (x.numpy() for x in make_batch(mech, 0, ncell, "cpu"))
    share = ncell // cores
    jobs = [(kind, mech, var[i * share:(i + 1) * share], fix[i * share:(i + 1) * share], rconst[i * share:(i + 1) * share])
            for i in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    nstp = sum(r[1] for r in res)
    return {"value": ncell / wall, "unit": "chemistry-timesteps/s", "cores": cores, "cores_available": cores_available, "kind": kind,
            "extrapolated_all_cores": {"value": ncell / wall / cores * cores_available,
                                       "what": "per-core rate x cores_available (%d): NOT measured — a one-GPU job is granted %d cores; "
                                               "an upper bound that assumes the serial reference scales linearly over every logical CPU of the host"
                                               % (cores_available, cores)},
            "sample": "%d cells of the same synthetic %s workload (cells 0..%d), %d processes x 1 thread, %.1f s wall, "
                      "%.1f internal steps/cell, %.0f timesteps/s/core" % (ncell, mech, ncell - 1, cores, wall,
                                                                         nstp / ncell, ncell / wall / cores)}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run the N ranks as a child torch.distributed.run (one process per GPU,
    127.0.0.1 rendezvous on a free port) and pass its output through.  A child process, never an exec: this one stays a
    plain Python process that has made no GPU call."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def root_io_leg(args, dist, dev, world, rank, engine, make_batch, shard):
    """The data path of a single-process caller (north_star: "RCCL broadcast/gather over xGMI only for the embarrassingly-parallel
    cell partition"): rank 0 holds the base states, broadcasts them, every rank builds and integrates its
    shard, VAR_out / ierr / stats of all shards are gathered on rank 0.  Timed with the same barrier + synchronize + max over
    ranks as the headline leg; returns whole-job timesteps/s including the collectives, or None at N = 1."""
    import torch
    from mistra_amd.workload import load_base
    if world == 1:
        return None
    on_dev = args.backend == "nccl"
    cdev = dev if on_dev else torch.device("cpu")
    shapes = {"gas": (102, 3, 331), "aer": (257, 5, 979), "tot": (417, 7, 1627)}[args.mech]
    if rank == 0:
        base = [torch.as_tensor(x, dtype=torch.float64, device=cdev).contiguous() for x in load_base(args.mech)]
        nbase = torch.tensor([base[0].shape[0]], dtype=torch.int64, device=cdev)
    else:
        nbase = torch.zeros(1, dtype=torch.int64, device=cdev)
    total = args.cells_per_gpu * world
    start, ncell = shard(total, rank, world)
    engine.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    dist.broadcast(nbase, 0)
    if rank != 0:
        base = [torch.empty((int(nbase.item()), n), dtype=torch.float64, device=cdev) for n in shapes]
    for x in base:
        dist.broadcast(x, 0)                      # ncclBroadcast of the base states (tables travel with the library itself)
    var, fix, rconst = make_batch(args.mech, start, ncell, dev, base=[x.to(dev) for x in base])
    out = torch.empty_like(var)
    ierr = torch.empty(ncell, dtype=torch.int32, device=dev)
    stats = torch.empty((ncell, 8), dtype=torch.int32, device=dev)
    engine.integrate_into(args.mech, var, fix, rconst, out, ierr, stats)
    engine.synchronize()
    # gather to the root by direct sends (xGMI is point-to-point): every rank ships its block of results to rank 0
    packed = torch.cat([out, ierr.double()[:, None], stats.double()], dim=1).contiguous().to(cdev)
    if rank == 0:
        gathered = [packed]
        reqs = []
        for r in range(1, world):
            _, n_r = shard(total, r, world)
            buf = torch.empty((n_r, packed.shape[1]), dtype=torch.float64, device=cdev)
            gathered.append(buf)
            reqs.append(dist.irecv(buf, src=r))
        for q in reqs:
            q.wait()
        ok = int(sum(int((g[:, shapes[0]] == 1).sum().item()) for g in gathered))
        if getattr(args, "dump_root_io", None):
            import numpy as np
            allr = torch.cat([g.cpu() for g in gathered]).numpy()
            np.savez(args.dump_root_io, var_out=allr[:, :shapes[0]], ierr=allr[:, shapes[0]].astype(np.int32),
                     stats=allr[:, shapes[0] + 1:].astype(np.int32))
    else:
        dist.send(packed, dst=0)
        ok = 0
    engine.synchronize()
    dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    return {"value": total / float(elapsed.item()), "unit": "chemistry-timesteps/s", "seconds": float(elapsed.item()),
            "cells_ok_at_root": ok, "what": "broadcast of the base states from rank 0, shard generation, one integration per rank, "
            "gather of VAR_out + ierr + stats to rank 0 by direct sends; one pass, collectives inside the clock"}


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cells-per-gpu", type=int, default=100000)
    ap.add_argument("--mech", default="tot", choices=["gas", "aer", "tot"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N>1 (nccl = RCCL; gloo only for rehearsing the rank logic)")
    ap.add_argument("--dump-root-io", default=None,
                    help="tests only: rank 0 writes what the root-I/O leg gathered (VAR_out, ierr, stats of all shards) to this .npz")
    ap.add_argument("--no-parity", action="store_true", help="skip the post-run parity sample against the CPU checker")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary aer / gas legs of the default (tot, one GPU) run")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (requires --backend gloo)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))          # nothing in this process has touched the GPU yet
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    cpu, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.mech)          # before any GPU initialisation in this process (forks workers)
    if rank == 0 and world == 1 and not args.no_parity:
        parity = ParityChecker(args.mech)      # idle worker processes, forked before any GPU initialisation

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product has no CPU path")
    if args.share_device:
        if args.backend != "gloo":
            sys.exit("--share-device needs --backend gloo (RCCL refuses two ranks on one GPU)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    line = rank_body(args, rank, world, dev, GpuEngine(local_rank), parity)
    if rank == 0:
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


class GpuEngine:
    """The product on one GPU: the HIP library through mistra_amd.chem, torch for memory, streams and events."""

    def __init__(self, local_rank):
        import torch
        from mistra_amd import chem
        self.torch, self.chem = torch, chem
        chem.init(local_rank)

    def integrate_into(self, mech, var, fix, rconst, out, ierr, stats):
        self.chem.integrate_into(mech, var, fix, rconst, out, ierr, stats)

    def synchronize(self):
        self.torch.cuda.synchronize()

    def event(self):
        e = self.torch.cuda.Event(enable_timing=True)
        e.record()       # same stream as the kernel (chem.integrate_into launches on torch's current stream)
        return e

    @staticmethod
    def elapsed_ms(a, b):
        return a.elapsed_time(b)


def rank_body(args, rank, world, dev, engine, parity=None):
    """What one rank does (the process group, if any, is up): build its shard in place, warm up, time `steps` passes between
    barrier + synchronize on both sides, reduce (max of the time, sums of the counters), run the root-I/O leg; returns the
    JSON line on rank 0.  `engine` is the product (GpuEngine); tests/test_partition_gloo.py drives this same function on two
    CPU ranks with a stand-in engine."""
    import torch
    import torch.distributed as dist
    from mistra_amd.workload import make_batch, shard
    total_cells = args.cells_per_gpu * world
    start, ncell = shard(total_cells, rank, world)
    var, fix, rconst = make_batch(args.mech, start, ncell, dev)
    out = torch.empty_like(var)
    ierr = torch.empty(ncell, dtype=torch.int32, device=dev)
    stats = torch.empty((ncell, 8), dtype=torch.int32, device=dev)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        engine.integrate_into(args.mech, var, fix, rconst, out, ierr, stats)
    engine.synchronize()
    barrier()
    ev = []
    engine.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        e0 = engine.event()
        engine.integrate_into(args.mech, var, fix, rconst, out, ierr, stats)
        ev.append((e0, engine.event()))
    engine.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(engine.elapsed_ms(a, b) for a, b in ev) / max(1, args.steps)

    cdev = dev if args.backend == "nccl" else torch.device("cpu")      # gloo reduces host tensors
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    agg = torch.stack([stats[:, 2].sum().double(), (ierr != 1).sum().double(),
                       torch.tensor(float(ncell), device=dev, dtype=torch.float64)]).to(cdev)
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    elapsed = float(t_el.item())
    nstp_total, nfail, cells_done = (float(x) for x in agg.tolist())

    root_io = root_io_leg(args, dist, dev, world, rank, engine, make_batch, shard)
    if rank != 0:
        return None
    parity_line = parity.check(var, fix, rconst, out, stats) if parity is not None else None      # outside the timed region
    extra = None
    if world == 1 and args.mech == "tot" and not getattr(args, "no_extra", False):
        del var, fix, rconst, out
        extra = {m: secondary_leg(m, args.cells_per_gpu, dev, engine, make_batch) for m in ("aer", "gas")}
    value = total_cells * args.steps / elapsed
    steps_per_cell = nstp_total / cells_done
    achieved = ncell * ALG_BYTES[args.mech] / (kernel_ms * 1e-3) / 1e9
    flops = ncell * steps_per_cell * flop_per_step(args.mech) / (kernel_ms * 1e-3)
    traffic, traffic_src = measured_traffic(args.mech, ncell)
    return {
        "metric": "chemistry-timesteps/sec (%s mechanism)" % args.mech, "value": value, "unit": "chemistry-timesteps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(1, args.steps),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "parity": parity_line,
        # the two mechanisms every shipped chem=T namelist actually runs, outside the headline timing: 1 warm-up + 2 passes each
        "extra": extra,
        "config": {"workload": "%s mechanism, %d synthetic cells per GPU (%d total), INTEGRATE_%s(0,10 s), Ros3 rtol 1e-3; "
                               "perturbed captured BTZ96 cloud states" % (args.mech, args.cells_per_gpu, total_cells, args.mech[0]),
                   "cells_per_gpu": args.cells_per_gpu, "mean_internal_steps_per_cell": steps_per_cell,
                   "failed_cells": int(nfail), "parallelism": "cells sharded over %d GPU(s), no data-path collective" % world,
                   "root_io_path": root_io},
        # `achieved` / `frac` are the contract's figures: ALGORITHMIC bytes over the kernel's time against the HBM roof.  The path
        # is not HBM-bound: `limiter` says what the counters show instead, `traffic` what the memory side really
        # moved per launch (schedule tables re-streamed through L2 by every workgroup), `fp64_*` the arithmetic rate.
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "traffic_rate_gbs": None if traffic is None else traffic / (kernel_ms * 1e-3) / 1e9,
                     "limiter": {"tot": "latency of one dependency chain per CU (one tot cell owns a CU's LDS)",
                                 "aer": "latency of two dependency chains per CU (two aer cells per CU at 128 registers per lane)",
                                 "gas": "instruction issue and the latency of eight dependency chains per CU (eight gas cells per CU at 128 registers per lane)"}[args.mech]
                                + ": waves parked at s_waitcnt / s_barrier",
                     "kernel": "ros3_integrate_kernel", "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_cell": ALG_BYTES[args.mech],
                     "fp64_tflops": flops / 1e12, "fp64_frac_of_vector_peak": flops / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                     "flop_per_internal_step": flop_per_step(args.mech)},
    }


def secondary_leg(mech, ncell, dev, engine, make_batch, passes=2):
    """One more mechanism on the same GPU after the headline timing: INTEGRATE_x(0, 10 s) over `ncell` synthetic cells, 1 warm-up + `passes`
    timed passes (HIP events on the launch stream + wall clock around them).  Never part of `value`."""
    import torch
    var, fix, rconst = make_batch(mech, 0, ncell, dev)
    out = torch.empty_like(var)
    ierr = torch.empty(ncell, dtype=torch.int32, device=dev)
    stats = torch.empty((ncell, 8), dtype=torch.int32, device=dev)
    engine.integrate_into(mech, var, fix, rconst, out, ierr, stats)
    engine.synchronize()
    t0 = time.perf_counter()
    ev = []
    for _ in range(passes):
        e0 = engine.event()
        engine.integrate_into(mech, var, fix, rconst, out, ierr, stats)
        ev.append((e0, engine.event()))
    engine.synchronize()
    wall = time.perf_counter() - t0
    kernel_ms = sum(engine.elapsed_ms(a, b) for a, b in ev) / passes
    steps_per_cell = float(stats[:, 2].sum().double().item()) / ncell
    flops = ncell * steps_per_cell * flop_per_step(mech) / (kernel_ms * 1e-3)
    achieved = ncell * ALG_BYTES[mech] / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_src = measured_traffic(mech, ncell)
    return {"metric": "chemistry-timesteps/sec (%s mechanism)" % mech, "value": ncell * passes / wall, "unit": "chemistry-timesteps/s", "cells": ncell,
            "passes": passes, "warmup": 1, "ms_per_pass": 1e3 * wall / passes, "kernel_ms": kernel_ms, "mean_internal_steps_per_cell": steps_per_cell,
            "failed_cells": int((ierr != 1).sum().item()),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src, "algorithmic_bytes_per_cell": ALG_BYTES[mech], "fp64_tflops": flops / 1e12,
                         "fp64_frac_of_vector_peak": flops / 1e12 / FP64_VECTOR_PEAK_TFLOPS, "flop_per_internal_step": flop_per_step(mech)}}


if __name__ == "__main__":
    main()
