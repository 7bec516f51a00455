#!/usr/bin/env python3
"""Headline benchmark: chemistry-timesteps/sec of the `tot` mechanism (BASELINE.json), one MI355X per rank.

A "step" is one pass of the hot path — INTEGRATE_t(0, 10 s) (tot.f:2812) — over one synthetic batch of
100 000 cells per GPU (BASELINE.json configs[2]; mistra_amd/workload.py builds it from captured reference states,
directly in HBM).  `value` = cells integrated by all ranks / wall time, inputs resident before the clock starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--cells-per-gpu C] [--mech tot]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Cells are independent, so ranks share nothing on the data path (no collective inside the timed region; weak scaling).
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
FLOP_PER_STEP = {"gas": 1.6e4, "aer": 1.9e5, "tot": 5.6e5}   # per internal Ros3 step, SURVEY.md §8d (tot), scaled tables


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


def measured_traffic(mech, ncell):
    """HBM bytes per launch from the most recent committed PMC pass (profiles/rNN_traffic.json), scaled to this launch's
    cell count; None when no pass exists for the mechanism.  bench.py cannot collect PMC counters itself."""
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_traffic.json")), reverse=True):
        d = json.load(open(path))
        if d.get("mech") == mech:
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
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    # calibrate on a few cells, then size the sample for ~budget_s seconds on all cores
    var, fix, rconst = (x.numpy() for x in make_batch(mech, 0, 4, "cpu"))
    dt, _ = _cpu_worker((kind, mech, var, fix, rconst))
    per_cell = dt / 4
    ncell = int(max(cores, min(32768, budget_s * cores / max(per_cell, 1e-6))))
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
    return {"value": ncell / wall, "unit": "chemistry-timesteps/s", "cores": cores, "kind": kind,
            "sample": "%d cells of the same synthetic %s workload (cells 0..%d), %d processes x 1 thread, %.1f s wall, "
                      "%.1f internal steps/cell, %.0f timesteps/s/core" % (ncell, mech, ncell - 1, cores, wall,
                                                                         nstp / ncell, ncell / wall / cores)}


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
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (requires --backend gloo)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.mech)          # before any GPU initialisation in this process (forks workers)

    import torch
    import torch.distributed as dist
    from mistra_amd import chem
    from mistra_amd.workload import make_batch, shard

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

    total_cells = args.cells_per_gpu * world
    start, ncell = shard(total_cells, rank, world)
    chem.init(local_rank)
    var, fix, rconst = make_batch(args.mech, start, ncell, dev)
    out = torch.empty_like(var)
    ierr = torch.empty(ncell, dtype=torch.int32, device=dev)
    stats = torch.empty((ncell, 8), dtype=torch.int32, device=dev)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        chem.integrate_into(args.mech, var, fix, rconst, out, ierr, stats)
    torch.cuda.synchronize()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()       # same stream as the kernel (chem.integrate_into launches on torch's current stream)
        chem.integrate_into(args.mech, var, fix, rconst, out, ierr, stats)
        ev[k][1].record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, args.steps)

    cdev = dev if args.backend == "nccl" else torch.device("cpu")      # gloo reduces host tensors
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    agg = torch.stack([stats[:, 2].sum().double(), (ierr != 1).sum().double(),
                       torch.tensor(float(ncell), device=dev, dtype=torch.float64)]).to(cdev)
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    elapsed = float(t_el.item())
    nstp_total, nfail, cells_done = (float(x) for x in agg.tolist())

    if rank == 0:
        value = total_cells * args.steps / elapsed
        steps_per_cell = nstp_total / cells_done
        achieved = ncell * ALG_BYTES[args.mech] / (kernel_ms * 1e-3) / 1e9
        flops = ncell * steps_per_cell * FLOP_PER_STEP[args.mech] / (kernel_ms * 1e-3)
        traffic, traffic_src = measured_traffic(args.mech, ncell)
        line = {
            "metric": "chemistry-timesteps/sec (%s mechanism)" % args.mech, "value": value, "unit": "chemistry-timesteps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(1, args.steps),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s mechanism, %d synthetic cells per GPU (%d total), INTEGRATE_%s(0,10 s), Ros3 rtol 1e-3; "
                                   "perturbed captured BTZ96 cloud states" % (args.mech, args.cells_per_gpu, total_cells, args.mech[0]),
                       "cells_per_gpu": args.cells_per_gpu, "mean_internal_steps_per_cell": steps_per_cell,
                       "failed_cells": int(nfail), "parallelism": "cells sharded over %d GPU(s), no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "ros3_integrate_kernel", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_cell": ALG_BYTES[args.mech],
                         "fp64_tflops": flops / 1e12, "fp64_frac_of_vector_peak": flops / 1e12 / FP64_VECTOR_PEAK_TFLOPS},
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
